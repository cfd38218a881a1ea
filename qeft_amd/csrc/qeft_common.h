// Shared device helpers for the gfx950 W4 kernels.  HIP / CDNA4 only.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>

namespace qeft {

typedef _Float16 f16;
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef u32x3 u32x3_u __attribute__((aligned(4)));   // 12-byte records of the 3-bit layout: dword alignment only
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Sum over the 64 lanes of a wave, the same value returned in every lane.  DPP adds inside the 16-lane rows (4 VALU
// instructions with a few cycles of latency each) and four v_readlane across the rows: a __shfl_xor butterfly is six
// DEPENDENT ds_bpermute round trips through the LDS crossbar, ~0.4 us at the end of every launch that carries one.
__device__ __forceinline__ float wave_sum(float v) {
    auto dpp_add = [](float x, auto ctrl_tag) {
        constexpr int CTRL = decltype(ctrl_tag)::value;
        const int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true);
        return x + __builtin_bit_cast(float, t);
    };
    v = dpp_add(v, std::integral_constant<int, 0xB1>{});      // quad_perm [1,0,3,2]
    v = dpp_add(v, std::integral_constant<int, 0x4E>{});      // quad_perm [2,3,0,1]: every lane holds its quad's sum
    v = dpp_add(v, std::integral_constant<int, 0x141>{});     // row_half_mirror: sums of 8
    v = dpp_add(v, std::integral_constant<int, 0x140>{});     // row_mirror: every lane holds its row's (16 lanes) sum
    const int iv = __builtin_bit_cast(int, v);
    return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16))) +
           (__builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48)));
}
__device__ __forceinline__ h2 as_h2(uint32_t v) { return __builtin_bit_cast(h2, v); }
__device__ __forceinline__ uint32_t as_u32(h2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ h2 splat(f16 v) { return h2{v, v}; }

// Checkpoint nibble order (qeft/qlinear.py:81-121, closed form in oracle.nibble_position):
// the 16 bytes of one (row, 32-k chunk) are 4 u32 words; word w holds, in nibble i,
//   i <  4 : k = 2w     + 8i
//   i >= 4 : k = 2w + 1 + 8(i-4)
// so masking with 0x000f000f / 0x00f000f0 (before and after >>8) yields the k-adjacent pairs
//   out[j] = (k = 2w + 8j, k = 2w + 8j + 1),  j = 0..3
// The integer -> fp16 conversion is the 0x6400 exponent trick the reference uses
// (dequantize.cuh:14-77): 0x6400|q == 1024+q, 0x6400|(q<<4) == 1024+16q; both are
// brought back to q exactly, then ONE rounded fp16 FMA q*s+sz gives the same fp16
// weight as the reference's __hfma2 (gemv_cuda_qeft.cu:158, gemm_cuda.cu:280-286).
__device__ __forceinline__ void nib8_to_q(uint32_t v, h2 (&q)[4]) {
    uint32_t MAGIC = 0x64006400u;
    asm volatile("" : "+v"(MAGIC));          // opaque: (v & mask) | MAGIC becomes ONE v_and_or_b32 (literal + VGPR) instead of and + or
    const h2 k1024 = {(f16)1024.f, (f16)1024.f};
    const h2 k16th = {(f16)0.0625f, (f16)0.0625f};
    const h2 kn64 = {(f16)-64.f, (f16)-64.f};
    const uint32_t t = v >> 8;
    q[0] = as_h2((v & 0x000f000fu) | MAGIC) - k1024;
    q[1] = __builtin_elementwise_fma(as_h2((v & 0x00f000f0u) | MAGIC), k16th, kn64);
    q[2] = as_h2((t & 0x000f000fu) | MAGIC) - k1024;
    q[3] = __builtin_elementwise_fma(as_h2((t & 0x00f000f0u) | MAGIC), k16th, kn64);
}

__device__ __forceinline__ void dequant8(uint32_t v, h2 s, h2 z, h2 (&w)[4]) {
    h2 q[4];
    nib8_to_q(v, q);
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = __builtin_elementwise_fma(q[j], s, z);
}

// 3-bit EXTENSION layout (oracle.pack_w3 / w3_position): a lane record (row, 32-k chunk) is three words; pair e (k = 2e, 2e + 1 of
// the chunk, natural order) sits at bits 3 (e % 5) of each half-word of word e / 5 for e < 15, pair 15 in bits 15 / 31 of the
// three words (bit b of the value in word b).  q[e] = the exact integer pair as fp16 (0x6400 trick, then - 1024).
__device__ __forceinline__ h2 w3_pair_q(uint32_t w0, uint32_t w1, uint32_t w2, int e, uint32_t MAGIC) {
    const h2 k1024 = {(f16)1024.f, (f16)1024.f};
    uint32_t f;
    if (e < 15) {
        const uint32_t v = e < 5 ? w0 : e < 10 ? w1 : w2;
        f = (v >> (3 * (e % 5))) & 0x00070007u;
    } else {
        f = ((w0 >> 15) & 0x00010001u) | ((w1 >> 14) & 0x00020002u) | ((w2 >> 13) & 0x00040004u);
    }
    return as_h2(f | MAGIC) - k1024;
}

// Arguments of the decode GEMV (gemv_w4.hip); built by the C ABI (capi.hip).
struct GemvArgs {
    const f16* x;
    const uint8_t* qw;
    const f16* scales;
    const f16* zeros;
    const f16* ow_il;
    const f16* bias;
    const int* ids;
    const f16* residual;
    f16* y;
    int N, K, G, n_out;
    int gshift;  // log2(G) for power-of-two groups, 31 when G == K (single group)
    const f16* xt_aux;   // x transform operand (gamma or up), see gemv_w4_kernel.h XT
    float xt_eps;
    const uint32_t* sz_blk;  // optional shadow [N/16][K/G][16] of (scale | scaled_zero << 16), see qeft_pack_scales
    unsigned long long* dbg; // tools/gemv_lab.hip only (ABL & 16): per-block time stamps; always NULL in the product
    unsigned long long* dbg2;
    const f16* ow_plain;     // outlier slice as plain [N, n_out] rows instead of ow_il (small-M route of the GEMM entry)
    int m_rt;                // real batch rows when the kernel is the M == 16 instantiation
};

struct GemvGroupArgs {
    const f16* x;
    const uint8_t* qw[3];
    const f16* scales[3];
    const f16* zeros[3];
    const f16* ow_il[3];
    const f16* bias[3];
    f16* y[3];
    int N[3];
    int blk_end[3];
    int K, G, n_out, gshift;
    const f16* xt_aux;
    float xt_eps;
    const uint32_t* sz_blk[3];
};

// Name of the kernel variant the calling thread's most recent entry point dispatched to (qeft_last_variant(), capi.hip).
// Tests assert it so that coverage cannot silently fall off a routing threshold.
extern thread_local const char* g_last_variant;

__device__ __forceinline__ float dot2(h2 a, h2 b, float c) { return __builtin_amdgcn_fdot2(a, b, c, false); }

// silu(g) = g / (1 + e^-g) in fp32 with the hardware reciprocal (1 ulp) instead of an IEEE division: the result is
// rounded to fp16 right after, and a division costs ~10 VALU instructions in a prologue every block repeats.
__device__ __forceinline__ float silu_f32(float g) { return g * __builtin_amdgcn_rcpf(1.f + __expf(-g)); }
// fp16(a * b) with the product rounded to fp32 FIRST -- what torch (and the reference's RMSNorm) computes.  Written out because
// hipcc may otherwise pick v_fma_mixlo_f16 (one rounding, different in 1 of ~10^4 values), and whether it does changes with the
// code around it: the norm outputs of different kernels (GEMV epilogue, residual_norm, token_begin) must agree bit for bit.
__device__ __forceinline__ f16 mul_f32_to_f16(float a, float b) {
    float p = a * b;
    asm volatile("" : "+v"(p));
    return (f16)p;
}

}  // namespace qeft
