// Small kernels around the quantized linears in a decode step: RMSNorm,
// rotary + KV-cache append + single-query attention, SiLU*mul.  They exist so the decode-token benchmark is a
// whole Llama step inside one hipGraph; none of them is on the packed-weight path itself.
//   RMSNorm    reference: qeft/kernel/layernorm/layernorm.cu:26-76 (generalT5LayerNorm): fp32 sum of squares,
//              y = x * rsqrt(mean(x^2) + eps) * gamma, one rounding to fp16.
//   attention  reference: qeft/kernel/attention (vendored FT masked MHA, ft_attention.cpp:110-181): rotary
//              (neox style = HF rotate_half), append k/v at `timestep`, softmax(q.K^T/sqrt(d)).V.
#include "qeft_common.h"

namespace qeft {

__device__ __forceinline__ float block_sum_256(float v, float* sm) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sm[wave] = v;
    __syncthreads();
    return sm[0] + sm[1] + sm[2] + sm[3];
}

__device__ __forceinline__ float block_max_256(float v, float* sm) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sm[wave] = v;
    __syncthreads();
    return fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
}

// y[m][H] = rmsnorm(x[m][H]) * gamma.  One block per row.  H % 8 == 0.  `res_out` (optional) receives x + add
// when `add` is given (fused residual add before the norm): h = x + add; res_out = h; y = norm(h).
__global__ __launch_bounds__(256) void rmsnorm_kernel(const f16* __restrict__ x, const f16* __restrict__ add,
                                                      const f16* __restrict__ gamma, f16* __restrict__ res_out,
                                                      f16* __restrict__ y, int H, float eps) {
    __shared__ float sm[4];
    const size_t base = (size_t)blockIdx.x * H;
    float ss = 0.f;
    for (int i = threadIdx.x * 8; i < H; i += 256 * 8) {
        h8 v = *(const h8*)(x + base + i);
        if (add) {
            const h8 w = *(const h8*)(add + base + i);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (f16)((float)v[j] + (float)w[j]);
            if (res_out) *(h8*)(res_out + base + i) = v;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) ss += (float)v[j] * (float)v[j];
    }
    ss = block_sum_256(ss, sm);
    const float rs = rsqrtf(ss / (float)H + eps);
    for (int i = threadIdx.x * 8; i < H; i += 256 * 8) {
        h8 v = *(const h8*)(x + base + i);
        if (add) {
            const h8 w = *(const h8*)(add + base + i);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (f16)((float)v[j] + (float)w[j]);
        }
        const h8 g = *(const h8*)(gamma + i);
        h8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (f16)((float)v[j] * rs * (float)g[j]);
        *(h8*)(y + base + i) = o;
    }
}

// out[i] = silu(gate[i]) * up[i]
__global__ __launch_bounds__(256) void silu_mul_kernel(const f16* __restrict__ gate, const f16* __restrict__ up,
                                                       f16* __restrict__ out, int n) {
    const int i = (blockIdx.x * 256 + threadIdx.x) * 8;
    if (i >= n) return;
    const h8 g = *(const h8*)(gate + i), u = *(const h8*)(up + i);
    h8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float gf = (float)g[j];
        o[j] = (f16)(gf / (1.f + __expf(-gf)) * (float)u[j]);
    }
    *(h8*)(out + i) = o;
}

// Single-token attention for one sequence.  grid = n_heads, block = 256, head_dim = 128.
//   q,k,v  : this token's projections [n_heads*128], [n_kv*128], [n_kv*128] (fp16)
//   cos/sin: [max_seq][64] fp32 rotary table
//   kc, vc : caches [n_kv][max_seq][128] fp16;  *pos_ptr = index of this token (0-based)
//   out    : [n_heads*128] fp16
__global__ __launch_bounds__(256) void rope_attn_decode_kernel(const f16* __restrict__ q, const f16* __restrict__ k,
                                                               const f16* __restrict__ v, const float* __restrict__ cs,
                                                               const float* __restrict__ sn, f16* __restrict__ kc,
                                                               f16* __restrict__ vc, const int* __restrict__ pos_ptr,
                                                               const int* __restrict__ out_pos, f16* __restrict__ out,
                                                               int n_heads, int n_kv, int max_seq) {
    constexpr int HD = 128;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_raw[];
    float* sc = (float*)smem_raw;            // [max_seq] scores / probabilities
    float* qs = sc + max_seq;                // [128] rotated, pre-scaled q
    f16* knew = (f16*)(qs + HD);             // [128]
    f16* vnew = knew + HD;                   // [128]
    float* part = (float*)(vnew + HD);       // [16][128] output partials
    __shared__ float sm[4];

    const int h = blockIdx.x, t = threadIdx.x;
    const int grp = n_heads / n_kv, hk = h / grp;
    const int pos = *pos_ptr, L = pos + 1;
    if (pos < 0 || pos >= max_seq) return;   // never index the cache / rotary table out of range
    f16* kch = kc + (size_t)hk * max_seq * HD;
    f16* vch = vc + (size_t)hk * max_seq * HD;

    // Every cache byte this block needs is known as soon as `pos` is: the K quarter-rows and V pieces of the first 256
    // positions are requested up front, right behind the token's own q/k/v and rotary entries, so the kernel pays two
    // dependent memory round trips (pos, then everything) instead of four (pos, q/k/v, K rows, V rows).
    const int qd = t & 3;
    const int dg = t & 15, pg = t >> 4;
    float rc = 0.f, rs = 0.f, ra = 0.f, rb = 0.f;
    if (t < 128) {
        const int i = t & 63;
        rc = cs[(size_t)pos * 64 + i];
        rs = sn[(size_t)pos * 64 + i];
        const f16* src = (t < 64) ? q + h * HD : k + hk * HD;
        ra = (float)src[i];
        rb = (float)src[i + 64];
    } else {
        ra = (float)v[hk * HD + (t - 128)];
    }
    constexpr int KPRE = 4, VPRE = 16;          // 4 passes x 64 positions; 16 x 16 positions
    h8 kpre[KPRE][4], vpre[VPRE];
#pragma unroll
    for (int i = 0; i < KPRE; ++i)
        if (i * 64 < L) {                       // block-uniform
            const h8* row = (const h8*)(kch + (size_t)min(i * 64 + (t >> 2), max_seq - 1) * HD + qd * 32);
#pragma unroll
            for (int j = 0; j < 4; ++j) kpre[i][j] = row[j];
        }
#pragma unroll
    for (int i = 0; i < VPRE; ++i)
        if (i * 16 < L) vpre[i] = *(const h8*)(vch + (size_t)min(pg + 16 * i, max_seq - 1) * HD + dg * 8);

    if (t < 64) {
        const float scale = 0.08838834764831845f;  // 1/sqrt(128)
        qs[t] = (ra * rc - rb * rs) * scale;
        qs[t + 64] = (rb * rc + ra * rs) * scale;
    } else if (t < 128) {
        const int i = t - 64;
        const f16 k0 = (f16)(ra * rc - rb * rs), k1 = (f16)(rb * rc + ra * rs);
        knew[i] = k0;
        knew[i + 64] = k1;
        if (h % grp == 0) {
            kch[(size_t)pos * HD + i] = k0;
            kch[(size_t)pos * HD + i + 64] = k1;
        }
    } else {
        const int i = t - 128;
        const f16 vv = (f16)ra;
        vnew[i] = vv;
        if (h % grp == 0) vch[(size_t)pos * HD + i] = vv;
    }
    __syncthreads();

    // scores: 4 lanes per position (32 dims each, q quarter kept in registers), 64 positions per pass
    float qreg[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) qreg[j] = qs[qd * 32 + j];
    float lmax = -3.0e38f;
    auto score_pass = [&](int p0, const h8* kr) {
        const int p = p0 + (t >> 2);
        float s = 0.f;
        if (p < L) {
            if (p == pos) {
#pragma unroll
                for (int j = 0; j < 32; ++j) s += qreg[j] * (float)knew[qd * 32 + j];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) s += qreg[j * 8 + e] * (float)kr[j][e];
                }
            }
        }
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        if (p < L) {
            if (qd == 0) sc[p] = s;
            lmax = fmaxf(lmax, s);
        }
    };
#pragma unroll
    for (int i = 0; i < KPRE; ++i)
        if (i * 64 < L) score_pass(i * 64, kpre[i]);
    for (int p0 = KPRE * 64; p0 < L; p0 += 64) {
        h8 kr[4];
        const h8* row = (const h8*)(kch + (size_t)min(p0 + (t >> 2), max_seq - 1) * HD + qd * 32);
#pragma unroll
        for (int j = 0; j < 4; ++j) kr[j] = row[j];
        score_pass(p0, kr);
    }
    const float mx = block_max_256(lmax, sm);
    float lsum = 0.f;
    for (int p = t; p < L; p += 256) {
        const float e = __expf(sc[p] - mx);
        sc[p] = e;
        lsum += e;
    }
    const float inv = 1.f / block_sum_256(lsum, sm);
    __syncthreads();

    // P.V: thread = (8-dim group, 1 of 16 position classes); 16-byte V pieces; partials combined through LDS
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.f;
    auto pv = [&](int p, h8 vv) {
        if (p == pos) vv = *(const h8*)(vnew + dg * 8);
        const float w = sc[p];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] += w * (float)vv[e];
    };
#pragma unroll
    for (int i = 0; i < VPRE; ++i)
        if (pg + 16 * i < L) pv(pg + 16 * i, vpre[i]);
    for (int p = pg + 16 * VPRE; p < L; p += 16) pv(p, *(const h8*)(vch + (size_t)p * HD + dg * 8));
#pragma unroll
    for (int e = 0; e < 8; ++e) part[pg * HD + dg * 8 + e] = o[e];
    __syncthreads();
    if (t < HD) {
        float acc = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) acc += part[g * HD + t];
        const int oi = h * HD + t;
        out[out_pos ? out_pos[oi] : oi] = (f16)(acc * inv);
    }
}

hipError_t rmsnorm_launch(const void* x, const void* add, const void* gamma, void* res_out, void* y, int m, int H,
                          float eps, hipStream_t st) {
    hipLaunchKernelGGL(rmsnorm_kernel, dim3(m), dim3(256), 0, st, (const f16*)x, (const f16*)add, (const f16*)gamma,
                       (f16*)res_out, (f16*)y, H, eps);
    return hipGetLastError();
}

hipError_t silu_mul_launch(const void* gate, const void* up, void* out, int n, hipStream_t st) {
    hipLaunchKernelGGL(silu_mul_kernel, dim3((n / 8 + 255) / 256), dim3(256), 0, st, (const f16*)gate, (const f16*)up,
                       (f16*)out, n);
    return hipGetLastError();
}

hipError_t rope_attn_decode_launch(const void* q, const void* k, const void* v, const void* cs, const void* sn, void* kc,
                                   void* vc, const int* pos, const int* out_pos, void* out, int n_heads, int n_kv,
                                   int max_seq, hipStream_t st) {
    const size_t smem = (size_t)max_seq * 4 + 128 * 4 + 2 * 128 * 2 + 16 * 128 * 4;
    if (smem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)rope_attn_decode_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(rope_attn_decode_kernel, dim3(n_heads), dim3(256), smem, st, (const f16*)q, (const f16*)k,
                       (const f16*)v, (const float*)cs, (const float*)sn, (f16*)kc, (f16*)vc, pos, out_pos, (f16*)out,
                       n_heads, n_kv, max_seq);
    return hipGetLastError();
}

}  // namespace qeft
