// Decode GEMV for the QEFT packed W4 (+ fp16 outlier slice) linear on gfx950.
//
// Replaces gemv_kernel / gemv_kernel_qeft (+ perchannel twins) of
// qeft/kernel/quantization_new/gemv/gemv_cuda{,_qeft}.cu.  Not a translation: the
// reference tiles for 32-wide warps (8 rows x 2048 k per 256-thread block, fp16
// accumulation); this kernel is laid out for 64-wide wavefronts and HBM streaming:
//
//  * The checkpoint stores 4 output rows interleaved per 64-k tile, i.e. one row-group
//    (4 rows) is ONE contiguous stream of 2K bytes.  A wave-wide 16 B/lane load covers
//    1 KiB = 4 rows x 512 k, perfectly coalesced; lane l owns row (l>>1)&3 and the 32 k
//    starting at (l>>3)*64 + (l&1)*32 of that 512-k step.
//  * A 256-thread block owns RG row-groups (4*RG rows).  Its 4 waves split K by 512-k
//    steps (wave w takes steps w, w+4, ...), so the x slice a lane loads (64 B per batch
//    row) is reused in registers for all RG row-groups and never crosses waves: no LDS
//    staging and no barrier in front of the dot.  Loads go straight to VGPRs and the next
//    step's loads are issued before the current step's math (guide: "GEMV / M <= 16").
//  * fp16 dequant is bit-identical to the reference (one rounded FMA per weight), the dot
//    accumulates in fp32 with v_dot2c_f32_f16, lanes sharing a row are combined with
//    4 xor-shuffles, the 4 waves through 1.8 KB of LDS.
//  * The fp16 outlier slice replaces the dead nibbles of the last n_out columns: lanes
//    whose 32-k chunk lies there skip the INT4 load and read 128 B of oweight_interleaved.
//  * Optional fusions (qeft_gemv_w4_fused): o_proj input gather through LDS, bias, residual.
#include "qeft_common.h"

namespace qeft {


constexpr int kWaves = 4;
constexpr int kBlock = kWaves * 64;
constexpr int kStep = 512;  // k covered by one wave-wide 16 B/lane load

template <int RG, int M, bool XPF>
struct StepRegs {
    u32x4 w[RG];
    uint32_t s[RG], z[RG];          // fp16 bits, zero-extended by global_load_ushort
    u32x4 xv[XPF ? M : 1][4];       // x slice, prefetched with the weights only for small M
};

// Issue the loads of one 512-k step: RG x 16 B of nibbles, RG scale/zero pairs and (small M) the x slice.
template <int RG, int M, bool XPF, bool XLDS>
__device__ __forceinline__ void load_step(const GemvArgs& a, int it, int lane, int rg0, int r, int koff, int kq,
                                          const f16* xs, StepRegs<RG, M, XPF>& b) {
    const int k0 = it * kStep + koff;
    if (k0 < kq) {
        const int g = k0 / a.G;
#pragma unroll
        for (int i = 0; i < RG; ++i) {
            const uint8_t* p = a.qw + (size_t)(rg0 + i) * a.K * 2 + (size_t)it * 1024 + lane * 16;
            b.w[i] = __builtin_nontemporal_load((const u32x4*)p);
            const size_t so = (size_t)g * a.N + (rg0 + i) * 4 + r;
            b.s[i] = ((const uint16_t*)a.scales)[so];
            b.z[i] = ((const uint16_t*)a.zeros)[so];
        }
        if (XPF) {
#pragma unroll
            for (int bm = 0; bm < M; ++bm) {
                const u32x4* px = (const u32x4*)((XLDS ? xs : a.x) + (size_t)bm * a.K + k0);
#pragma unroll
                for (int j = 0; j < 4; ++j) b.xv[bm][j] = px[j];
            }
        }
    }
}

template <int RG, int M, bool XPF, bool XLDS>
__device__ __forceinline__ void compute_step(const GemvArgs& a, int it, int koff, int kq, const f16* xs,
                                             const StepRegs<RG, M, XPF>& b, float (&acc)[RG][M]) {
    const int k0 = it * kStep + koff;
    if (k0 >= kq) return;
    h2 wd[RG][4][4];
#pragma unroll
    for (int i = 0; i < RG; ++i) {
        const h2 s = as_h2(b.s[i] * 0x10001u), z = as_h2(b.z[i] * 0x10001u);
#pragma unroll
        for (int w = 0; w < 4; ++w) dequant8(b.w[i][w], s, z, wd[i][w]);
    }
#pragma unroll
    for (int bm = 0; bm < M; ++bm) {
        u32x4 xv[4];
        if (XPF) {
#pragma unroll
            for (int j = 0; j < 4; ++j) xv[j] = b.xv[bm][j];
        } else {
            const u32x4* px = (const u32x4*)((XLDS ? xs : a.x) + (size_t)bm * a.K + k0);
#pragma unroll
            for (int j = 0; j < 4; ++j) xv[j] = px[j];
        }
#pragma unroll
        for (int i = 0; i < RG; ++i)
#pragma unroll
            for (int w = 0; w < 4; ++w)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][bm] = dot2(wd[i][w][j], as_h2(xv[j][w]), acc[i][bm]);
    }
}

template <int RG, int M, bool OUTL, bool XLDS>
__global__ __launch_bounds__(kBlock) void gemv_w4_kernel(GemvArgs a) {
    constexpr bool XPF = (M <= 2);
    constexpr int NIR = OUTL ? (RG == 1 ? 4 : RG * 2) : 0;  // oweight_interleaved rows this block touches
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    float* red = (float*)smem;                   // [kWaves][M][RG*4]
    float* part = red + kWaves * RG * 4 * M;     // [NIR][32][2][M] outlier partial sums
    f16* xs = (f16*)(part + NIR * 32 * 2 * M);   // [M][K] gathered input (XLDS only)

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int r = (lane >> 1) & 3;
    const int koff = (lane >> 3) * 64 + (lane & 1) * 32;
    const int rg0 = blockIdx.x * RG;
    const int kq = a.K - (OUTL ? a.n_out : 0);   // INT4 columns: [0, kq); fp16 outlier slice: [kq, K)
    const int nsteps = (kq + kStep - 1) / kStep;

    StepRegs<RG, M, XPF> cur, nxt;
    int it = wave;
    // the weight stream does not depend on x: get the first step in flight before anything else
    if (!XLDS && it < nsteps) load_step<RG, M, XPF, XLDS>(a, it, lane, rg0, r, koff, kq, nullptr, cur);

    if (XLDS) {
        // gathered input (qlinear.py:275): xs[bm][k] = x[bm][ids[k]]
        for (int idx = threadIdx.x; idx < M * a.K; idx += kBlock) {
            const int bm = idx / a.K, k = idx - bm * a.K;
            xs[idx] = a.x[(size_t)bm * a.K + a.ids[k]];
        }
        __syncthreads();
        if (it < nsteps) load_step<RG, M, XPF, XLDS>(a, it, lane, rg0, r, koff, kq, xs, cur);
    }

    if (OUTL) {
        // fp16 outlier slice (gemv_cuda_qeft.cu:170-176).  oweight_interleaved row (n/8)*4 + n%4 holds, per
        // 32-column chunk, 64 halves alternating rows n (n%8 < 4) and n+4 (qlinear.py:70-79).  32 threads per
        // interleaved row, 16 B (4 column pairs) per thread and pass; partial sums go through LDS in a fixed order.
        const int ir = threadIdx.x >> 5, sl = threadIdx.x & 31;
        if (ir < NIR) {
            float lo[M], hi[M];
#pragma unroll
            for (int bm = 0; bm < M; ++bm) lo[bm] = hi[bm] = 0.f;
            const f16* orow = a.ow_il + ((size_t)(rg0 >> 1) * 4 + ir) * (size_t)(2 * a.n_out);
            const f16* xo = (XLDS ? xs : a.x) + kq;
            for (int seg = sl; seg < a.n_out / 4; seg += 32) {
                const u32x4 wv = *(const u32x4*)(orow + seg * 8);
                const int col = (seg >> 3) * 32 + (seg & 7) * 4;
#pragma unroll
                for (int bm = 0; bm < M; ++bm) {
                    const u32x2 xv = *(const u32x2*)(xo + (size_t)bm * a.K + col);
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const h2 wp = as_h2(wv[p]);
                        const float xf = (float)as_h2(xv[p >> 1])[p & 1];
                        lo[bm] += (float)wp[0] * xf;
                        hi[bm] += (float)wp[1] * xf;
                    }
                }
            }
#pragma unroll
            for (int bm = 0; bm < M; ++bm) {
                part[((ir * 32 + sl) * 2 + 0) * M + bm] = lo[bm];
                part[((ir * 32 + sl) * 2 + 1) * M + bm] = hi[bm];
            }
        }
    }

    float acc[RG][M];
#pragma unroll
    for (int i = 0; i < RG; ++i)
#pragma unroll
        for (int bm = 0; bm < M; ++bm) acc[i][bm] = 0.f;

    for (; it < nsteps; it += kWaves) {
        const int itn = it + kWaves;
        if (itn < nsteps) load_step<RG, M, XPF, XLDS>(a, itn, lane, rg0, r, koff, kq, xs, nxt);
        compute_step<RG, M, XPF, XLDS>(a, it, koff, kq, xs, cur, acc);
        cur = nxt;
    }

    // lanes sharing a row differ in lane bits 0,3,4,5
#pragma unroll
    for (int i = 0; i < RG; ++i)
#pragma unroll
        for (int bm = 0; bm < M; ++bm) {
            float v = acc[i][bm];
            v += __shfl_xor(v, 1);
            v += __shfl_xor(v, 8);
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            acc[i][bm] = v;
        }
    if ((lane & 0x39) == 0) {  // lanes 0,2,4,6 -> r = lane>>1
#pragma unroll
        for (int i = 0; i < RG; ++i)
#pragma unroll
            for (int bm = 0; bm < M; ++bm) red[wave * (RG * 4 * M) + bm * (RG * 4) + i * 4 + r] = acc[i][bm];
    }
    __syncthreads();
    if (threadIdx.x < RG * 4 * M) {
        const int bm = threadIdx.x / (RG * 4), lrow = threadIdx.x % (RG * 4);
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) v += red[w * (RG * 4 * M) + threadIdx.x];
        if (OUTL) {
            const int ir = (RG == 1) ? lrow : ((lrow >> 3) * 4 + (lrow & 3));
            const int hsel = (RG == 1) ? (rg0 & 1) : ((lrow >> 2) & 1);
            float o = 0.f;
            for (int sl = 0; sl < 32; ++sl) o += part[((ir * 32 + sl) * 2 + hsel) * M + bm];
            v += o;
        }
        const int row = rg0 * 4 + lrow;
        if (a.bias) v += (float)a.bias[row];
        if (a.residual) v += (float)a.residual[(size_t)bm * a.N + row];
        a.y[(size_t)bm * a.N + row] = (f16)v;
    }
}

template <int RG, int M, bool OUTL, bool XLDS>
static hipError_t launch(const GemvArgs& a, hipStream_t st) {
    const int grid = a.N / (4 * RG);
    constexpr int NIR = OUTL ? (RG == 1 ? 4 : RG * 2) : 0;
    size_t smem = (kWaves * RG * 4 * M + NIR * 32 * 2 * M) * sizeof(float);
    if (XLDS) smem += (size_t)M * a.K * sizeof(f16);
    if (smem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)gemv_w4_kernel<RG, M, OUTL, XLDS>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((gemv_w4_kernel<RG, M, OUTL, XLDS>), dim3(grid), dim3(kBlock), smem, st, a);
    return hipGetLastError();
}

template <int RG, int M>
static hipError_t launch_rg_m(const GemvArgs& a, hipStream_t st) {
    const bool outl = a.n_out > 0, xlds = a.ids != nullptr;
    if (outl) return xlds ? launch<RG, M, true, true>(a, st) : launch<RG, M, true, false>(a, st);
    return xlds ? launch<RG, M, false, true>(a, st) : launch<RG, M, false, false>(a, st);
}

template <int RG>
static hipError_t launch_rg(const GemvArgs& a, int m, hipStream_t st) {
    switch (m) {
        case 1: return launch_rg_m<RG, 1>(a, st);
        case 2: return launch_rg_m<RG, 2>(a, st);
        case 3: return launch_rg_m<RG, 3>(a, st);
        case 4: return launch_rg_m<RG, 4>(a, st);
        case 5: return launch_rg_m<RG, 5>(a, st);
        case 6: return launch_rg_m<RG, 6>(a, st);
        default: return launch_rg_m<RG, 7>(a, st);
    }
}

// Rows per block: keep >= ~2 blocks per CU (256 CUs) when N allows, otherwise fewer rows per
// block so small (sharded) layers still cover the chip.
hipError_t gemv_w4_dispatch(const GemvArgs& a, int m, hipStream_t st) {
    const int rgs = a.N / 4;
    int rg = 4;
    if (m >= 3) rg = 2;  // register budget: the dequantised RG x 32 weights stay live across the batch rows
    while (rg > 1 && (rgs % rg != 0 || rgs / rg < 512)) rg >>= 1;
    if (a.n_out > 0 && rg == 1 && (rgs & 1)) return hipErrorInvalidValue;  // interleave pairs need N % 8 == 0
    switch (rg) {
        case 4: return launch_rg<4>(a, m, st);
        case 2: return launch_rg<2>(a, m, st);
        default: return launch_rg<1>(a, m, st);
    }
}

}  // namespace qeft
