// Decode GEMV for the QEFT packed W4 (+ fp16 outlier slice) linear on gfx950: launch logic.
//
// Replaces gemv_kernel / gemv_kernel_qeft (+ perchannel twins) of
// qeft/kernel/quantization_new/gemv/gemv_cuda{,_qeft}.cu.  Not a translation: the reference tiles for 32-wide warps
// (8 rows x 2048 k per 256-thread block, fp16 accumulation).  Two device implementations live in headers:
//
//  * gemv_w4_mfma.h   the production path (K % 128 == 0, n_out % 128 == 0, group 128 or per-channel, N % 16 == 0): one
//                     wave-wide 16 B/lane load = 16 rows x 128 k = the B operand of v_mfma_f32_16x16x32_f16; x, the
//                     scale shadow and the de-interleaved outlier slab staged in LDS once per long-lived block; fused
//                     gather / RMSNorm / SiLU*up / bias / residual; 1..16 batch rows per weight pass.
//  * gemv_w4_kernel.h the first (VALU, v_dot2c) formulation, kept for everything the MFMA path does not take: other
//                     group sizes, n_out in {32, 64, 96}, ragged N.  Bit-identical fp16 dequantisation to the
//                     reference (one rounded FMA per weight), fp32 accumulation.
//
// This file picks the kernel, the grid (row sets per block, XCD-contiguous) and the ring depth, slices a batch that
// does not fit the LDS, and assembles the grouped (q|k|v, gate|up) launches.
#include <cstdlib>

#include "gemv_w4_mfma.h"

namespace qeft {

constexpr int kNW = 8;  // waves per block


constexpr size_t kMaxLds = 160 * 1024;

template <int RGI, int M, int D, bool OUTL, bool XG>
static hipError_t launch(const GemvArgs& a, hipStream_t st) {
    const int grid = a.N / (4 * RGI);
    const size_t smem = gemv_smem_bytes(kNW, RGI, M, a.K, a.G, a.n_out);
    auto kern = gemv_w4_kernel<kNW, RGI, M, D, OUTL, XG, 0>;
    if (smem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
    }
    g_last_variant = "gemv_valu";
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kNW * 64), smem, st, a);
    return hipGetLastError();
}

template <int RGI, int M, int D>
static hipError_t launch_rm(const GemvArgs& a, hipStream_t st) {
    const bool outl = a.n_out > 0, xg = a.ids != nullptr;
    if (xg) return outl ? launch<RGI, M, D, true, true>(a, st) : launch<RGI, M, D, false, true>(a, st);
    return outl ? launch<RGI, M, D, true, false>(a, st) : launch<RGI, M, D, false, false>(a, st);
}

template <int RGI>
static hipError_t launch_r(const GemvArgs& a, int m, hipStream_t st) {
    // ring depth: 2 steps in flight per wave when the wave only has a handful of steps (K <= 6144 with 8 waves),
    // 4 for long rows (measured on MI355X: tools/gemv_lab.hip)
    const bool deep = a.K > 6144;
    switch (m) {
        case 1: return deep ? launch_rm<RGI, 1, 4>(a, st) : launch_rm<RGI, 1, 2>(a, st);
        case 2: return deep ? launch_rm<RGI, 2, 4>(a, st) : launch_rm<RGI, 2, 2>(a, st);
        case 3: return deep ? launch_rm<RGI, 3, 4>(a, st) : launch_rm<RGI, 3, 2>(a, st);
        case 4: return deep ? launch_rm<RGI, 4, 4>(a, st) : launch_rm<RGI, 4, 2>(a, st);
        case 5: return deep ? launch_rm<RGI, 5, 4>(a, st) : launch_rm<RGI, 5, 2>(a, st);
        case 6: return deep ? launch_rm<RGI, 6, 4>(a, st) : launch_rm<RGI, 6, 2>(a, st);
        default: return deep ? launch_rm<RGI, 7, 4>(a, st) : launch_rm<RGI, 7, 2>(a, st);
    }
}

// ---- MFMA formulation (gemv_w4_mfma.h): the production path for the Llama shapes
static bool mfma_ok(int N, int K, int G, int n_out) {
    // no lower bound on N: a row's arithmetic must not depend on how many rows share its launch (row-sharding, §8e)
    return K % 128 == 0 && n_out % 128 == 0 && (G == 128 || G == K) && N % 16 == 0;
}

// Blocks for a layer with `nsets` 16-row sets.  Every block pays ~1 us of start-up plus the x staging before its
// first MFMA (tools/gemv_lab.hip timeline), so wide layers run as 256*k long-lived blocks that each walk ~3 row sets
// with the ring never draining; narrow layers (< 2 sets per CU) keep one block per set.  QEFT_GEMV_BLOCKS overrides.
static int mfma_grid(int nsets) {
    static int forced = -1;
    if (forced < 0) {
        const char* e = getenv("QEFT_GEMV_BLOCKS");
        forced = e ? atoi(e) : 0;
    }
    if (forced > 0) return nsets < forced ? nsets : forced;
    if (nsets < 512) return nsets;
    int k = (nsets + 384) / 768;
    if (k < 1) k = 1;
    return 256 * k;
}
static int ceil_div(int a, int b) { return (a + b - 1) / b; }

template <int M, int D>
static hipError_t launch_mfma(const GemvArgs& a, hipStream_t st) {
    const int nsets = a.N / 16;
    int nblk = (M == 1) ? mfma_grid(nsets) : nsets;
    int rs_cap = ceil_div(nsets, nblk);
    if (gemv_mfma_smem_bytes(kNW, M, a.K, a.n_out, rs_cap) > 64 * 1024) { nblk = nsets; rs_cap = 1; }
    const size_t smem = gemv_mfma_smem_bytes(kNW, M, a.K, a.n_out, rs_cap);
    const dim3 grid(nblk), block(kNW * 64);
    const bool outl = a.n_out > 0, xg = a.ids != nullptr;
#define QEFT_LAUNCH(OUTL_, XG_)                                                                                        \
    do {                                                                                                                \
        auto kern = gemv_w4_mfma_kernel<kNW, M, D, OUTL_, XG_, 0, 0>;                                                    \
        if (smem > 64 * 1024) {                                                                                         \
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
            if (e != hipSuccess) return e;                                                                              \
        }                                                                                                               \
        hipLaunchKernelGGL(kern, grid, block, smem, st, a, rs_cap);                                                     \
    } while (0)
    if (xg) { if (outl) QEFT_LAUNCH(true, true); else QEFT_LAUNCH(false, true); }
    else { if (outl) QEFT_LAUNCH(true, false); else QEFT_LAUNCH(false, false); }
#undef QEFT_LAUNCH
    g_last_variant = "gemv_mfma";
    return hipGetLastError();
}

template <int D>
static hipError_t launch_mfma_m(const GemvArgs& a, int m, hipStream_t st) {
    switch (m) {
        case 1: return launch_mfma<1, D>(a, st);
        case 2: return launch_mfma<2, D>(a, st);
        case 3: return launch_mfma<3, D>(a, st);
        case 4: return launch_mfma<4, D>(a, st);
        case 5: return launch_mfma<5, D>(a, st);
        case 6: return launch_mfma<6, D>(a, st);
        default: return launch_mfma<7, D>(a, st);
    }
}

// ring depth (steps in flight per wave), measured with tools/gemv_lab.hip and bench.py: 4, 6 for K > 6144; deeper is slower
static int mfma_depth(int K);
// Small-M route of the GEMM entry (8 <= M <= ~128): the MFMA GEMV contracts up to 16 batch rows per pass at the
// cost of one weight stream, far cheaper than a 128-row GEMM tile at these sizes.  Processes rows in slices of <= 16.
hipError_t gemv_w4_smallm_dispatch(const GemvArgs& a0, int m, hipStream_t st) {
    if (!mfma_ok(a0.N, a0.K, a0.G, a0.n_out)) return hipErrorNotSupported;
    int mmax = 16;
    while (mmax > 4 && gemv_mfma_smem_bytes(kNW, mmax, a0.K, a0.n_out) > kMaxLds) --mmax;
    if (gemv_mfma_smem_bytes(kNW, mmax, a0.K, a0.n_out) > kMaxLds) return hipErrorNotSupported;
    const int nsets = a0.N / 16;
    for (int m0 = 0; m0 < m; m0 += mmax) {
        GemvArgs a = a0;
        a.m_rt = (m - m0 < mmax) ? m - m0 : mmax;
        a.x = a0.x + (size_t)m0 * a0.K;
        a.y = a0.y + (size_t)m0 * a0.N;
        if (a.m_rt < 4) {   // tail slice: the fixed-M kernels (M = 1..3)
            a.ow_plain = a0.ow_plain;
            GemvArgs b = a;
            hipError_t e = b.m_rt == 1 ? launch_mfma<1, 4>(b, st) : b.m_rt == 2 ? launch_mfma<2, 4>(b, st) : launch_mfma<3, 4>(b, st);
            if (e != hipSuccess) return e;
            continue;
        }
        const size_t smem = gemv_mfma_smem_bytes(kNW, a.m_rt, a.K, a.n_out);
        const dim3 grid(nsets), block(kNW * 64);
        hipError_t e = hipSuccess;
        if (a.n_out > 0) {
            auto kern = gemv_w4_mfma_kernel<kNW, 16, 4, true, false, 0, 0>;
            if (smem > 64 * 1024) e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(kern, grid, block, smem, st, a, 1);
        } else {
            auto kern = gemv_w4_mfma_kernel<kNW, 16, 4, false, false, 0, 0>;
            if (smem > 64 * 1024) e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(kern, grid, block, smem, st, a, 1);
        }
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    g_last_variant = "gemv_smallm";
    return hipSuccess;
}

static int mfma_depth(int K) {
    static const char* env = getenv("QEFT_GEMV_DEPTH");   // lab: force 4 or 6
    if (env) return atoi(env) == 6 ? 6 : 4;
    return K > 6144 ? 6 : 4;
}

// Row-groups per wave-load: 4 (16 rows per block) when that still gives >= 1 block per CU, fewer for small
// (e.g. row-sharded) layers so they still cover the 256 CUs.  The staged activations must fit the 160 KB LDS:
// a batch that does not is processed in slices (the weights are then streamed once per slice).
hipError_t gemv_w4_dispatch(const GemvArgs& a0, int m, hipStream_t st) {
    if (mfma_ok(a0.N, a0.K, a0.G, a0.n_out)) {
        int mmax = m;
        while (mmax > 1 && gemv_mfma_smem_bytes(kNW, mmax, a0.K, a0.n_out) > kMaxLds) --mmax;
        if (gemv_mfma_smem_bytes(kNW, mmax, a0.K, a0.n_out) <= kMaxLds) {
            for (int m0 = 0; m0 < m; m0 += mmax) {
                const int mc = (m - m0 < mmax) ? m - m0 : mmax;
                GemvArgs a = a0;
                a.x = a0.x + (size_t)m0 * a0.K;
                a.y = a0.y + (size_t)m0 * a0.N;
                if (a0.residual) a.residual = a0.residual + (size_t)m0 * a0.N;
                const hipError_t e = mfma_depth(a.K) == 6 ? launch_mfma_m<6>(a, mc, st) : launch_mfma_m<4>(a, mc, st);
                if (e != hipSuccess) return e;
            }
            return hipSuccess;
        }
    }
    const int rgs = a0.N / 4;
    int rgi = 4;
    while (rgi > 1 && (rgs % rgi != 0 || rgs / rgi < 256)) rgi >>= 1;
    if (a0.n_out > 0 && rgi == 1 && (rgs & 1)) return hipErrorInvalidValue;  // interleave pairs need N % 8 == 0
    int mmax = m;
    while (mmax > 1 && gemv_smem_bytes(kNW, rgi, mmax, a0.K, a0.G, a0.n_out) > kMaxLds) --mmax;
    if (gemv_smem_bytes(kNW, rgi, mmax, a0.K, a0.G, a0.n_out) > kMaxLds) return hipErrorInvalidValue;
    for (int m0 = 0; m0 < m; m0 += mmax) {
        const int mc = (m - m0 < mmax) ? m - m0 : mmax;
        GemvArgs a = a0;
        a.x = a0.x + (size_t)m0 * a0.K;
        a.y = a0.y + (size_t)m0 * a0.N;
        if (a0.residual) a.residual = a0.residual + (size_t)m0 * a0.N;
        hipError_t e;
        switch (rgi) {
            case 4: e = launch_r<4>(a, mc, st); break;
            case 2: e = launch_r<2>(a, mc, st); break;
            default: e = launch_r<1>(a, mc, st); break;
        }
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// ---- grouped launch (decode engine): up to 3 linears sharing x, batch 1, optional fused RMSNorm on x
template <int RGI, int D>
static hipError_t launch_group(const GemvGroupArgs& g, int nblocks, hipStream_t st) {
    const size_t smem = gemv_smem_bytes(kNW, RGI, 1, g.K, g.G, g.n_out);
    const dim3 grid(nblocks), block(kNW * 64);
    g_last_variant = "gemv_valu_group";
    if (g.xt_aux) {
        if (g.n_out > 0) hipLaunchKernelGGL((gemv_w4_group_kernel<kNW, RGI, 1, D, true, 1>), grid, block, smem, st, g);
        else hipLaunchKernelGGL((gemv_w4_group_kernel<kNW, RGI, 1, D, false, 1>), grid, block, smem, st, g);
    } else {
        if (g.n_out > 0) hipLaunchKernelGGL((gemv_w4_group_kernel<kNW, RGI, 1, D, true, 0>), grid, block, smem, st, g);
        else hipLaunchKernelGGL((gemv_w4_group_kernel<kNW, RGI, 1, D, false, 0>), grid, block, smem, st, g);
    }
    return hipGetLastError();
}

template <int D>
static hipError_t launch_mfma_group(const GemvGroupArgs& g, int nblocks, int rs_cap, hipStream_t st) {
    const size_t smem = gemv_mfma_smem_bytes(kNW, 1, g.K, g.n_out, rs_cap);
    const dim3 grid(nblocks), block(kNW * 64);
    g_last_variant = "gemv_mfma_group";
    if (g.xt_aux) {
        if (g.n_out > 0) hipLaunchKernelGGL((gemv_w4_mfma_group_kernel<kNW, D, true, 1>), grid, block, smem, st, g, rs_cap);
        else hipLaunchKernelGGL((gemv_w4_mfma_group_kernel<kNW, D, false, 1>), grid, block, smem, st, g, rs_cap);
    } else {
        if (g.n_out > 0) hipLaunchKernelGGL((gemv_w4_mfma_group_kernel<kNW, D, true, 0>), grid, block, smem, st, g, rs_cap);
        else hipLaunchKernelGGL((gemv_w4_mfma_group_kernel<kNW, D, false, 0>), grid, block, smem, st, g, rs_cap);
    }
    return hipGetLastError();
}

hipError_t gemv_w4_group_dispatch(GemvGroupArgs g, int nparts, hipStream_t st) {
    {
        int ntot = 0;
        bool ok = g.K <= 16384 && gemv_mfma_smem_bytes(kNW, 1, g.K, g.n_out) <= 64 * 1024;
        for (int p = 0; p < nparts; ++p) {
            ok = ok && g.N[p] % 16 == 0;
            ntot += g.N[p];
        }
        if (ok && mfma_ok(ntot, g.K, g.G, g.n_out)) {
            // blocks per part proportional to its rows (every part gets >= 1 and <= its row sets)
            const int nsets = ntot / 16;
            const int total_blocks = mfma_grid(nsets);
            int acc = 0, rs_cap = 1;
            for (int p = 0; p < 3; ++p) {
                if (p < nparts) {
                    const int sp = g.N[p] / 16;
                    int bp = (int)((long long)total_blocks * sp / nsets);
                    if (bp < 1) bp = 1;
                    if (bp > sp) bp = sp;
                    acc += bp;
                    if (ceil_div(sp, bp) > rs_cap) rs_cap = ceil_div(sp, bp);
                }
                g.blk_end[p] = acc;
            }
            if (gemv_mfma_smem_bytes(kNW, 1, g.K, g.n_out, rs_cap) > 64 * 1024) {
                acc = 0;
                rs_cap = 1;
                for (int p = 0; p < 3; ++p) {
                    if (p < nparts) acc += g.N[p] / 16;
                    g.blk_end[p] = acc;
                }
            }
            return mfma_depth(g.K) == 6 ? launch_mfma_group<6>(g, acc, rs_cap, st) : launch_mfma_group<4>(g, acc, rs_cap, st);
        }
    }
    int total_rgs = 0;
    for (int p = 0; p < nparts; ++p) total_rgs += g.N[p] / 4;
    int rgi = 4;
    for (;;) {
        bool ok = total_rgs / rgi >= 256 || rgi == 1;
        for (int p = 0; p < nparts; ++p) ok = ok && ((g.N[p] / 4) % rgi == 0);
        if (ok || rgi == 1) break;
        rgi >>= 1;
    }
    int acc = 0;
    for (int p = 0; p < 3; ++p) {
        if (p < nparts) acc += g.N[p] / (4 * rgi);
        g.blk_end[p] = acc;
    }
    if (gemv_smem_bytes(kNW, rgi, 1, g.K, g.G, g.n_out) > 64 * 1024) return hipErrorInvalidValue;
    if (g.xt_aux && g.K > 4 * 8 * kNW * 64) return hipErrorInvalidValue;   // transform works on register-held x
    const bool deep = g.K > 6144;
    switch (rgi) {
        case 4: return deep ? launch_group<4, 4>(g, acc, st) : launch_group<4, 2>(g, acc, st);
        case 2: return deep ? launch_group<2, 4>(g, acc, st) : launch_group<2, 2>(g, acc, st);
        default: return deep ? launch_group<1, 4>(g, acc, st) : launch_group<1, 2>(g, acc, st);
    }
}

// ---- down_proj of the decode engine: x := silu(gate) * up fused into the staging, batch 1
template <int RGI, int D>
static hipError_t launch_silu(const GemvArgs& a, hipStream_t st) {
    const size_t smem = gemv_smem_bytes(kNW, RGI, 1, a.K, a.G, a.n_out);
    const dim3 grid(a.N / (4 * RGI)), block(kNW * 64);
    g_last_variant = "gemv_valu_silu";
    if (a.n_out > 0) hipLaunchKernelGGL((gemv_w4_kernel<kNW, RGI, 1, D, true, false, 0, 2>), grid, block, smem, st, a);
    else hipLaunchKernelGGL((gemv_w4_kernel<kNW, RGI, 1, D, false, false, 0, 2>), grid, block, smem, st, a);
    return hipGetLastError();
}

template <int D>
static hipError_t launch_mfma_silu(const GemvArgs& a, hipStream_t st) {
    const int nsets = a.N / 16;
    int nblk = mfma_grid(nsets), rs_cap = ceil_div(nsets, nblk);
    if (gemv_mfma_smem_bytes(kNW, 1, a.K, a.n_out, rs_cap) > 64 * 1024) { nblk = nsets; rs_cap = 1; }
    const size_t smem = gemv_mfma_smem_bytes(kNW, 1, a.K, a.n_out, rs_cap);
    const dim3 grid(nblk), block(kNW * 64);
    g_last_variant = "gemv_mfma_silu";
    if (a.n_out > 0) hipLaunchKernelGGL((gemv_w4_mfma_kernel<kNW, 1, D, true, false, 2, 0>), grid, block, smem, st, a, rs_cap);
    else hipLaunchKernelGGL((gemv_w4_mfma_kernel<kNW, 1, D, false, false, 2, 0>), grid, block, smem, st, a, rs_cap);
    return hipGetLastError();
}

hipError_t gemv_w4_silu_dispatch(const GemvArgs& a, hipStream_t st) {
    if (mfma_ok(a.N, a.K, a.G, a.n_out) && a.K <= 16384 && gemv_mfma_smem_bytes(kNW, 1, a.K, a.n_out) <= 64 * 1024)
        return mfma_depth(a.K) == 6 ? launch_mfma_silu<6>(a, st) : launch_mfma_silu<4>(a, st);
    const int rgs = a.N / 4;
    int rgi = 4;
    while (rgi > 1 && (rgs % rgi != 0 || rgs / rgi < 256)) rgi >>= 1;
    if (a.n_out > 0 && rgi == 1 && (rgs & 1)) return hipErrorInvalidValue;
    if (gemv_smem_bytes(kNW, rgi, 1, a.K, a.G, a.n_out) > 64 * 1024 || a.K > 4 * 8 * kNW * 64) return hipErrorInvalidValue;
    const bool deep = a.K > 6144;
    switch (rgi) {
        case 4: return deep ? launch_silu<4, 4>(a, st) : launch_silu<4, 2>(a, st);
        case 2: return deep ? launch_silu<2, 4>(a, st) : launch_silu<2, 2>(a, st);
        default: return deep ? launch_silu<1, 4>(a, st) : launch_silu<1, 2>(a, st);
    }
}

}  // namespace qeft
