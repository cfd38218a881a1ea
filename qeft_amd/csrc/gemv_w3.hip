// 3-bit EXTENSION of the packed-weight linear (BASELINE config 5).  The reference has no 3-bit pack or kernel
// (QuantLinear asserts bits == 4, qlinear.py:127); the layout is this build's own (oracle/qeft_oracle.py: pack_w3)
// and is shaped by the decode GEMV: one wave-wide 12 B/lane load = 16 rows x 128 k.
//   * decode (m <= 16): the MFMA GEMV of gemv_w4_mfma.h instantiated with BITS = 3 -- same prologue, same fused
//     transforms, 24 integer ops per 32 weights instead of 20, 25 % fewer weight bytes;
//   * everything else (GEMM forward, dX, dense dequant): qeft_expand_w3 rewrites the 3-bit stream into the 4-bit
//     checkpoint layout in a scratch buffer (reads 3/8 byte, writes 1/2 byte per weight: a few us per layer) and the
//     4-bit kernels run on that.
// Requirements: K % 128 == 0, n_out % 128 == 0, n_out < K, N % 16 == 0, group size 128 or K.
#include <cstdlib>

#include "gemv_w4_mfma.h"

namespace qeft {

namespace {
constexpr int kNW = 8;
constexpr size_t kMaxLds = 160 * 1024;

bool w3_ok(int N, int K, int G, int n_out) {
    return K % 128 == 0 && n_out % 128 == 0 && n_out < K && (G == 128 || G == K) && N % 16 == 0;
}
int grid_for(int nsets) {
    if (nsets < 512) return nsets;
    int k = (nsets + 384) / 768;
    return 256 * (k < 1 ? 1 : k);
}
int ceil_div(int a, int b) { return (a + b - 1) / b; }
// ring depth, as for 4 bits.  Measured (bench.py --bits 3, QEFT_W3_DEPTH): 4 / 6 / 8 -> 661 / 643 / 623 tokens/s: a
// 3-bit step is 768 bytes per wave instead of 1024, yet more steps in flight do not pay -- the launch is bound by
// its fixed costs and request latency, not by bytes.
int depth_for(int K) {
    static const char* env = getenv("QEFT_W3_DEPTH");   // lab
    if (env) { const int d = atoi(env); return d >= 8 ? 8 : d >= 6 ? 6 : 4; }
    return K > 6144 ? 6 : 4;
}

template <typename K>
hipError_t raise_lds(K kern, size_t smem) {
    if (smem <= 64 * 1024) return hipSuccess;
    return hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
}

// XT: 0 plain, 2 SiLU(gate)*up staging; batch 1
template <int D, int XT>
hipError_t launch_one(const GemvArgs& a, hipStream_t st) {
    const int nsets = a.N / 16;
    int nblk = grid_for(nsets), rs_cap = ceil_div(nsets, nblk);
    if (gemv_mfma_smem_bytes(kNW, 1, a.K, a.n_out, rs_cap) > 64 * 1024) { nblk = nsets; rs_cap = 1; }
    const size_t smem = gemv_mfma_smem_bytes(kNW, 1, a.K, a.n_out, rs_cap);
    if (smem > kMaxLds) return hipErrorInvalidValue;
    const dim3 grid(nblk), block(kNW * 64);
    g_last_variant = XT == 2 ? "gemv_w3_silu" : "gemv_w3";
    if (a.n_out > 0) {
        auto kern = gemv_w4_mfma_kernel<kNW, 1, D, true, false, XT, 0, 3>;
        if (hipError_t e = raise_lds(kern, smem)) return e;
        hipLaunchKernelGGL(kern, grid, block, smem, st, a, rs_cap);
    } else {
        auto kern = gemv_w4_mfma_kernel<kNW, 1, D, false, false, XT, 0, 3>;
        if (hipError_t e = raise_lds(kern, smem)) return e;
        hipLaunchKernelGGL(kern, grid, block, smem, st, a, rs_cap);
    }
    return hipGetLastError();
}

// 2..16 batch rows per pass: the run-time-M instantiation
hipError_t launch_rows(const GemvArgs& a, hipStream_t st) {
    const size_t smem = gemv_mfma_smem_bytes(kNW, a.m_rt, a.K, a.n_out);
    if (smem > kMaxLds) return hipErrorInvalidValue;
    const dim3 grid(a.N / 16), block(kNW * 64);
    g_last_variant = "gemv_w3_rows";
    if (a.n_out > 0) {
        auto kern = gemv_w4_mfma_kernel<kNW, 16, 4, true, false, 0, 0, 3>;
        if (hipError_t e = raise_lds(kern, smem)) return e;
        hipLaunchKernelGGL(kern, grid, block, smem, st, a, 1);
    } else {
        auto kern = gemv_w4_mfma_kernel<kNW, 16, 4, false, false, 0, 0, 3>;
        if (hipError_t e = raise_lds(kern, smem)) return e;
        hipLaunchKernelGGL(kern, grid, block, smem, st, a, 1);
    }
    return hipGetLastError();
}
}  // namespace

// y[m, N] = x[m, K] . W3^T (+ bias, + residual); any m >= 1 (rows are processed 16 per weight pass)
hipError_t gemv_w3_dispatch(const GemvArgs& a0, int m, hipStream_t st) {
    if (!w3_ok(a0.N, a0.K, a0.G, a0.n_out) || a0.ids != nullptr) return hipErrorNotSupported;
    if (m == 1) {
        GemvArgs a = a0;
        a.m_rt = 1;
        const int d = depth_for(a.K);
        return d == 8 ? launch_one<8, 0>(a, st) : d == 6 ? launch_one<6, 0>(a, st) : launch_one<4, 0>(a, st);
    }
    int mmax = 16;
    while (mmax > 1 && gemv_mfma_smem_bytes(kNW, mmax, a0.K, a0.n_out) > kMaxLds) --mmax;
    for (int m0 = 0; m0 < m; m0 += mmax) {
        GemvArgs a = a0;
        a.m_rt = (m - m0 < mmax) ? m - m0 : mmax;
        a.x = a0.x + (size_t)m0 * a0.K;
        a.y = a0.y + (size_t)m0 * a0.N;
        if (a0.residual) a.residual = a0.residual + (size_t)m0 * a0.N;
        if (hipError_t e = launch_rows(a, st)) return e;
    }
    return hipSuccess;
}

hipError_t gemv_w3_silu_dispatch(const GemvArgs& a, hipStream_t st) {
    if (!w3_ok(a.N, a.K, a.G, a.n_out) || a.K > 16384) return hipErrorNotSupported;
    const int d = depth_for(a.K);
    return d == 8 ? launch_one<8, 2>(a, st) : d == 6 ? launch_one<6, 2>(a, st) : launch_one<4, 2>(a, st);
}

template <int D>
static hipError_t launch_group3(const GemvGroupArgs& g, int nblocks, int rs_cap, hipStream_t st) {
    const size_t smem = gemv_mfma_smem_bytes(kNW, 1, g.K, g.n_out, rs_cap);
    const dim3 grid(nblocks), block(kNW * 64);
    g_last_variant = "gemv_w3_group";
    if (g.xt_aux) {
        if (g.n_out > 0) hipLaunchKernelGGL((gemv_w4_mfma_group_kernel<kNW, D, true, 1, 3>), grid, block, smem, st, g, rs_cap);
        else hipLaunchKernelGGL((gemv_w4_mfma_group_kernel<kNW, D, false, 1, 3>), grid, block, smem, st, g, rs_cap);
    } else {
        if (g.n_out > 0) hipLaunchKernelGGL((gemv_w4_mfma_group_kernel<kNW, D, true, 0, 3>), grid, block, smem, st, g, rs_cap);
        else hipLaunchKernelGGL((gemv_w4_mfma_group_kernel<kNW, D, false, 0, 3>), grid, block, smem, st, g, rs_cap);
    }
    return hipGetLastError();
}

hipError_t gemv_w3_group_dispatch(GemvGroupArgs g, int nparts, hipStream_t st) {
    int ntot = 0;
    for (int p = 0; p < nparts; ++p) {
        if (g.N[p] % 16 != 0) return hipErrorNotSupported;
        ntot += g.N[p];
    }
    if (!w3_ok(ntot, g.K, g.G, g.n_out) || g.K > 16384 || gemv_mfma_smem_bytes(kNW, 1, g.K, g.n_out) > 64 * 1024)
        return hipErrorNotSupported;
    const int nsets = ntot / 16;
    const int total_blocks = grid_for(nsets);
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
    const int d = depth_for(g.K);
    return d == 8 ? launch_group3<8>(g, acc, rs_cap, st) : d == 6 ? launch_group3<6>(g, acc, rs_cap, st)
                                                                     : launch_group3<4>(g, acc, rs_cap, st);
}

// 3-bit stream -> 4-bit checkpoint layout (qlinear.py:81-121 nibble order; the fp16 columns get dead zero nibbles).
// One thread per (row, 32-k chunk): 12 bytes in, 16 bytes out.
__global__ __launch_bounds__(256) void expand_w3_kernel(const uint32_t* __restrict__ q3, uint8_t* __restrict__ q4, int N,
                                                        int K, int n_out) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int chunks = K / 32;
    if (idx >= (size_t)N * chunks) return;
    const int n = (int)(idx / chunks), c = (int)(idx % chunks);
    const int k0 = c * 32, kq = K - n_out, nfull = kq / 128;
    u32x4 out = {0u, 0u, 0u, 0u};
    if (k0 < kq) {
        const int step = k0 / 128, lane = ((k0 >> 5) & 3) * 16 + (n & 15);
        const uint32_t* w = q3 + (((size_t)(n >> 4) * nfull + step) * 64 + lane) * 3;
        const uint32_t w0 = w[0], w1 = w[1], w2 = w[2];
        uint32_t ext[16];   // pair e: low half = value of k = 2e, high half = value of k = 2e + 1
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const uint32_t v = i == 0 ? w0 : i == 1 ? w1 : w2;
#pragma unroll
            for (int t = 0; t < 5; ++t) ext[5 * i + t] = (v >> (3 * t)) & 0x00070007u;
        }
        ext[15] = ((w0 >> 15) & 0x00010001u) | ((w1 >> 14) & 0x00020002u) | ((w2 >> 13) & 0x00040004u);
        // 4-bit word wd holds pairs e = 4 j + wd (j = 0..3) at bits 4 j (low half) and 16 + 4 j (high half)
#pragma unroll
        for (int wd = 0; wd < 4; ++wd)
#pragma unroll
            for (int j = 0; j < 4; ++j) out[wd] |= ext[4 * j + wd] << (4 * j);
    }
    uint8_t* dst = q4 + (size_t)(n >> 2) * K * 2 + (size_t)(k0 >> 6) * 128 + (n & 3) * 32 + ((k0 >> 5) & 1) * 16;
    *(u32x4*)dst = out;
}

hipError_t expand_w3_launch(const void* q3, void* q4, int N, int K, int n_out, hipStream_t st) {
    if (!w3_ok(N, K, 128, n_out)) return hipErrorNotSupported;
    const size_t total = (size_t)N * (K / 32);
    hipLaunchKernelGGL(expand_w3_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, st, (const uint32_t*)q3,
                       (uint8_t*)q4, N, K, n_out);
    return hipGetLastError();
}

}  // namespace qeft
