// Kernel instantiation and dispatch of the v3 decode GEMV, shared by its two translation units: gemv_v3.hip (FL = true: launches
// that carry run-time flags -- the reference's entry points, the consumer-side norm, per-channel scales, several batch rows) and
// gemv_v3_plain.hip (FL = false: the decode engine's plain launches).  Two files so that hipcc builds the two halves in parallel.
#pragma once
#include "gemv_v3.h"

namespace qeft {

// Instantiations.  8 waves per block (the 16-wave form of round 2 lost to it on every launch kind once the step loop had its row
// sets at compile time, profiles/r03_gemv_lab.txt): ring depth 2 for every RSC, 4 for RSC <= 2, 6 for RSC >= 3.  4 waves per
// block with ring depth 4 for launches of more than 256 blocks (several blocks per CU: gate|up 10.47 vs 10.76 us).  12 waves, ring
// depth 2, for one-set launches with a long K (round 4: down_proj 7.44 -> 7.31 us, profiles/r04_gemv_lab.txt).
template <int NW, int D, bool OUTL, int BITS, int RSC, bool FL>
static hipError_t launch_dmr(const V3Args& a, int mode, int nblk, size_t smem, hipStream_t st) {
    auto go = [&](auto kern) -> hipError_t {
        if (smem > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kern, dim3(nblk), dim3(NW * 64), smem, st, V3_KERNEL_ARGS(a));
        return hipGetLastError();
    };
    if constexpr (NW == 8 && BITS == 4 && FL) {       // several batch rows (the reference's gemv entries, m = 2..7; 8..16 from the GEMM entries): plain launches
        if constexpr (D >= 4) {
            if (a.m > 1 && a.xg) return go(gemv_v3_kernel<8, D, OUTL, V3_MODE_PLAIN, 4, 2, RSC, true, true>);     // x read from global memory
        }
        if (a.m > 1 && a.xg) return hipErrorInvalidValue;
        if (a.m > 1) return go(gemv_v3_kernel<8, D, OUTL, V3_MODE_PLAIN, 4, 2, RSC, true>);
    }
    if (a.m > 1) return hipErrorInvalidValue;
    return mode == V3_MODE_PAIR ? go(gemv_v3_kernel<NW, D, OUTL, V3_MODE_PAIR, BITS, 1, RSC, FL>) : go(gemv_v3_kernel<NW, D, OUTL, V3_MODE_PLAIN, BITS, 1, RSC, FL>);
}

template <bool OUTL, int BITS, bool FL>
static hipError_t launch_d(const V3Args& a, int mode, int nblk, size_t smem, int depth, hipStream_t st) {
    if (a.nw == 4) {
        switch (a.rs_cap) {
            case 1: return launch_dmr<4, 4, OUTL, BITS, 1, FL>(a, mode, nblk, smem, st);
            case 2: return launch_dmr<4, 4, OUTL, BITS, 2, FL>(a, mode, nblk, smem, st);
            case 3: return launch_dmr<4, 4, OUTL, BITS, 3, FL>(a, mode, nblk, smem, st);
            case 4: return launch_dmr<4, 4, OUTL, BITS, 4, FL>(a, mode, nblk, smem, st);
        }
        return hipErrorInvalidValue;
    }
    if (a.nw == 12) {             // long one-set launches (down_proj: 85 full steps = 8 / 7 per wave instead of 11 / 10)
        if (a.rs_cap == 1) return launch_dmr<12, 2, OUTL, BITS, 1, FL>(a, mode, nblk, smem, st);
        return hipErrorInvalidValue;
    }
    switch (a.rs_cap) {           // row sets per block: a compile-time constant of the kernel
        case 1: return depth >= 4 ? launch_dmr<8, 4, OUTL, BITS, 1, FL>(a, mode, nblk, smem, st) : launch_dmr<8, 2, OUTL, BITS, 1, FL>(a, mode, nblk, smem, st);
        case 2: return depth >= 4 ? launch_dmr<8, 4, OUTL, BITS, 2, FL>(a, mode, nblk, smem, st) : launch_dmr<8, 2, OUTL, BITS, 2, FL>(a, mode, nblk, smem, st);
        case 3: return depth >= 4 ? launch_dmr<8, 6, OUTL, BITS, 3, FL>(a, mode, nblk, smem, st) : launch_dmr<8, 2, OUTL, BITS, 3, FL>(a, mode, nblk, smem, st);
        case 4: return depth >= 4 ? launch_dmr<8, 6, OUTL, BITS, 4, FL>(a, mode, nblk, smem, st) : launch_dmr<8, 2, OUTL, BITS, 4, FL>(a, mode, nblk, smem, st);
    }
    return hipErrorInvalidValue;
}

template <int BITS, bool FL>
static hipError_t launch_b(const V3Args& a, int mode, int nblk, size_t smem, int depth, hipStream_t st) {
    return a.g.n_out > 0 ? launch_d<true, BITS, FL>(a, mode, nblk, smem, depth, st) : launch_d<false, BITS, FL>(a, mode, nblk, smem, depth, st);
}

}  // namespace qeft
