// Decode GEMV, MFMA formulation (the production path for K % 128 == 0, n_out % 128 == 0, G == 128 or G == K).
//
// Measured on MI355X (tools/gemv_lab.hip): with the VALU formulation (gemv_w4_kernel.h) the kernel is co-limited
// by vector issue: ~50 VALU instructions per 16 weight bytes per lane is ~2 us of pure issue time on a 26 MB layer
// that HBM can stream in ~4 us.  The matrix pipe is idle in a GEMV, and it takes exactly the fragment shape the
// checkpoint already has:
//
//   v_mfma_f32_16x16x32_f16:  B operand lane L = (n = L & 15, k-group = L >> 4) holds 8 k-values of ONE column n
//   checkpoint:               one u32 word holds 8 k-values of ONE weight row n
//
// One wave-wide 16 B/lane load = 16 rows (4 row-groups) x 128 k.  Lane L loads the 16 bytes of row (L & 15), 32-k
// chunk (L >> 4); its 4 words are the B fragments of 4 MFMAs, which together contract the step's 128 k for all 16
// rows and up to 16 batch rows at once — no cross-lane reduction, and batch 1..7 cost the same.
// The A operand is x, staged in LDS in natural order: MFMA j of a step contracts the 8 consecutive k of 16-byte slot j
// of the lane's chunk (one ds_read_b128), and the matching B fragment is pair j of each of the 4 nibble words --
// a pure register naming of the v_and_or_b32 results.
//
// Arithmetic (identical to gemv_w4_kernel.h): nibbles become 1024+q / 1024+16q with one v_and_or_b32 each, x is
// pre-multiplied by 1/16 where it meets high nibbles, and per step (= one quantisation group of 128 k)
//   acc[m][n] += s[n] * (P[m][n] - A[m]) + sz[n] * B[m],   A = 1024 * sum x', B = sum x over the step's 128 k.
// 5 integer ops per word + 1 MFMA + 1 LDS read per 8 weights per lane.
#pragma once
#include <type_traits>

#include "gemv_w4_kernel.h"

namespace qeft {

__host__ __device__ constexpr size_t gemv_mfma_smem_bytes(int NW, int M, int K, int n_out, int RS = 1) {
    return (size_t)RS * NW * 16 * M * 4 +                              // red   [RS][NW][M][16] f32
           (n_out > 0 ? (size_t)RS * 16 * gemv_slab_stride(n_out) * 2 : 0) +  // slab  [RS*16][n_out+8] f16
           (size_t)RS * (K / 128) * 16 * 4 +                            // szl   [RS][K/128][16] (s | sz << 16)
           ((size_t)M * (K / 128) * 8 + 15) / 16 * 16 +                 // corr  [M][K/128] (A, B) f32
           (size_t)M * K * 2;                                           // xs    [M][K] f16 (natural order, prescaled)
}

typedef _Float16 h8v __attribute__((ext_vector_type(8)));

// Sum over the 16 lanes of a DPP row with 4 VALU DPP adds (quad swaps, half-row mirror, row mirror); every lane ends
// with the row's total.  ds_bpermute-based __shfl_xor costs ~100 cycles per step and sits on the prologue's critical path.
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
    return v;
}

// A block owns RS consecutive 16-row sets (run time).  The block prologue (x staging, optional RMSNorm / SiLU,
// correction sums) is paid once per block, so wide layers are launched with fewer, longer-lived blocks (a few per
// CU): one prologue then feeds RS x more weight bytes, and a wave walks its (row set, step) pairs as ONE sequence so
// the register ring never drains between row sets.
template <int NW, int M, int D, bool OUTL, bool XG, int XT = 0, int ABL = 0, int BITS = 4>
__device__ __forceinline__ void gemv_w4_mfma_body(const GemvArgs& a, const int set0, const int RS, const int rs_cap) {
    static_assert(BITS == 4 || BITS == 3, "4-bit checkpoint layout or the 3-bit extension layout (oracle: pack_w3)");
    // this block owns the 16-row sets [set0, set0 + RS) of the layer; LDS is carved for rs_cap sets (launch constant)
    static_assert(XT == 0 || (M == 1 && !XG), "x transforms are for the batch-1 decode engine");
    constexpr int kWaves = NW, kBlock = NW * 64;
    // M == 16 is the "any batch 4..16" instantiation used by the small-M GEMM route: the MFMA contracts 16 batch rows
    // anyway, only LDS sizes and the stored rows depend on the real count a.m_rt.
    const int MR = (M == 16) ? a.m_rt : M;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ float ssq[NW];                 // XT == 1: per-wave sums of squares of x, read in the epilogue
    float* red = (float*)smem;                                                    // [rs_cap][NW][M][16]
    f16* slab = (f16*)(smem + rs_cap * NW * 16 * MR * 4);                          // [rs_cap*16][n_out + 8]
    const int slab_stride = gemv_slab_stride(a.n_out);
    uint32_t* szl = (uint32_t*)(slab + (OUTL ? rs_cap * 16 * slab_stride : 0));   // [rs_cap][K/128][16]
    const int nsteps = a.K / 128;
    float* corr = (float*)(szl + rs_cap * nsteps * 16);
    uint32_t* xs32 = (uint32_t*)((uint8_t*)corr + ((size_t)MR * nsteps * 8 + 15) / 16 * 16);   // x' as dwords (k-pairs)
    const f16* xs = (const f16*)xs32;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: step bookkeeping stays scalar
    const int nl = lane & 15, kc = lane >> 4;            // row within the block / 32-k chunk within the step
    const int rg0 = set0 * 4;
    const int row0 = rg0 * 4;
    const int kq = a.K - (OUTL ? a.n_out : 0);
    const int nfull = kq / 128;                          // INT4 steps; steps [nfull, nsteps) are the fp16 outlier slice
    const int nsw = (nfull - wave + kWaves - 1) / kWaves;
    const bool per_channel = a.gshift == 31;
    // diagnostic stamps (lab builds only): [block][8] = realtime at entry, shader clock at the phase boundaries
    unsigned long long stamp[10];
    auto mark = [&](int i) {
        if (ABL & 16) { __builtin_amdgcn_sched_barrier(0); stamp[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
    };
    unsigned long long rt0 = 0;
    if (ABL & 16) rt0 = __builtin_amdgcn_s_memrealtime();
    mark(0);

    // ---- 1. oldest loads: x, the transform operand, the outlier slab, the scales
    // Every load here is UNCONDITIONAL with a clamped / substituted address and nothing consumes a loaded value before
    // the weight ring has been issued: a load inside a branch (even a block-uniform one) ends in register copies at
    // the join, for which the compiler waits on the spot -- one exposed memory round trip in front of the weight
    // stream.  Passes beyond the data re-read the last vector (an L1 hit).
    const int xtotal = MR * a.K;
    const int xvecs = xtotal / 8;
    u32x4 xst[XG ? 1 : 4];
    u32x4 ast[XT ? 4 : 1];
    (void)xst;
    (void)ast;
    if (!XG) {
        {
            const size_t e = (size_t)min(tid, xvecs - 1) * 8;
            xst[0] = *(const u32x4*)(a.x + e);
            if (XT) ast[0] = *(const u32x4*)(a.xt_aux + e);
        }
        // rows longer than one pass (K > 4096 at batch 1): ONE block-uniform region whose registers are undefined on
        // the other path (nothing to merge at the join, hence no wait there -- checked with tools/isa_waits.py)
        if (xvecs > kBlock) {
#pragma unroll
            for (int p = 1; p < 4; ++p) {
                const size_t e = (size_t)min(p * kBlock + tid, xvecs - 1) * 8;
                xst[p] = *(const u32x4*)(a.x + e);
                if (XT) ast[p] = *(const u32x4*)(a.xt_aux + e);
            }
        }
    }
    const int slab_vecs = OUTL ? RS * 8 * (2 * a.n_out) / 8 : 0;      // 8 interleaved rows of 2*n_out halves per row set
    const bool slab_il = OUTL && !a.ow_plain;
    const f16* osrc = slab_il ? a.ow_il + (size_t)(rg0 >> 1) * 4 * (2 * a.n_out) : a.x;
    u32x4 ost[2];
#pragma unroll
    for (int p = 0; p < 2; ++p)
        ost[p] = *(const u32x4*)(osrc + (size_t)(slab_il ? min(p * kBlock + tid, slab_vecs - 1) : 0) * 8);
    const int ngroups = per_channel ? 1 : nsteps;
    const int szn = ngroups * 8 * RS;                                  // dword pairs of two adjacent rows
    uint32_t sst[2], zst[2];
    u32x4 szv[2];
    const int szvecs = RS * ngroups * 4;                               // 16-byte pieces of the block's shadow scales
    const bool shadow = a.sz_blk != nullptr;
    const uint32_t* szsrc = shadow ? a.sz_blk + (size_t)set0 * ngroups * 16 : (const uint32_t*)a.x;
    // contiguous [RS][K/G][16] dwords for this block: coalesced 16-byte loads
#pragma unroll
    for (int p = 0; p < 2; ++p)
        szv[p] = *(const u32x4*)(szsrc + (size_t)(shadow ? min(p * kBlock + tid, szvecs - 1) : 0) * 4);
    if (!shadow) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int i = min(p * kBlock + tid, szn - 1);
            const uint32_t so = (uint32_t)(i / (8 * RS)) * (uint32_t)a.N + row0 + (i % (8 * RS)) * 2;
            sst[p] = (ABL & 1) ? 0x1c001c00u : *(const uint32_t*)(a.scales + so);
            zst[p] = (ABL & 1) ? 0xa000a000u : *(const uint32_t*)(a.zeros + so);
        }
    }

    // ---- 2. weight stream: ring of D steps per wave (steps wave, wave+NW, ...), branch-free, oldest first
    // The wave's work is the sequence t = 0 .. RS*nsw-1 of (row set rs = t / nsw, step s = wave + NW * (t % nsw)).
    // BITS == 3: 12 bytes per lane (three words = the 32 3-bit weights of (row nl, chunk kc)), a step of one row set
    // is 768 contiguous bytes, a row set (K - n_out) / 128 steps; the fp16 columns have no fields (oracle: pack_w3).
    typedef typename std::conditional<BITS == 3, u32x3_u, u32x4>::type ring_t;
    ring_t ring[D];
    const uint32_t step_bytes = BITS == 3 ? 768u : 256u;
    // address = wave-uniform base (scalar registers) + a 32-bit per-lane offset: the load takes the `saddr + voffset`
    // form and a step costs two scalar adds instead of a 64-bit vector multiply-add
    const uint8_t* wbase = BITS == 3 ? a.qw + (size_t)set0 * nfull * 768u : a.qw + (size_t)rg0 * a.K * 2;
    const uint32_t lane_off = BITS == 3 ? (uint32_t)lane * 12u
                                        : (uint32_t)(nl >> 2) * (uint32_t)a.K * 2u + (kc >> 1) * 128 + (nl & 3) * 32 + (kc & 1) * 16;
    const uint32_t rs_bytes = BITS == 3 ? (uint32_t)nfull * 768u : (uint32_t)a.K * 8u;   // bytes per row set
    const uint32_t last_off = BITS == 3 ? (uint32_t)max(nfull - 1, 0) * 768u : (uint32_t)a.K * 2 - 256u;
    int p_rs = 0, p_i = 0;                                             // (row set, index) of the next step to issue
    auto issue = [&](ring_t& b) {
        // past the end (or a wave without ring steps): harmless re-read of a valid address inside the row set
        const int irs = min(p_rs, RS - 1);
        const uint32_t woff = min((uint32_t)(wave + p_i * kWaves) * step_bytes, last_off);
        const uint8_t* sp = wbase + ((size_t)irs * rs_bytes + woff);     // wave-uniform
        b = __builtin_nontemporal_load((const ring_t*)(sp + lane_off));
        if (++p_i >= nsw) { p_i = 0; ++p_rs; }
    };
#pragma unroll
    for (int d = 0; d < D; ++d) {
        issue(ring[d]);
        __builtin_amdgcn_sched_barrier(0);
    }

    mark(1);
    // ---- 2b. optional x transform on the register-held vectors
    if (XT == 1) {
        // RMSNorm, with the 1/rms factor DEFERRED: y = W . (x * rs * gamma) = rs * (W . (x * gamma)).  x * gamma is
        // staged right away (one fp16 rounding per element, as the unfused norm has), the sum of squares goes to LDS
        // beside it and is only read in the epilogue -- no block-wide reduction sits between the x loads and the
        // staging barrier any more (it cost ~0.8 us per launch).
        float ss = 0.f;
#pragma unroll
        for (int p = 0; p < 4; ++p)
            if (p == 0 || p * kBlock < xvecs) {            // block-uniform and registers only: whole passes are skipped
                float sp = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const h2 t = as_h2(xst[p][j]);
                    sp += (float)t[0] * (float)t[0] + (float)t[1] * (float)t[1];
                }
                ss += (p * kBlock + tid < xvecs) ? sp : 0.f;   // the tail of a partial pass re-read the last vector
            }
        mark(6);
        ss = row16_sum(ss);
        ss += __shfl_xor(ss, 16);
        ss += __shfl_xor(ss, 32);
        if (lane == 0) ssq[wave] = ss;
#pragma unroll
        for (int p = 0; p < 4; ++p)
            if (p == 0 || p * kBlock < xvecs) {            // block-uniform, registers only
#pragma unroll
                for (int j = 0; j < 4; ++j) xst[p][j] = as_u32(as_h2(xst[p][j]) * as_h2(ast[p][j]));
            }
    } else if (XT == 2) {
#pragma unroll
        for (int p = 0; p < 4; ++p)
            if (p == 0 || p * kBlock < xvecs) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const h2 t = as_h2(xst[p][j]), u = as_h2(ast[p][j]);
                    xst[p][j] = as_u32(h2{mul_f32_to_f16(silu_f32((float)t[0]), (float)u[0]), mul_f32_to_f16(silu_f32((float)t[1]), (float)u[1])});
                }
            }
    }

    mark(7);
    // ---- 3. stage scales, x' (+ per-step sums) and the outlier slab into LDS
    if (a.sz_blk) {
        // shadow is [rs][group][16] dwords, LDS [rs][nsteps][16]: the same thing unless the layer is per-channel
        for (int v = tid; v < szvecs; v += kBlock) {
            const u32x4 t = (v == tid) ? szv[0] : (v == tid + kBlock) ? szv[1] : *(const u32x4*)(szsrc + (size_t)v * 4);
            const int dst = per_channel ? ((v >> 2) * nsteps * 16 + (v & 3) * 4) : v * 4;
            *(u32x4*)(szl + dst) = t;
        }
    } else
    for (int i = tid, p = 0; i < szn; i += kBlock, ++p) {
        uint32_t sv = sst[0], zv = zst[0];
        if (p == 1) { sv = sst[1]; zv = zst[1]; }
        const int grp = i / (8 * RS), pr = i % (8 * RS);
        if (p >= 2) {
            const uint32_t so = (uint32_t)grp * (uint32_t)a.N + row0 + pr * 2;
            sv = *(const uint32_t*)(a.scales + so);
            zv = *(const uint32_t*)(a.zeros + so);
        }
        // row set pr/8, rows (pr%8)*2, +1 of it
        *(u32x2*)(szl + ((pr >> 3) * nsteps + grp) * 16 + (pr & 7) * 2) =
            u32x2{__builtin_amdgcn_perm(zv, sv, 0x05040100u), __builtin_amdgcn_perm(zv, sv, 0x07060302u)};
    }
    {
        const h2 k16th = {(f16)0.0625f, (f16)0.0625f};
        for (int v = tid, p = 0; v < xvecs; v += kBlock, ++p) {
            u32x4 xv;
            const int e = v * 8;
            const int bm = (M == 1) ? 0 : e / a.K, k = e - bm * a.K;
            if (XG) {
                f16 t[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) t[j] = a.x[(size_t)bm * a.K + a.ids[k + j]];   // qlinear.py:275
#pragma unroll
                for (int j = 0; j < 4; ++j) xv[j] = as_u32(h2{t[2 * j], t[2 * j + 1]});
            } else {
                xv = xst[0];
                if (p == 1) xv = xst[1];
                if (p == 2) xv = xst[2];
                if (p == 3) xv = xst[3];
                if (p >= 4) xv = *(const u32x4*)(a.x + (size_t)e);
            }
            float bsum = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) bsum += (float)as_h2(xv[j])[0] + (float)as_h2(xv[j])[1];
            float asum = bsum;
            if (BITS == 4) {
                const int q = (k >> 3) & 3;                       // quarter of the 32-k chunk
                const bool hi = (q & 1) && (k < kq);              // k%32 in [8,16) or [24,32): meets the high nibbles
                if (hi) {
                    asum = 0.f;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const h2 t = as_h2(xv[j]) * k16th;
                        xv[j] = as_u32(t);
                        asum += (float)t[0] + (float)t[1];
                    }
                }
            } else if (k < kq) {
                // 3-bit: pair e = (k % 32) / 2 sits at bit 3 * (e % 5) (mod 9) of its half-word -> 1024 + q * 2^sh
                asum = 0.f;
                const int e0 = (k >> 1) & 15;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int e = e0 + j;
                    const int sh = (e == 15) ? 0 : ((e % 5) % 3) * 3;
                    const f16 f = (f16)(1.0f / (float)(1 << sh));
                    const h2 t = as_h2(xv[j]) * h2{f, f};
                    xv[j] = as_u32(t);
                    asum += (float)t[0] + (float)t[1];
                }
            }
            *(u32x4*)(xs32 + (size_t)(e >> 1)) = xv;      // natural order: slot q of the chunk
            // the 16 vectors of a 128-k step sit in the 16 lanes of one DPP row
            asum = row16_sum(asum);
            bsum = row16_sum(bsum);
            if ((v & 15) == 0) {
                corr[(e >> 7) * 2] = 1024.f * asum;
                corr[(e >> 7) * 2 + 1] = bsum;
            }
        }
    }
    if (OUTL && a.ow_plain) {
        // plain oweight [N, n_out] (the GEMM entry's operand): a 16-byte piece = 8 columns (4 dwords) of one row
        uint32_t* slab32 = (uint32_t*)slab;
        const int sstr = slab_stride / 2;
        const int per_row = a.n_out / 8;
        for (int v = tid; v < RS * 16 * per_row; v += kBlock) {
            const int lrow = v / per_row, piece = v % per_row;
            const u32x4 ov = *(const u32x4*)(a.ow_plain + (size_t)(row0 + lrow) * a.n_out + piece * 8);
            *(u32x4*)(slab32 + lrow * sstr + piece * 4) = ov;
        }
    } else if (OUTL) {
        // pack_oweight layout (qlinear.py:70-79); a 16-byte piece = 4 columns (2 dwords) of rows blk*8+rr and +4.
        uint32_t* slab32 = (uint32_t*)slab;
        const int sstr = slab_stride / 2;   // dwords per slab row
        for (int v = tid; v < slab_vecs; v += kBlock) {
            const u32x4 ov = (v == tid) ? ost[0] : (v == tid + kBlock) ? ost[1] : *(const u32x4*)(osrc + (size_t)v * 8);
            const int per_row = (2 * a.n_out) / 8;
            int ir, piece;
            if (a.n_out == 128) { ir = v >> 5; piece = v & 31; }       // the checkpoint's r: no integer division
            else { ir = v / per_row; piece = v % per_row; }
            const int d0 = (piece >> 3) * 16 + (piece & 7) * 2;          // natural dwords d0, d0+1 of the row
            const u32x2 lo = {__builtin_amdgcn_perm(ov[1], ov[0], 0x05040100u), __builtin_amdgcn_perm(ov[3], ov[2], 0x05040100u)};
            const u32x2 hi = {__builtin_amdgcn_perm(ov[1], ov[0], 0x07060302u), __builtin_amdgcn_perm(ov[3], ov[2], 0x07060302u)};
            const int lrow = (ir >> 2) * 8 + (ir & 3);
            *(u32x2*)(slab32 + lrow * sstr + d0) = lo;
            *(u32x2*)(slab32 + (lrow + 4) * sstr + d0) = hi;
        }
    }
    mark(8);
    __syncthreads();
    if (ABL & 32) {   // lab experiment: start the weight stream only after the staging loads have returned
#pragma unroll
        for (int d = 0; d < D; ++d) {
            issue(ring[d]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    mark(2);

    // ---- 4. steps
    constexpr int NACC = M < 4 ? M : 4;   // batch rows a lane can own: m = 4*kc + j (rows >= M are never stored)
    float acc[NACC];                      // acc[j]: batch row m = 4*kc + j, weight row nl of the current row set
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = 0.f;
    uint32_t MAGIC = 0x64006400u;
    asm volatile("" : "+v"(MAGIC));
    const int am = min(nl, MR - 1);         // A-operand row of this lane (rows >= M replicate row M-1, never stored)
    const f16* xa = xs + (size_t)am * a.K + kc * 32;

    // fp16 outlier steps [nfull, nsteps) of row set rs: B fragments straight from the LDS slab (same slot order)
    auto outlier_steps = [&](int rs) {
        if (!OUTL) return;
        for (int s = nfull + ((wave - nfull) % kWaves + kWaves) % kWaves; s < nsteps; s += kWaves) {
            const h8v* px = (const h8v*)(xa + (size_t)s * 128);
            const h8v* pw = (const h8v*)(slab + (size_t)(rs * 16 + nl) * slab_stride + (s - nfull) * 128 + kc * 32);
            f32x4 P = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < 4; ++w) P = __builtin_amdgcn_mfma_f32_16x16x32_f16(px[w], pw[w], P, 0, 0, 0);
            if (!(ABL & 4)) {
#pragma unroll
                for (int j = 0; j < NACC; ++j) acc[j] += P[j];
            }
        }
    };
    auto flush = [&](int rs) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) {
            const int m = 4 * kc + j;
            if (m < MR) red[((rs * kWaves + wave) * MR + m) * 16 + nl] = acc[j];
            acc[j] = 0.f;
        }
    };

    // LDS operands of a step (A fragments, the row's scale word, the step's correction sums) are fetched one step
    // ahead into two alternating register sets, so no step waits for LDS after its weights arrived.
    struct StepOps {
        h8v x[4];
        uint32_t szw;
        float2 ab[NACC];
    };
    int c_rs = 0, c_i = 0;                 // (row set, index) of the next step to consume
    auto fetch_ops = [&](StepOps& o) {
        const int rs = min(c_rs, RS - 1);  // past the end: harmless in-range read
        const int s = wave + c_i * kWaves;
        const h8v* px = (const h8v*)(xa + (size_t)s * 128);
#pragma unroll
        for (int w = 0; w < 4; ++w) o.x[w] = px[w];
        o.szw = szl[(rs * nsteps + (per_channel ? 0 : s)) * 16 + nl];
#pragma unroll
        for (int j = 0; j < NACC; ++j) {                  // batch row m = 4*kc + j; rows >= M are never stored
            const int m = min(4 * kc + j, MR - 1);
            o.ab[j] = *(const float2*)(corr + ((size_t)m * nsteps + s) * 2);
        }
    };
    auto consume = [&](const ring_t& wv, const StepOps& cur, StepOps& nxt) {
        if (c_i == 0) outlier_steps(c_rs);
        const int rs_now = c_rs;
        const bool last_of_set = c_i + 1 >= nsw;
        if (last_of_set) { c_i = 0; ++c_rs; } else { ++c_i; }
        fetch_ops(nxt);                                   // LDS reads for the NEXT step go out before this one's math
        if (ABL & 4) {
            acc[0] += __builtin_bit_cast(float, wv[0] ^ wv[1] ^ wv[2] ^ wv[3]);
        } else {
            f32x4 P0 = {0.f, 0.f, 0.f, 0.f}, P1 = {0.f, 0.f, 0.f, 0.f};   // two chains: MFMA latency is exposed at 2 waves/SIMD
            // fragment j = pair j of every word w: k = 8j + 2w, +1 (w = 0..3) -- the 8 consecutive k of x slot j
            u32x4 bf[4];
            if (BITS == 4) {
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const uint32_t v = wv[w], t = v >> 8;
                    bf[0][w] = (v & 0x000f000fu) | MAGIC;    // 1024 + q
                    bf[1][w] = (v & 0x00f000f0u) | MAGIC;    // 1024 + 16 q   (x' = x / 16 on these k)
                    bf[2][w] = (t & 0x000f000fu) | MAGIC;
                    bf[3][w] = (t & 0x00f000f0u) | MAGIC;
                }
            } else {
                // pair e = 4 j + w (k = 2e, 2e+1 of the chunk): word e / 5, field e % 5; pair 15 from bits 15 / 31
                uint32_t ext[16];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const uint32_t v = wv[i], t = v >> 9;
                    ext[5 * i + 0] = (v & 0x00070007u) | MAGIC;   // 1024 + q
                    ext[5 * i + 1] = (v & 0x00380038u) | MAGIC;   // 1024 + 8 q    (x' = x / 8)
                    ext[5 * i + 2] = (v & 0x01c001c0u) | MAGIC;   // 1024 + 64 q   (x' = x / 64)
                    ext[5 * i + 3] = (t & 0x00070007u) | MAGIC;
                    ext[5 * i + 4] = (t & 0x00380038u) | MAGIC;
                }
                ext[15] = ((wv[0] >> 15) & 0x00010001u) | MAGIC;
                ext[15] |= (wv[1] >> 14) & 0x00020002u;
                ext[15] |= (wv[2] >> 13) & 0x00040004u;
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int w = 0; w < 4; ++w) bf[j][w] = ext[4 * j + w];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j & 1) P1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur.x[j], __builtin_bit_cast(h8v, bf[j]), P1, 0, 0, 0);
                else P0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur.x[j], __builtin_bit_cast(h8v, bf[j]), P0, 0, 0, 0);
            }
            const h2 szp = as_h2(cur.szw);
            const float sf = (float)szp[0], zf = (float)szp[1];
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] += sf * ((P0[j] + P1[j]) - cur.ab[j].x) + zf * cur.ab[j].y;
        }
        if (last_of_set) flush(rs_now);
    };

    if (nsw == 0) {
        // K has no full INT4 step for this wave: only (possibly) outlier steps
        for (int rs = 0; rs < RS; ++rs) {
            outlier_steps(rs);
            flush(rs);
        }
    } else {
        static_assert(D % 2 == 0, "ring depth must be even: LDS operand sets alternate per slot");
        const int total = RS * nsw;
        const int nrounds = (total + D - 1) / D;
        StepOps opA, opB;
        fetch_ops(opA);
        for (int rd = 0; rd + 1 < nrounds; ++rd) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                if (d & 1) consume(ring[d], opB, opA);
                else consume(ring[d], opA, opB);
                issue(ring[d]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if ((nrounds - 1) * D + d < total) {
                if (d & 1) consume(ring[d], opB, opA);
                else consume(ring[d], opA, opB);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    mark(3);
    // ---- 5. combine the waves
    __syncthreads();
    mark(4);
    float rs_norm = 1.f;
    if (XT == 1) {
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) tot += ssq[w];
        rs_norm = rsqrtf(tot / (float)a.K + a.xt_eps);
    }
    for (int o = tid; o < RS * 16 * MR; o += kBlock) {
        const int rs = o / (16 * MR), m = (o / 16) % MR, n = o & 15;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) v += red[((rs * kWaves + w) * MR + m) * 16 + n];
        const int orow = row0 + rs * 16 + n;
        if (XT == 1) v *= rs_norm;
        if (a.bias) v += (float)a.bias[orow];
        if (a.residual) v += (float)a.residual[(size_t)m * a.N + orow];
        a.y[(size_t)m * a.N + orow] = (f16)v;
    }
    if ((ABL & 16) && a.dbg && (tid & 63) == 0) {
        mark(5);
        unsigned long long* d = a.dbg + ((size_t)blockIdx.x * kWaves + wave) * 8;
        d[0] = rt0;
        d[1] = __builtin_amdgcn_s_memrealtime();
#pragma unroll
        for (int i = 0; i < 6; ++i) d[2 + i] = stamp[i];
        if (a.dbg2) {
            unsigned long long* e = a.dbg2 + ((size_t)blockIdx.x * kWaves + wave) * 4;
            e[0] = stamp[6] - stamp[1];   // loads issued -> x usable (sum of squares computed)
            e[1] = stamp[7] - stamp[6];   // block reduction + normalisation
            e[2] = stamp[8] - stamp[7];   // staging writes (scales, x', sums, slab)
            e[3] = stamp[2] - stamp[8];   // barrier
        }
    }
}

// Blocks are dealt round-robin over the 8 XCDs (b and b+8 share an L2).  Neighbouring row blocks share the 128-byte
// lines of scales / scaled_zeros and of oweight_interleaved, so give each XCD a contiguous range of row blocks.
// Speed only; any placement is correct.
__device__ __forceinline__ int xcd_contiguous_block(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// Row sets [0, nsets) dealt to nblk blocks as evenly as possible: the first (nsets % nblk) blocks get one more.
__device__ __forceinline__ void block_sets(int b, int nblk, int nsets, int& set0, int& cnt) {
    const int q = nsets / nblk, r = nsets % nblk;
    set0 = b * q + min(b, r);
    cnt = q + (b < r ? 1 : 0);
}

template <int NW, int M, int D, bool OUTL, bool XG, int XT = 0, int ABL = 0, int BITS = 4>
__global__ __launch_bounds__(NW * 64) void gemv_w4_mfma_kernel(GemvArgs a, int rs_cap) {
    const int b = (ABL & 8) ? (int)blockIdx.x : xcd_contiguous_block(blockIdx.x, gridDim.x);
    int set0, cnt;
    block_sets(b, gridDim.x, a.N / 16, set0, cnt);
    gemv_w4_mfma_body<NW, M, D, OUTL, XG, XT, ABL, BITS>(a, set0, cnt, rs_cap);
}

// Several linears that share the same input (q/k/v, gate/up) in ONE launch, batch 1: blocks [blk_end[p-1], blk_end[p])
// work on part p and split its row sets evenly.
template <int NW, int D, bool OUTL, int XT = 0, int BITS = 4>
__global__ __launch_bounds__(NW * 64) void gemv_w4_mfma_group_kernel(GemvGroupArgs g, int rs_cap) {
    const int gb = xcd_contiguous_block(blockIdx.x, gridDim.x);
    int p = 0, blk = gb, nb = g.blk_end[0];
    if (gb >= g.blk_end[0]) {
        p = 1; blk = gb - g.blk_end[0]; nb = g.blk_end[1] - g.blk_end[0];
        if (gb >= g.blk_end[1]) { p = 2; blk = gb - g.blk_end[1]; nb = g.blk_end[2] - g.blk_end[1]; }
    }
    GemvArgs a;
    a.x = g.x;
    a.qw = g.qw[p];
    a.scales = g.scales[p];
    a.zeros = g.zeros[p];
    a.ow_il = g.ow_il[p];
    a.bias = g.bias[p];
    a.ids = nullptr;
    a.residual = nullptr;
    a.y = g.y[p];
    a.N = g.N[p];
    a.K = g.K;
    a.G = g.G;
    a.n_out = g.n_out;
    a.gshift = g.gshift;
    a.xt_aux = g.xt_aux;
    a.xt_eps = g.xt_eps;
    a.sz_blk = g.sz_blk[p];
    a.dbg = nullptr;
    a.dbg2 = nullptr;
    a.ow_plain = nullptr;
    a.m_rt = 1;
    int set0, cnt;
    block_sets(blk, nb, a.N / 16, set0, cnt);
    gemv_w4_mfma_body<NW, 1, D, OUTL, false, XT, 0, BITS>(a, set0, cnt, rs_cap);
}

}  // namespace qeft
