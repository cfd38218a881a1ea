// Persistent decode chain (round 4): several dependent batch-1 GEMVs of a decoder layer in ONE launch -- o_proj -> gate|up ->
// down_proj -> q|k|v of the next layer -- so that the weight stream never stops at a dependency: the weights of phase p + 1 do
// not depend on x, and every compute wave keeps a ring of LDS-DMA loads (global_load_lds, 1 KB each, non-temporal) running
// ACROSS the phase boundary while the activation vector makes its all-to-all trip through memory.
//
//   grid   = one block per CU (nblk <= CU count: every block must be resident), NW waves
//   phase  = one GEMV (gemv_v3.h's arithmetic: raw x from LDS as the MFMA A operand, 1024 + q nibble trick with the bias removed
//            by -1024 MFMAs, fp32 accumulation, step-major over the block's row sets), its epilogue on wave 0
//   edge   = the phase's outputs published as 8-byte {2 x fp16, tag} granules (ONE sc1 store each, the data is the flag:
//            MI355X guide, Guideline 16 recipe R2); every wave of every block sweeps its share of the vector into LDS with
//            16-byte sc1 loads, re-reading until every tag carries this edge's epoch; bounded spins with a give-up code
//            (status word), never a hang
//   ring   = per compute wave D slots of 1 KB in LDS; the issue cursor walks (phase, step, row set) D - 1 loads ahead of the
//            consume cursor and crosses phase boundaries on its own
//
// Reference counterpart of every phase: gemv_kernel_qeft (qeft/kernel/quantization_new/gemv/gemv_cuda_qeft.cu:75-222) behind
// QuantLinear.forward_outlier (qeft/qlinear.py:244-271); the chain itself has no counterpart (the reference launches one
// kernel per linear).
#pragma once
#include "gemv_v3.h"

namespace qeft {

constexpr int CH_NW_MAX = 16;                  // waves per block: 8 or 16 (wave w owns the 128-k steps w, w + NW, ...)
constexpr int CH_MAX_PHASES = 4;
constexpr int CH_MAX_RSC = 8;                  // row sets per block and phase
constexpr int CH_EPI_STORE = 0;                // y fp16 [N] = rs * sum: plain store (read by the NEXT launch)
constexpr int CH_EPI_RESID = 1;                // h = h + sum (fp32 residual stream); with gamma_out: publish fp16(h * gamma_out) + the block's sum of h^2
constexpr int CH_EPI_PAIR = 2;                 // rows pair-interleaved (V3_MODE_PAIR): publish silu(rs * gate) * (rs * up)
constexpr uint32_t CH_ST_TIMEOUT = 0x7100;     // status code: a gather gave up (| phase)

struct ChPhase {
    const uint8_t* qw;        // int16 [N/4][K] checkpoint layout
    const uint8_t* szp;       // u32 [N/16][K/128][16] (scale | scaled_zero << 16), qeft_pack_scales
    const uint8_t* ow;        // fp16 [N][128] plain outlier rows
    const f16* gamma_out;     // RESID: optional
    float* h32;               // RESID: the residual stream, fp32 [N] (read by the first RESID phase of a launch, written by the last)
    f16* y;                   // STORE: fp16 [N]
    const f16* x;             // phase 0 only: the plain fp16 input vector [K]
    int K, nsets;             // in_features, N / 16
    int epi;                  // CH_EPI_*
    int norm_in;              // multiply the row sums by rsqrt(mean of the producer's h^2 + eps) (the ssq granules of the incoming edge)
    int sets_q, sets_r;       // nsets / nblk, nsets % nblk
    int store_h;              // RESID: store h32 (the last RESID phase of the launch)
    float eps;
};

struct ChArgs {
    ChPhase ph[CH_MAX_PHASES];
    int nph, nblk;
    unsigned long long* gran[CH_MAX_PHASES];   // edge e (phase e -> e + 1): granules {value, tag}; ssq granules behind them at gran_ssq_off
    uint32_t gran_ssq_off;                     // index of the first ssq granule inside an edge buffer (>= max K / 2)
    uint32_t* epoch;                           // device word: tags of this launch are *epoch + 1 + edge; block 0 adds nph at the end
    uint32_t* status;                          // [0]: first give-up code (0 = none)
    long long* dbg;                            // lab: per block 8 stamps per phase (100 MHz), or NULL
    uint32_t timeout_ticks;                    // give-up bound of one gather (100 MHz ticks)
};

// ---- LDS carve-up
struct ChLds { uint32_t xs, szl, owl, epl, red, misc, ring, total; };
// kmax: the longest input vector; sz_max: the largest (row sets per block) x (scale bytes per set) of a phase; rsc_max: the most row sets
__host__ __device__ inline ChLds ch_lds(int kmax, uint32_t sz_max, int rsc_max, int D, int CH_NW) {
    ChLds L; uint32_t o = 0;
    L.xs = o;   o += (uint32_t)v3_x_bytes(kmax);
    L.szl = o;  o += sz_max;
    L.owl = o;  o += (uint32_t)rsc_max * 4096u;
    L.epl = o;  o += 2048u;                    // two images (phase parity): wave 0 still reads phase p's while phase p + 1's is staged
    L.red = o;  o += ((uint32_t)rsc_max * CH_NW * 16 * 4 + 1023u) / 1024u * 1024u;
    L.misc = o; o += 1024u;                    // [0] rs_norm, [1] abort flag, [64..] publish scratch
    L.ring = o; o += (uint32_t)CH_NW * D * 1024u;
    L.total = o;
    return L;
}

#if defined(__HIPCC__)
typedef __attribute__((address_space(1))) unsigned long long ch_gu64;
typedef __attribute__((address_space(1))) uint32_t ch_gu32;

// 1 KB LDS-DMA, non-temporal (the weights are read once: guide row nt-weights)
__device__ __forceinline__ void ch_dma16_nt(const void* gsrc, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

#define CH_STAMP(i) do { if (DBG) ts[i] = wall_clock64(); } while (0)

struct ChXF { v3h8 f[4]; float lo, hi; };      // the x fragments of a 128-k step and -1024 x (sum of x over its low / high nibble k)

// CH_NW: waves per block.  D: ring slots per wave.  DBG: time stamps (lab).
// NOMATH (lab ablation): the consumes read their ring slot and nothing else -- what the structure costs without the arithmetic.
template <int CH_NW, int D, bool DBG, bool NOMATH = false>
__global__ __launch_bounds__(CH_NW * 64) void gemv_chain_kernel(ChArgs A) {
    extern __shared__ __attribute__((aligned(1024))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nl = lane & 15, kc = lane >> 4;
    const int nph = A.nph, nblk = A.nblk;
    const int bslot = v3_xcd_block(blockIdx.x, nblk);        // this block's position in the deal of row sets

    // geometry of the LDS regions: sized for the largest phase (host and device agree through ch_lds)
    int kmax = 0, rscmax = 1;
    uint32_t szmax = 0;
#pragma unroll
    for (int p = 0; p < CH_MAX_PHASES; ++p)
        if (p < nph) {
            const int rsc = A.ph[p].sets_q + (A.ph[p].sets_r ? 1 : 0);
            kmax = max(kmax, A.ph[p].K);
            rscmax = max(rscmax, rsc);
            szmax = max(szmax, (uint32_t)rsc * (uint32_t)v3_sz_bytes(A.ph[p].K >> 7));
        }
    const ChLds L = ch_lds(kmax, szmax, rscmax, D, CH_NW);
    uint8_t* const xs = smem + L.xs;
    uint8_t* const szl = smem + L.szl;
    uint8_t* const owl = smem + L.owl;
    uint8_t* const epl = smem + L.epl;
    float* const red = (float*)(smem + L.red);
    float* const misc = (float*)(smem + L.misc);
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    // (a scalar load: no compiler-visible vector load may sit in front of the ring -- hipcc would place a vmcnt(0) at its first use)
    uint32_t epoch0;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(epoch0) : "s"(A.epoch) : "memory");
    long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    CH_STAMP(0);

    const uint32_t lane_c = (uint32_t)(kc >> 1) * 128u + (uint32_t)(nl & 3) * 32u + (uint32_t)(kc & 1) * 16u;
    const uint32_t lane_r = (uint32_t)(nl >> 2) * 2u;
    const uint32_t ring0 = __builtin_amdgcn_readfirstlane(lds0 + L.ring + (uint32_t)wave * (D * 1024u));
    const uint8_t* const ringp = smem + L.ring + (size_t)wave * (D * 1024) + (size_t)lane * 16;

    // ---- the issue cursor (wave-uniform): load number c of the wave's sequence over all phases
    int is_p = 0;                       // phase being issued
    int is_rs = 0, is_RS = 0, is_left = 0;
    const uint8_t* is_ptr = nullptr;    // current step, row set 0 of the block
    uint32_t is_setb = 0, is_lane = 0;
    bool is_active = true;
    auto is_enter = [&](int p) {
        while (p < nph) {
            const ChPhase& P = A.ph[p];
            int set0;
            v3_block_sets(bslot, P.sets_q, P.sets_r, set0, is_RS);
            const int K = P.K, nfull = (K >> 7) - 1;
            is_setb = (uint32_t)K * 8u;
            is_left = (nfull - wave + CH_NW - 1) / CH_NW;
            is_ptr = P.qw + (size_t)set0 * is_setb + (size_t)wave * 256u;
            is_lane = lane_r * (uint32_t)K + lane_c;
            is_rs = 0;
            if (is_left > 0 && is_RS > 0) break;
            ++p;
        }
        is_p = p;
        is_active = p < nph;
    };
    auto issue_next = [&](uint32_t slot_lds) {
        if (!is_active) return;
        ch_dma16_nt(is_ptr + (size_t)is_rs * is_setb + is_lane, slot_lds);
        if (++is_rs == is_RS) {
            is_rs = 0;
            is_ptr += CH_NW * 256;
            if (--is_left == 0) is_enter(is_p + 1);
        }
    };

    // ---- staging of a phase's block-constant operands (scale words, outlier rows, epilogue operands) by DMA: a flat list of 1 KB
    //      pieces dealt over the waves w0 .. NW - 1 (w0 = 1 behind a phase: wave 0 is busy with the epilogue and must reach its
    //      sweep without waiting for DMAs of its own)
    auto stage_phase = [&](int p, int w0) {
        if (wave < w0) return;
        const ChPhase& P = A.ph[p];
        int set0, RS;
        v3_block_sets(bslot, P.sets_q, P.sets_r, set0, RS);
        const int K = P.K;
        const V3Geom G{K, 128, K >> 7, (K >> 7) - 1, K >> 7, P.nsets};
        const int SZB = v3_sz_bytes(G.ngroups), SPS = SZB >> 10, PPS = SPS + 4;      // pieces per row set: scale words, 4 KB of outlier rows
        const int total = RS * PPS + (P.epi == CH_EPI_RESID ? 1 : 0);
        for (int t = wave - w0; t < total; t += CH_NW - w0) {
            const int rs = t / PPS, j = t - rs * PPS;
            if (rs >= RS) {
                // lanes [0, 32): the residual rows (4 floats each), lanes [32, 48): gamma_out (8 halves each); clamped lanes re-read element 0
                const uint8_t* src;
                if (lane < 32) src = (const uint8_t*)P.h32 + ((size_t)set0 * 16 + (size_t)(lane < 4 * RS ? lane : 0) * 4) * 4;
                else {
                    const int l = (lane - 32) & 15;
                    src = P.gamma_out ? (const uint8_t*)P.gamma_out + ((size_t)set0 * 16 + (size_t)(l < 2 * RS ? l : 0) * 8) * 2 : (const uint8_t*)P.h32;
                }
                v3_dma16(src, __builtin_amdgcn_readfirstlane(lds0 + L.epl + (uint32_t)(p & 1) * 1024u));
            } else if (j < SPS) {
                v3_dma16(P.szp + v3_sz_off(G, set0 + rs, j, lane), __builtin_amdgcn_readfirstlane(lds0 + L.szl + (uint32_t)rs * SZB + ((uint32_t)j << 10)));
            } else {
                v3_dma16(P.ow + v3_ow_off(set0 + rs, j - SPS, lane), __builtin_amdgcn_readfirstlane(lds0 + L.owl + (uint32_t)rs * 4096u + ((uint32_t)(j - SPS) << 10)));
            }
        }
    };

    // ---- prologue: the launch's input vector FIRST (a CU's memory queue is served in order), then phase 0's staging, then the
    //      ring (D - 1 loads; consume 0 issues the D-th into the last slot)
    {
        const ChPhase& P = A.ph[0];
        const V3Geom G{P.K, 128, P.K >> 7, (P.K >> 7) - 1, P.K >> 7, P.nsets};
        const int PX = v3_x_bytes(P.K) >> 10;
        for (int i = wave; i < PX; i += CH_NW)
            v3_dma16((const uint8_t*)P.x + v3_x_off(G, i, lane), __builtin_amdgcn_readfirstlane(lds0 + L.xs + ((uint32_t)i << 10)));
    }
    stage_phase(0, 0);
    is_enter(0);
    uint32_t slot = 0;                  // slot of the next consume
    for (int d = 0; d < D - 1; ++d) issue_next(ring0 + (uint32_t)d * 1024u);
    uint32_t refill = D - 1;            // the slot the next consume's refill goes to (= the previously consumed one)

    uint32_t MAGIC = 0x64006400u, NEG1024 = 0xE400E400u;
    asm volatile("" : "+v"(MAGIC), "+v"(NEG1024));
    const v3h8 c8 = __builtin_bit_cast(v3h8, u32x4{NEG1024, NEG1024, NEG1024, NEG1024});
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    float h_keep = 0.f;                 // wave 0, lane = row: the residual value between two RESID phases of the launch
    bool h_valid = false;
    bool aborted = false;

    for (int p = 0; p < nph; ++p) {
        const ChPhase& P = A.ph[p];
        int set0, RS;
        v3_block_sets(bslot, P.sets_q, P.sets_r, set0, RS);
        const int RSC = P.sets_q + (P.sets_r ? 1 : 0);
        const int K = P.K, nfull = (K >> 7) - 1, epi = P.epi;
        const int SZB = v3_sz_bytes(K >> 7);
        const int nsw = (nfull - wave + CH_NW - 1) / CH_NW;
        // this wave's staging pieces (phase 0: they and x are OLDER than the D - 1 ring loads; later phases: everything, also the
        // ring so far and wave 0's publication stores)
        if (p == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D - 1) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        CH_STAMP(1);
        if (p > 0) {
            // ---- edge p - 1: every wave sweeps its share of the granules into xs until every tag is this edge's
            const uint32_t tag = epoch0 + (uint32_t)p;
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)A.gran[p - 1], 0, (int)((A.gran_ssq_off + 512u) * 8u), 0x00020000);
            const int nvec = K >> 2;                                // 16-byte vectors = granule pairs
            const long long t0 = wall_clock64();
            constexpr int VPW = 48 / CH_NW;                         // vectors per lane and wave: K <= 48 * 64 * 4 = 12288
            u32x4 v[VPW];
            for (unsigned spins = 0;; ++spins) {
                bool ok = true;
#pragma unroll
                for (int k = 0; k < VPW; ++k) {
                    const int idx = (k * CH_NW + wave) * 64 + lane;
                    if ((k * CH_NW + wave) * 64 < nvec) {
                        v[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (idx < nvec ? idx : nvec - 1) * 16, 0, 16);     // aux 16 = sc1
                        ok &= v[k][1] == tag && v[k][3] == tag;
                    }
                }
                if (__all(ok) || aborted) break;
                if ((spins & 7u) == 7u) {
                    if (wall_clock64() - t0 > (long long)A.timeout_ticks ||
                        __hip_atomic_load((ch_gu32*)A.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                        aborted = true;
                        if (lane == 0) atomicCAS(A.status, 0u, CH_ST_TIMEOUT | (uint32_t)p);
                    }
                }
                __builtin_amdgcn_s_sleep(4);
            }
#pragma unroll
            for (int k = 0; k < VPW; ++k) {
                const int idx = (k * CH_NW + wave) * 64 + lane;
                if (idx < nvec) *(u32x2*)(xs + (size_t)idx * 8) = u32x2{v[k][0], v[k][2]};
            }
            if (P.norm_in && wave == CH_NW - 1) {
                // the producers' partial sums of squares: one granule per block, summed in a fixed order
                float part = 0.f;
                for (unsigned spins = 0;; ++spins) {
                    bool ok = true;
                    part = 0.f;
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int idx = k * 64 + lane;                                  // vector = granules 2 idx, 2 idx + 1
                        const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(A.gran_ssq_off * 8u) + idx * 16, 0, 16);
                        // (scalars first: hipcc 7.2 compiles __builtin_bit_cast(float, q[2]) of a vector ELEMENT as element 0)
                        const uint32_t q0 = q[0], q2 = q[2];
                        if (2 * idx < nblk) { ok &= q[1] == tag; part += __builtin_bit_cast(float, q0); }
                        if (2 * idx + 1 < nblk) { ok &= q[3] == tag; part += __builtin_bit_cast(float, q2); }
                    }
                    if (__all(ok) || aborted) break;
                    if ((spins & 7u) == 7u && wall_clock64() - t0 > (long long)A.timeout_ticks) {
                        aborted = true;
                        if (lane == 0) atomicCAS(A.status, 0u, CH_ST_TIMEOUT | (uint32_t)p);
                    }
                    __builtin_amdgcn_s_sleep(4);
                }
                part = wave_sum(part);
                if (lane == 0) misc[0] = __builtin_amdgcn_rsqf(part * (1.f / (float)K) + P.eps);
            }
        }
        CH_STAMP(2);
        if (p == 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // B1: x and every wave's staging are in LDS
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        CH_STAMP(3);
        const float rs_norm = P.norm_in ? misc[0] : 1.f;
        const uint8_t* const eplp = epl + (p & 1) * 1024;

        float acc[CH_MAX_RSC];          // row sets 0 .. RS - 2 (the block's LAST row set accumulates in acc_tail)
#pragma unroll
        for (int r = 0; r < CH_MAX_RSC; ++r) acc[r] = 0.f;
        float acc_tail = 0.f, acc_out = 0.f;
        const uint8_t* xa = xs + kc * 64;

        // the fp16 outlier columns: one MFMA step per row set, by the wave whose turn step nfull would be
        if (wave == (nfull & (CH_NW - 1))) {
            const v3h8* px = (const v3h8*)(xa + (size_t)nfull * 256);
            v3h8 xo[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) xo[jj] = px[jj];
#pragma unroll
            for (int rs = 0; rs < CH_MAX_RSC; ++rs)
                if (rs < RS) {
                    f32x4 Pm = z4;
                    const uint8_t* prow = owl + rs * 4096 + nl * 256;
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj)
                        Pm = __builtin_amdgcn_mfma_f32_16x16x32_f16(xo[jj], *(const v3h8*)(prow + (((kc * 4 + jj) ^ nl) & 15) * 16), Pm, 0, 0, 0);
                    if (rs == RS - 1) acc_out = Pm[0]; else acc[rs] = Pm[0];
                }
        }

        // ---- the steps: step-major over the block's RS row sets, software-pipelined: the ring slot and scale word of consume
        //      c + 1 are read from LDS, and the products of consume c - 1 folded, behind the MFMAs of consume c; the x fragments
        //      and bias sums of step i + 1 are fetched / formed during step i
        if (nsw > 0) {
            const uint8_t* xp = xa + (size_t)wave * 256;
            const uint8_t* sp = szl + (size_t)wave * 64 + nl * 4;
            ChXF XA, XB;
            auto load_xf = [&](ChXF& X, const uint8_t* ptr) {
#pragma unroll
                for (int w = 0; w < 4; ++w) X.f[w] = ((const v3h8*)ptr)[w];
            };
            load_xf(XA, xp);
            {
                f32x4 A0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(XA.f[0], c8, z4, 0, 0, 0);
                f32x4 A1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(XA.f[1], c8, z4, 0, 0, 0);
                A0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(XA.f[2], c8, A0, 0, 0, 0);
                A1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(XA.f[3], c8, A1, 0, 0, 0);
                XA.lo = A0[0]; XA.hi = A1[0];
            }
            XB.lo = XB.hi = 0.f;
            // pipeline registers: the current consume's ring data and scale word; the previous consume's products
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D - 2) : "memory");     // (phase 0: the ring's first load; later phases: all landed long ago)
            u32x4 wv = *(const u32x4*)(ringp + (size_t)slot * 1024);
            uint32_t szw = *(const uint32_t*)sp;
            f32x4 pPlo = z4, pPhi = z4;
            uint32_t pszw = 0;
            auto fold = [&](float& dst, float lo, float hi) {
                const h2 sz2 = as_h2(pszw);
                dst = dst + ((float)sz2[0] * ((pPlo[0] + lo) + 0.0625f * (pPhi[0] + hi)) + (float)sz2[1] * ((lo + hi) * -0.0009765625f));
            };
            // one step: C = this step's fragments / sums, N = the next step's (loaded here; its sums formed behind the last row set)
            auto step = [&](ChXF& C, ChXF& N, bool more) {
                f32x4 B0 = z4, B1 = z4;
#pragma unroll
                for (int rs = 0; rs < CH_MAX_RSC; ++rs) {
                    if (rs < RS) {
                        const bool last = rs == RS - 1;
                        if (!NOMATH && rs == 0 && more) load_xf(N, xp + CH_NW * 256);
                        issue_next(ring0 + refill * 1024u);
                        refill = slot;
                        slot = slot + 1 == D ? 0 : slot + 1;
                        f32x4 Plo = z4, Phi = z4;
                        if constexpr (NOMATH) {
                            const uint32_t w0 = wv[0] ^ wv[1], w1 = wv[2] ^ wv[3];
                            Plo[0] = __builtin_bit_cast(float, w0); Phi[0] = __builtin_bit_cast(float, w1);
                        } else {
                            u32x4 bf[4];
#pragma unroll
                            for (int w = 0; w < 4; ++w) {
                                const uint32_t v = wv[w], t = v >> 8;
                                bf[0][w] = (v & 0x000f000fu) | MAGIC;
                                bf[1][w] = (v & 0x00f000f0u) | MAGIC;
                                bf[2][w] = (t & 0x000f000fu) | MAGIC;
                                bf[3][w] = (t & 0x00f000f0u) | MAGIC;
                            }
                            Plo = __builtin_amdgcn_mfma_f32_16x16x32_f16(C.f[0], __builtin_bit_cast(v3h8, bf[0]), z4, 0, 0, 0);
                            Phi = __builtin_amdgcn_mfma_f32_16x16x32_f16(C.f[1], __builtin_bit_cast(v3h8, bf[1]), z4, 0, 0, 0);
                            Plo = __builtin_amdgcn_mfma_f32_16x16x32_f16(C.f[2], __builtin_bit_cast(v3h8, bf[2]), Plo, 0, 0, 0);
                            Phi = __builtin_amdgcn_mfma_f32_16x16x32_f16(C.f[3], __builtin_bit_cast(v3h8, bf[3]), Phi, 0, 0, 0);
                        }
                        // the NEXT consume's operands: its load is D - 2 issues back
                        if (is_active) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D - 2) : "memory");
                        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        const u32x4 wv_n = *(const u32x4*)(ringp + (size_t)slot * 1024);
                        const uint32_t szw_n = last ? *(const uint32_t*)(sp + (more ? CH_NW * 64 : 0)) : *(const uint32_t*)(sp + (size_t)(rs + 1) * SZB);
                        // the PREVIOUS consume's products (its MFMAs retired long ago): rs == 0: the previous step's last row set
                        if (rs == 0) fold(acc_tail, N.lo, N.hi); else fold(acc[rs - 1], C.lo, C.hi);
                        if (!NOMATH && last && more) {         // the next step's bias sums ride behind this step's last products
                            B0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(N.f[0], c8, z4, 0, 0, 0);
                            B1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(N.f[1], c8, z4, 0, 0, 0);
                            B0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(N.f[2], c8, B0, 0, 0, 0);
                            B1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(N.f[3], c8, B1, 0, 0, 0);
                        }
                        pPlo = Plo; pPhi = Phi; pszw = szw;
                        wv = wv_n; szw = szw_n;
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (more) {
                    N.lo = B0[0]; N.hi = B1[0];
                    xp += CH_NW * 256;
                    sp += CH_NW * 64;
                }
            };
            int i = 0;
            for (; i + 2 <= nsw; i += 2) {
                step(XA, XB, true);
                step(XB, XA, i + 2 < nsw);
            }
            if (i < nsw) {
                step(XA, XB, false);
                fold(acc_tail, XA.lo, XA.hi);
            } else {
                fold(acc_tail, XB.lo, XB.hi);
            }
        }
        acc_tail += acc_out;
        CH_STAMP(4);
        if (kc == 0) {
#pragma unroll
            for (int rs = 0; rs < CH_MAX_RSC; ++rs)
                if (rs < RS) red[(rs * CH_NW + wave) * 16 + nl] = rs == RS - 1 ? acc_tail : acc[rs];
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");             // B2: partial sums complete; xs / szl / owl free
        CH_STAMP(5);
        // the next phase's block-constant operands (they have the whole edge to land)
        if (p + 1 < nph) stage_phase(p + 1, 1);

        // ---- epilogue + publication: wave 0
        if (wave == 0) {
            const uint32_t tag = epoch0 + (uint32_t)p + 1u;
            ch_gu64* const gout = (ch_gu64*)A.gran[p];
            const bool lastp = p + 1 == nph;
            if (epi == CH_EPI_PAIR) {
                const int rs = lane >> 3, n = lane & 7;
                float gv = 0.f, uv = 0.f;
                if (lane < RS * 8) {
#pragma unroll
                    for (int w = 0; w < CH_NW; ++w) {
                        gv += red[(rs * CH_NW + w) * 16 + n];
                        uv += red[(rs * CH_NW + w) * 16 + n + 8];
                    }
                }
                gv *= rs_norm; uv *= rs_norm;
                const f16 g16 = (f16)gv, u16 = (f16)uv;
                const f16 r16 = (f16)(silu_f32((float)g16) * (float)u16);
                const uint32_t mine = (uint32_t)__builtin_bit_cast(uint16_t, r16);
                const uint32_t nb = (uint32_t)__shfl_down((int)mine, 1);
                if (lastp) {
                    if (lane < RS * 8) P.y[set0 * 8 + lane] = r16;
                } else if (lane < RS * 8 && !(lane & 1)) {
                    __hip_atomic_store(gout + ((set0 * 8 + lane) >> 1), ((unsigned long long)tag << 32) | (mine | (nb << 16)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else {
                float v = 0.f;
                const int rs = lane >> 4, n = lane & 15;
                const int row = set0 * 16 + lane;
                const bool act = lane < RS * 16;
                if (act) {
#pragma unroll
                    for (int w = 0; w < CH_NW; ++w) v += red[(rs * CH_NW + w) * 16 + n];
                }
                v *= rs_norm;
                if (epi == CH_EPI_STORE) {
                    if (act) P.y[row] = (f16)v;
                } else {
                    const float h0 = h_valid ? h_keep : ((const float*)eplp)[lane];
                    v += h0;
                    h_keep = v;
                    h_valid = true;
                    if (act && P.store_h) P.h32[row] = v;
                    if (P.gamma_out) {
                        const f16 xn = (f16)(v * (float)((const f16*)(eplp + 512))[lane]);
                        const uint32_t mine = (uint32_t)__builtin_bit_cast(uint16_t, xn);
                        const uint32_t nb = (uint32_t)__shfl_down((int)mine, 1);
                        float sq = act ? v * v : 0.f;
                        sq = wave_sum(sq);
                        if (!lastp) {
                            if (act && !(lane & 1))
                                __hip_atomic_store(gout + (row >> 1), ((unsigned long long)tag << 32) | (mine | (nb << 16)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (lane == 0)
                                __hip_atomic_store(gout + A.gran_ssq_off + bslot, ((unsigned long long)tag << 32) | __builtin_bit_cast(uint32_t, sq), __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                }
            }
            CH_STAMP(6);
            if (DBG && A.dbg) {
                long long* d = A.dbg + ((size_t)blockIdx.x * CH_MAX_PHASES + p) * 16;
                // checksum of the x vector this block consumed (wave 0 re-reads xs: only valid while nobody rewrites it -- lab use)
                if (lane == 0) {
                    for (int i = 0; i < 7; ++i) d[i] = ts[i];
                    d[7] = __builtin_amdgcn_s_getreg(6164) & 15;
                    d[8] = (long long)__builtin_bit_cast(uint32_t, rs_norm);
                }
            }
        }
    }
    // the launch's tags are spent: the next launch starts above them (all blocks have read epoch0 long ago: block 0 has gathered
    // every edge, so every block has published, so every block has started)
    if (blockIdx.x == 0 && tid == 0 && nph > 1) atomicAdd(A.epoch, (uint32_t)(nph - 1));
}
#endif  // __HIPCC__

}  // namespace qeft
