// Single-token attention of the decode harness (rotary + KV append + flash-decoding split): the kernel of decode_aux.hip, with its
// BODY as a device function (the round-4 lab kernel tools/attn_oproj_lab_kernel.h -- attention and o_proj in one launch, measured
// and rejected: profiles/r04_attn_oproj_fused.txt -- runs the same body).
#pragma once
#include "qeft_common.h"

namespace qeft {

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// maximum over the 64 lanes of a wave, every lane gets it: 4 DPP steps inside each row of 16, then two cross-row swaps
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v));    // quad_perm [1,0,3,2]
    v = fmaxf(v, dpp_mov<0x4E>(v));    // quad_perm [2,3,0,1]
    v = fmaxf(v, dpp_mov<0x141>(v));   // row_half_mirror
    v = fmaxf(v, dpp_mov<0x140>(v));   // row_mirror
    v = fmaxf(v, __shfl_xor(v, 16));
    v = fmaxf(v, __shfl_xor(v, 32));
    return v;
}

__device__ __forceinline__ void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

constexpr int kAttnRec = 132;   // floats per (head, split) record of the workspace: acc[128], max, sum, pad

// Single-token attention for one sequence.  grid = n_heads * S, block = 256 (4 waves), head_dim = 128.
//   q,k,v  : this token's projections [n_heads*128], [n_kv*128], [n_kv*128] (fp16)
//   cos/sin: [tab_rows][64] fp32 rotary table; tab_rows == 1: the row of THIS position, selected by the caller
//   kc, vc : caches [n_kv][max_seq][128] fp16;  *pos_ptr = index of this token (0-based)
//   out    : [n_heads*128] fp16, element i stored at out_pos[i] when out_pos is given
//   ws     : S > 1 only: [n_heads*S][kAttnRec] floats + [n_heads] uint32 arrival counters (zero before first use)
// 32 blocks pulling a whole head's K and V each are bound by what ONE CU can load (~100 KB took ~4 us), so a head is
// split over S blocks and the kernel is organised around latency:
//   * everything that does not depend on `pos` is requested first -- q/k/v, out_pos and, unconditionally, the K
//     quarter-rows and V pieces of the first PRE runs of each wave (rows past the context are fetched and ignored);
//     only the rotary entry waits for `pos`;
//   * positions are dealt in runs of 16 to the 4*S waves of a head (run r belongs to wave r % (4*S)); each wave
//     computes its scores, its own maximum, exp and P.V partials with wave-level operations only (flash-decoding
//     split), the block merges its 4 waves once through LDS;
//   * S > 1: every block publishes its (acc, max, sum) record with write-through stores and takes a ticket from the
//     head's counter; the block that draws the last ticket merges the S records in split order (deterministic) and
//     re-arms the counter.  Nobody waits for anybody (MI355X_MICROARCH.md, inter-workgroup visibility, table row 1).
// DH = 2 (one block per head only): the head's 128 output dims are dealt over DH blocks.  Both compute the scores (K is
// fetched twice), each fetches half of every V row and does half of P.V; their outputs are disjoint, so there is nothing
// to merge.  Per block 3/4 of the bytes (the bound at these sizes is what ONE CU can pull) and half of the P.V math.
// KFT: the key cache has the reference's FasterTransformer layout [n_kv][128/8][max_seq][8] (ft_attention.cpp:131-133,
// ftllama_modeling.py:62-65) instead of [n_kv][max_seq][128]; only the address of a 16-byte (position, 8-dim chunk)
// piece changes -- consecutive positions of one chunk are then contiguous.
template <bool KFT>
__device__ __forceinline__ size_t kcache_off(int p, int chunk, int max_seq) {
    return KFT ? ((size_t)chunk * max_seq + p) * 8 : (size_t)p * 128 + chunk * 8;
}

// Parameter order: what the first loads need comes first -- the leading 13 dwords of the kernel-argument segment are
// preloaded into SGPRs at wave launch (build flag -amdgpu-kernarg-preload-count), the rest arrives by scalar loads that
// overlap those first vector loads.  DBG (lab only, tools/attn_timeline.py): per-wave phase stamps.
// The kernel's BODY is a device function so that the lab's fused attention + o_proj launch can run it too:
// `bid` = the block's index among the attention blocks, `smem_raw` = its dynamic LDS, and -- S == 1 only -- `pub` != NULL sends
// every output element to pub[opos] by an agent-scope (write-through) store instead of a plain store to out[opos]: readers on
// other XCDs of the SAME launch can then be released by a flag (lab only; the product passes NULL).
template <int PRE, int DH = 1, bool KFT = false, bool DBG = false, bool ALIBI = false>
__device__ __forceinline__ void rope_attn_decode_body(uint8_t* smem_raw, const int bid, const int* __restrict__ pos_ptr,
                                                      const int* __restrict__ out_pos, const f16* __restrict__ q,
                                                      const f16* __restrict__ k, const f16* __restrict__ v,
                                                      const float* __restrict__ cs, uint32_t heads_kv_s_tab,
                                                      const float* __restrict__ sn, f16* __restrict__ kc, f16* __restrict__ vc,
                                                      f16* __restrict__ out, float* __restrict__ ws, int max_seq,
                                                      unsigned long long* dbg_ptr, const float* __restrict__ alibi,
                                                      f16* __restrict__ pub = nullptr) {
    constexpr int HD = 128;
    const int n_heads = (int)(heads_kv_s_tab & 0xfffu), n_kv = (int)((heads_kv_s_tab >> 12) & 0xfffu);
    const int S = (int)((heads_kv_s_tab >> 24) & 0xfu), tab_rows = (int)(heads_kv_s_tab >> 28);     // 1: cs / sn are this position's row; 0: the whole table
    unsigned long long* const dbg = DBG ? dbg_ptr : nullptr;
    unsigned long long stamp[10];
    auto mark = [&](int i) {
        if (dbg) { __builtin_amdgcn_sched_barrier(0); stamp[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
    };
    unsigned long long rt0 = dbg ? __builtin_amdgcn_s_memrealtime() : 0;
    mark(0);
    float* prob = (float*)smem_raw;          // [max_seq + 16] raw scores
    float* part = prob + max_seq + 16;       // [16*DH][128/DH] P.V partials: (wave, position class) x dims of this block
    float* psum = part + 16 * HD;            // [16*DH] exp sums
    float* wm = psum + 64;                   // [4] wave maxima
    f16* knew = (f16*)(wm + 4);              // [128]
    f16* vnew = knew + HD;                   // [128]
    f16* qs = vnew + HD;                     // [128] rotated, pre-scaled q (fp16 like the reference's rotated q)
    __shared__ int last_ticket;

    constexpr int HDB = HD / DH;             // output dims of this block
    constexpr int NDG = 16 / DH;             // 8-dim groups of this block
    constexpr int NPC = 64 / NDG;            // position classes of a 16-position run
    constexpr int PPC = 16 / NPC;            // positions per class
    const int dhi = (int)(bid % DH);
    const int hs_idx = bid / DH;
    const int h = hs_idx / S, sp = hs_idx % S, t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int gw = sp * 4 + w, NW = 4 * S;   // this wave among the head's waves
    const int grp = n_heads / n_kv, hk = h / grp;
    f16* kch = kc + (size_t)hk * max_seq * HD;
    f16* vch = vc + (size_t)hk * max_seq * HD;
    const bool appender = (sp == 0) && (dhi == 0) && (h % grp == 0);
    // ALIBI (the reference's single_query_attention boundary only): the head's linear position bias, slope * (key - query position)
    // added to the scaled score (decoder_masked_multihead_attention_template.hpp:1335-1345, no padding tokens)
    const float slope = ALIBI ? alibi[h] : 0.f;

    // ---- loads that do not depend on pos.  All of them are unconditional (addresses selected, never branched on)
    // and nothing is converted here: a conversion or a divergent branch makes the compiler wait for the load on the
    // spot, which serialises one memory round trip per load group at the top of the kernel.
    const int ti = dhi * HDB + (t & (HDB - 1));                        // output element of this thread (within the head)
    const int* opp = out_pos ? out_pos + h * HD + ti : pos_ptr;        // dummy in-range address when there is no map
    const int opos_raw = *opp;
    const f16* src = (t < 64) ? q + h * HD + t : (t < 128) ? k + hk * HD + (t - 64) : v + hk * HD + (t - 128);
    const f16 raw_a = src[0];
    const f16 raw_b = src[(t < 128) ? 64 : 0];
    float rc = 0.f, rsn = 0.f;
    if (tab_rows == 1) {                      // the caller already selected this position's rotary row: no wait for pos
        rc = cs[t & 63];
        rsn = sn[t & 63];
    }
    // `pos` is a scalar load: it arrives while the vector loads above are in flight, and nothing above waits for it
    mark(1);
    const int pos = *pos_ptr, L = pos + 1;
    if (pos < 0 || pos >= max_seq) return;   // never index the cache / rotary table out of range (grid-uniform)
    if (dbg) { asm volatile("" :: "s"(pos)); }
    mark(2);
    // ---- K/V of the first PRE runs of this wave.  One CU pulls ~50 GB/s, so only rows inside the context are
    // fetched; the load COUNT stays fixed (runs past the context re-read row 0, an L1 hit), which keeps the compiler's
    // vmcnt bookkeeping exact and lets the rotary / scores start while later rows are still in flight.
    // score role: position pj of the run, dims qd*8 + 32*j .. +8 (j = 0..3): the 4 lanes of a position read 64
    // contiguous bytes per load instruction
    const int qd = lane & 3, pj = lane >> 2;
    // P.V role: dims dhi*HDB + dg*8 .., positions pc*PPC .. of the run
    const int dg = lane & (NDG - 1), pc = lane / NDG;
    const int dim0 = dhi * HDB + dg * 8;
    h8 kpre[PRE][4], vpre[PRE][PPC];
#pragma unroll
    for (int i = 0; i < PRE; ++i) {
        const int r0 = (i * NW + gw) * 16;                      // max_seq % 16 == 0: a run never straddles the cache end
        const int krow = r0 < L ? r0 + pj : 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) kpre[i][j] = *(const h8*)(kch + kcache_off<KFT>(krow, qd + 4 * j, max_seq));
    }
#pragma unroll
    for (int i = 0; i < PRE; ++i) {
        const int r0 = (i * NW + gw) * 16;
        const int vrow = r0 < L ? r0 + pc * PPC : 0;
        const h8* row = (const h8*)(vch + (size_t)vrow * HD + dim0);
#pragma unroll
        for (int j = 0; j < PPC; ++j) vpre[i][j] = row[j * (HD / 8)];
    }
    const int opos = out_pos ? opos_raw : h * HD + ti;
    const float ra = (float)raw_a, rb = (float)raw_b;
    if (t < 128) {
        const int i = t & 63;
        const float c = (tab_rows == 1) ? rc : cs[(size_t)pos * 64 + i];
        const float sv = (tab_rows == 1) ? rsn : sn[(size_t)pos * 64 + i];
        const float r0 = ra * c - rb * sv, r1 = rb * c + ra * sv;
        if (t < 64) {
            const float scale = 0.08838834764831845f;  // 1/sqrt(128)
            qs[i] = (f16)(r0 * scale);
            qs[i + 64] = (f16)(r1 * scale);
        } else {
            const f16 k0 = (f16)r0, k1 = (f16)r1;
            knew[i] = k0;
            knew[i + 64] = k1;
            if (appender) {
                kch[kcache_off<KFT>(pos, i >> 3, max_seq) + (i & 7)] = k0;
                kch[kcache_off<KFT>(pos, (i >> 3) + 8, max_seq) + (i & 7)] = k1;
            }
        }
    } else {
        const int i = t - 128;
        const f16 vv = (f16)ra;
        vnew[i] = vv;
        if (appender) vch[(size_t)pos * HD + i] = vv;
    }
    // workgroup barrier for LDS only: __syncthreads() would also drain the K/V prefetch that is still in flight
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    mark(3);

    // ---- scores of this wave's runs: raw score -> prob[], running maximum
    h2 qreg[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) qreg[j] = *(const h2*)(qs + (j >> 2) * 32 + qd * 8 + 2 * (j & 3));
    float lmax = -3.0e38f;
    auto score_run = [&](int i, const h8* kr) {
        const int p = (i * NW + gw) * 16 + pj;
        float sdot = 0.f;
        // v_dot2_f32_f16: two products per instruction, fp32 accumulation, no conversions
        if (p == pos) {
#pragma unroll
            for (int j = 0; j < 16; ++j) sdot = dot2(qreg[j], *(const h2*)(knew + (j >> 2) * 32 + qd * 8 + 2 * (j & 3)), sdot);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const u32x4 kw = __builtin_bit_cast(u32x4, kr[j]);
#pragma unroll
                for (int e = 0; e < 4; ++e) sdot = dot2(qreg[j * 4 + e], as_h2(kw[e]), sdot);
            }
        }
        sdot += dpp_mov<0xB1>(sdot);
        sdot += dpp_mov<0x4E>(sdot);
        if constexpr (ALIBI) sdot += slope * (float)(p - pos);
        if (p >= L) sdot = -3.0e38f;         // fetched but outside the context (possibly uninitialised cache rows)
        if (qd == 0) prob[p] = sdot;
        lmax = fmaxf(lmax, sdot);
    };
#pragma unroll
    for (int i = 0; i < PRE; ++i)
        if ((i * NW + gw) * 16 < L) score_run(i, kpre[i]);
    for (int i = PRE; (i * NW + gw) * 16 < L; ++i) {
        h8 kr[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) kr[j] = *(const h8*)(kch + kcache_off<KFT>((i * NW + gw) * 16 + pj, qd + 4 * j, max_seq));
        score_run(i, kr);
    }
    const float mw = wave_max(lmax);
    mark(4);
    __builtin_amdgcn_wave_barrier();         // prob[] of this wave's runs is written and read by this wave only

    // ---- P.V partials of this wave's runs
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.f;
    float lsum = 0.f;
    auto pv_run = [&](int i, const h8* vr) {
        const int p0 = (i * NW + gw) * 16 + pc * PPC;
        float sraw[PPC];
#pragma unroll
        for (int j = 0; j < PPC; ++j) sraw[j] = prob[p0 + j];
#pragma unroll
        for (int j = 0; j < PPC; ++j) {
            const int p = p0 + j;
            if (p < L) {
                const float e = __expf(sraw[j] - mw);
                h8 vv = vr[j];
                if (p == pos) vv = *(const h8*)(vnew + dim0);
                lsum += e;
#pragma unroll
                for (int d = 0; d < 8; ++d) o[d] += e * (float)vv[d];
            }
        }
    };
#pragma unroll
    for (int i = 0; i < PRE; ++i)
        if ((i * NW + gw) * 16 < L) pv_run(i, vpre[i]);
    for (int i = PRE; (i * NW + gw) * 16 < L; ++i) {
        h8 vr[PPC];
#pragma unroll
        for (int j = 0; j < PPC; ++j)
            vr[j] = *(const h8*)(vch + (size_t)((i * NW + gw) * 16 + pc * PPC + j) * HD + dim0);
        pv_run(i, vr);
    }
    {
        float* dst = part + (w * NPC + pc) * HDB + dg * 8;
        *(f32x4*)dst = f32x4{o[0], o[1], o[2], o[3]};
        *(f32x4*)(dst + 4) = f32x4{o[4], o[5], o[6], o[7]};
        if (dg == 0) psum[w * NPC + pc] = lsum;
        if (lane == 0) wm[w] = mw;
    }
    mark(5);
    __syncthreads();
    mark(6);
    // ---- merge the block's 4 waves (a wave without positions has max -3e38: factor 0)
    const float M = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
    float acc = 0.f, den = 0.f;
    if (t < HDB) {
#pragma unroll
        for (int g = 0; g < 4 * NPC; ++g) {
            const float f = __expf(wm[g / NPC] - M);
            acc += f * part[g * HDB + t];
            den += f * psum[g];
        }
    }
    if (S == 1) {
        if (t < HDB) {
            const f16 ov = (f16)(acc / den);
            if (pub) __hip_atomic_store((uint16_t*)pub + opos, __builtin_bit_cast(uint16_t, ov), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else out[opos] = ov;
        }
        if (dbg && lane == 0) {
            mark(7);
            unsigned long long* d = dbg + ((size_t)bid * 4 + w) * 12;
            d[0] = rt0; d[1] = __builtin_amdgcn_s_memrealtime();
            for (int i = 0; i < 8; ++i) d[2 + i] = stamp[i];
            d[10] = d[11] = 0;
        }
        return;
    }
    // ---- publish this split's record, take a ticket; the last arriver merges the head
    float* rec = ws + (size_t)(h * S + sp) * kAttnRec;
    if (t < HD) st_agent(rec + t, acc);
    if (t == 0) {
        st_agent(rec + HD, M);
        st_agent(rec + HD + 1, den);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    mark(7);
    unsigned* ctr = (unsigned*)(ws + (size_t)n_heads * S * kAttnRec) + h;
    if (t == 0) {
        const unsigned ticket = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_ticket = (ticket == (unsigned)(S - 1));
        if (ticket == (unsigned)(S - 1)) __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    mark(8);
    if (dbg && lane == 0 && !last_ticket) {
        unsigned long long* d = dbg + ((size_t)bid * 4 + w) * 12;
        d[0] = rt0; d[1] = __builtin_amdgcn_s_memrealtime();
        for (int i = 0; i < 9; ++i) d[2 + i] = stamp[i];
        d[11] = 0;
    }
    if (!last_ticket) return;
    if (t < HD) {
        const float* r0 = ws + (size_t)h * S * kAttnRec;
        float Mh = -3.0e38f;
        for (int j = 0; j < S; ++j) Mh = fmaxf(Mh, ld_agent(r0 + j * kAttnRec + HD));
        float a2 = 0.f, d2 = 0.f;
        for (int j = 0; j < S; ++j) {
            const float f = __expf(ld_agent(r0 + j * kAttnRec + HD) - Mh);   // a split without positions: factor 0
            a2 += f * ld_agent(r0 + j * kAttnRec + t);
            d2 += f * ld_agent(r0 + j * kAttnRec + HD + 1);
        }
        out[opos] = (f16)(a2 / d2);
    }
    if (dbg && lane == 0) {
        mark(9);
        unsigned long long* d = dbg + ((size_t)bid * 4 + w) * 12;
        d[0] = rt0; d[1] = __builtin_amdgcn_s_memrealtime();
        for (int i = 0; i < 9; ++i) d[2 + i] = stamp[i];
        d[11] = stamp[9];
    }
}

template <int PRE, int DH = 1, bool KFT = false, bool DBG = false, bool ALIBI = false>
__global__ __launch_bounds__(256) void rope_attn_decode_kernel(const int* __restrict__ pos_ptr, const int* __restrict__ out_pos,
                                                               const f16* __restrict__ q, const f16* __restrict__ k,
                                                               const f16* __restrict__ v, const float* __restrict__ cs,
                                                               uint32_t heads_kv_s_tab, const float* __restrict__ sn,
                                                               f16* __restrict__ kc, f16* __restrict__ vc, f16* __restrict__ out,
                                                               float* __restrict__ ws, int max_seq, unsigned long long* dbg_ptr,
                                                               const float* __restrict__ alibi) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_raw[];
    rope_attn_decode_body<PRE, DH, KFT, DBG, ALIBI>(smem_raw, (int)blockIdx.x, pos_ptr, out_pos, q, k, v, cs, heads_kv_s_tab, sn, kc, vc, out, ws,
                                                    max_seq, dbg_ptr, alibi);
}

// LDS bytes of the body for a cache of max_seq positions
__host__ __device__ inline size_t rope_attn_smem_bytes(int max_seq) { return (size_t)(max_seq + 16) * 4 + 16 * 128 * 4 + 64 * 4 + 4 * 4 + 3 * 128 * 2; }

}  // namespace qeft
