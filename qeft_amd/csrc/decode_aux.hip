// Small kernels around the quantized linears in a decode step: RMSNorm,
// rotary + KV-cache append + single-query attention, SiLU*mul.  They exist so the decode-token benchmark is a
// whole Llama step inside one hipGraph; none of them is on the packed-weight path itself.
//   RMSNorm    reference: qeft/kernel/layernorm/layernorm.cu:26-76 (generalT5LayerNorm): fp32 sum of squares,
//              y = x * rsqrt(mean(x^2) + eps) * gamma, one rounding to fp16.
//   attention  reference: qeft/kernel/attention (vendored FT masked MHA, ft_attention.cpp:110-181): rotary
//              (neox style = HF rotate_half), append k/v at `timestep`, softmax(q.K^T/sqrt(d)).V.
#include <cstdlib>

#include "qeft_common.h"

namespace qeft {

__device__ __forceinline__ float block_sum_256(float v, float* sm) {
    v = wave_sum(v);
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
        o[j] = (f16)(silu_f32((float)g[j]) * (float)u[j]);
    }
    *(h8*)(out + i) = o;
}

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
template <int PRE, int DH = 1, bool KFT = false, bool DBG = false, bool ALIBI = false>
__global__ __launch_bounds__(256) void rope_attn_decode_kernel(const int* __restrict__ pos_ptr, const int* __restrict__ out_pos,
                                                               const f16* __restrict__ q, const f16* __restrict__ k,
                                                               const f16* __restrict__ v, const float* __restrict__ cs,
                                                               uint32_t heads_kv_s_tab, const float* __restrict__ sn,
                                                               f16* __restrict__ kc, f16* __restrict__ vc, f16* __restrict__ out,
                                                               float* __restrict__ ws, int max_seq, unsigned long long* dbg_ptr,
                                                               const float* __restrict__ alibi) {
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
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_raw[];
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
    const int dhi = (int)(blockIdx.x % DH);
    const int hs_idx = blockIdx.x / DH;
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
        if (t < HDB) out[opos] = (f16)(acc / den);
        if (dbg && lane == 0) {
            mark(7);
            unsigned long long* d = dbg + ((size_t)blockIdx.x * 4 + w) * 12;
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
        unsigned long long* d = dbg + ((size_t)blockIdx.x * 4 + w) * 12;
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
        unsigned long long* d = dbg + ((size_t)blockIdx.x * 4 + w) * 12;
        d[0] = rt0; d[1] = __builtin_amdgcn_s_memrealtime();
        for (int i = 0; i < 9; ++i) d[2 + i] = stamp[i];
        d[11] = stamp[9];
    }
}

// ---- token boundary of the decode loop (main.py:340-371 / benchmark.py:293-338: embedding lookup in front of the
// layers, greedy argmax behind lm_head).  Two small launches instead of four framework kernels per token.
// begin: h = embed[tok], rope_row = rope_tab[pos] (cos 64 | sin 64).  grid = hidden / 2048 (+1), block 256.
__global__ __launch_bounds__(256) void token_begin_kernel(const f16* __restrict__ embed, const long long* __restrict__ tok,
                                                          const float* __restrict__ rope_tab, const int* __restrict__ pos,
                                                          f16* __restrict__ h, float* __restrict__ rope_row, int hidden,
                                                          int vocab, int max_seq) {
    const long long tk = min(max(*tok, 0ll), (long long)vocab - 1);
    const int i = (blockIdx.x * 256 + threadIdx.x) * 8;
    if (i < hidden) *(h8*)(h + i) = *(const h8*)(embed + (size_t)tk * hidden + i);
    if (blockIdx.x == 0 && threadIdx.x < 128 && rope_row) {
        const int p = min(max(*pos, 0), max_seq - 1);
        rope_row[threadIdx.x] = rope_tab[(size_t)p * 128 + threadIdx.x];
    }
}

// begin + the first layer's RMSNorm in the producer form the v3 GEMV consumes: hnorm = fp16(h * gamma), ssq_out[block] =
// this block's sum of h^2 (the consumer multiplies its outputs by rsqrt(sum / hidden + eps)).  grid = ceil(hidden / 2048).
__global__ __launch_bounds__(256) void token_begin_norm_kernel(const f16* __restrict__ embed, const long long* __restrict__ tok,
                                                               const float* __restrict__ rope_tab, const int* __restrict__ pos,
                                                               float* __restrict__ h, float* __restrict__ rope_row,
                                                               const f16* __restrict__ gamma, f16* __restrict__ hnorm,
                                                               float* __restrict__ ssq_out, int hidden, int vocab, int max_seq) {
    __shared__ float sm[4];
    const long long tk = min(max(*tok, 0ll), (long long)vocab - 1);
    const int i = (blockIdx.x * 256 + threadIdx.x) * 8;
    float ss = 0.f;
    if (i < hidden) {
        const h8 v = *(const h8*)(embed + (size_t)tk * hidden + i), g = *(const h8*)(gamma + i);
        h8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            ss += (float)v[j] * (float)v[j];
            o[j] = (f16)((float)v[j] * (float)g[j]);
            h[i + j] = (float)v[j];
        }
        *(h8*)(hnorm + i) = o;
    }
    ss = block_sum_256(ss, sm);
    if (threadIdx.x == 0) ssq_out[blockIdx.x] = ss;
    if (blockIdx.x == 0 && threadIdx.x < 128 && rope_row) {
        const int p = min(max(*pos, 0), max_seq - 1);
        rope_row[threadIdx.x] = rope_tab[(size_t)p * 128 + threadIdx.x];
    }
}

// h_out = h (+ add); with gamma: hnorm = fp16(h_out * gamma), ssq_out[block] = this block's sum of h_out^2.  The stand-alone
// form of what the v3 GEMV epilogue emits (tensor-parallel path: the residual add follows an all-reduce).  grid = ceil(hidden / 2048).
__global__ __launch_bounds__(256) void residual_norm_kernel(const float* __restrict__ h, const f16* __restrict__ add,
                                                            const f16* __restrict__ gamma, float* __restrict__ h_out,
                                                            f16* __restrict__ hnorm, float* __restrict__ ssq_out, int hidden) {
    __shared__ float sm[4];
    const int i = (blockIdx.x * 256 + threadIdx.x) * 8;
    float ss = 0.f;
    if (i < hidden) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = h[i + j];
        if (add) {
            const h8 w = *(const h8*)(add + i);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += (float)w[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) h_out[i + j] = v[j];
        if (gamma) {
            const h8 g = *(const h8*)(gamma + i);
            h8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                ss += v[j] * v[j];
                o[j] = (f16)(v[j] * (float)g[j]);
            }
            *(h8*)(hnorm + i) = o;
        }
    }
    if (gamma) {
        ss = block_sum_256(ss, sm);
        if (threadIdx.x == 0) ssq_out[blockIdx.x] = ss;
    }
}

// y[m][H] (fp16) = rmsnorm(x[m][H] fp32) * gamma: the final norm of the decode engine, whose residual stream is fp32.
__global__ __launch_bounds__(256) void rmsnorm_f32_kernel(const float* __restrict__ x, const f16* __restrict__ gamma,
                                                          f16* __restrict__ y, int H, float eps) {
    __shared__ float sm[4];
    const size_t base = (size_t)blockIdx.x * H;
    float ss = 0.f;
    for (int i = threadIdx.x; i < H; i += 256) ss += x[base + i] * x[base + i];
    ss = block_sum_256(ss, sm);
    const float rs = rsqrtf(ss / (float)H + eps);
    for (int i = threadIdx.x; i < H; i += 256) y[base + i] = (f16)(x[base + i] * rs * (float)gamma[i]);
}

// Rotary embedding of a whole prompt, in place: x [T][H][128] fp16, cos / sin [T][64] fp32 (NeoX pairing i, i + 64; fp32
// math, one rounding -- the arithmetic of the decode kernel's rope).  Prefill helper: torch's float / mul / cat sequence for
// the same was ~0.13 ms per layer at T = 2048.
__global__ __launch_bounds__(256) void rope_rows_kernel(f16* __restrict__ x, const float* __restrict__ cs,
                                                        const float* __restrict__ sn, int H, int row_stride) {
    const int t = blockIdx.x, i = threadIdx.x & 63;
    const float c = cs[(size_t)t * 64 + i], s = sn[(size_t)t * 64 + i];
    for (int h = threadIdx.x >> 6; h < H; h += 4) {
        f16* p = x + (size_t)t * row_stride + h * 128;
        const float a = (float)p[i], b = (float)p[i + 64];
        p[i] = (f16)(a * c - b * s);
        p[i + 64] = (f16)(b * c + a * s);
    }
}

hipError_t rope_rows_launch(void* x, const void* cs, const void* sn, int T, int H, int row_stride, hipStream_t st) {
    hipLaunchKernelGGL(rope_rows_kernel, dim3(T), dim3(256), 0, st, (f16*)x, (const float*)cs, (const float*)sn, H, row_stride);
    return hipGetLastError();
}

// Token tail of the decode harness: logits = fp16(W . fp16(rmsnorm(h32) * gamma)) for the fp16 head W [vocab][H] -- the final
// RMSNorm and the head GEMV in one launch (hipBLASLt's GEMV of the 262 MB Llama-2 head ran at 4.7 TB/s behind a 9 us norm
// launch).  Every block normalises the (16 KB, L2-resident) vector itself and keeps its lanes' slice of it in registers: a
// wave then streams whole rows, 16 B per lane per load, 8 loads in flight per row and the next row's loads issued before
// the current row is reduced; v_dot2_f32_f16 with fp32 accumulation, one DPP wave sum per row.  H = 512 * LPR.
template <int LPR>
__global__ __launch_bounds__(512) void lm_head_f16_kernel(const float* __restrict__ h32, const f16* __restrict__ gamma,
                                                          const f16* __restrict__ W, f16* __restrict__ logits, int vocab,
                                                          float eps, int rows_per_block) {
    constexpr int H = 512 * LPR;
    __shared__ float red[8];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the lane's x elements: chunk c covers k = 512 c + 8 lane .. + 7
    float xv[LPR][8];
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < LPR; ++c) {
        const f32x4 a = *(const f32x4*)(h32 + c * 512 + lane * 8), b = *(const f32x4*)(h32 + c * 512 + lane * 8 + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            xv[c][j] = a[j];
            xv[c][4 + j] = b[j];
            ss += a[j] * a[j] + b[j] * b[j];
        }
    }
    ss = wave_sum(ss);                          // every wave holds the whole vector: no block reduction
    const float rs = rsqrtf(ss / (float)H + eps);
    h2 xh[LPR][4];
#pragma unroll
    for (int c = 0; c < LPR; ++c) {
        const h8 g = *(const h8*)(gamma + c * 512 + lane * 8);
#pragma unroll
        for (int j = 0; j < 4; ++j)             // the rounding of qeft_rmsnorm_f32: fp16(x * rs * gamma)
            xh[c][j] = h2{(f16)(xv[c][2 * j] * rs * (float)g[2 * j]), (f16)(xv[c][2 * j + 1] * rs * (float)g[2 * j + 1])};
    }
    (void)red;
    const int r_end = min(vocab, (int)(blockIdx.x + 1) * rows_per_block);
    int r = blockIdx.x * rows_per_block + wave;
    u32x4 cur[LPR], nxt[LPR];
    auto load_row = [&](int row, u32x4 (&dst)[LPR]) {
        const u32x4* p = (const u32x4*)(W + (size_t)min(row, vocab - 1) * H) + lane;      // clamped: never out of range
#pragma unroll
        for (int c = 0; c < LPR; ++c) dst[c] = __builtin_nontemporal_load(p + c * 64);
    };
    load_row(r, cur);
    for (; r < r_end; r += 8) {
        load_row(r + 8, nxt);
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < LPR; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_fdot2(as_h2(cur[c][j]), xh[c][j], acc, false);
        acc = wave_sum(acc);
        if (lane == 0) logits[r] = (f16)acc;
#pragma unroll
        for (int c = 0; c < LPR; ++c) cur[c] = nxt[c];
    }
}

hipError_t lm_head_f16_launch(const void* h32, const void* gamma, const void* W, void* logits, int H, int vocab, float eps,
                              hipStream_t st) {
    const int blocks = vocab >= 4096 ? 512 : (vocab + 7) / 8, rpb = (vocab + blocks - 1) / blocks;
    auto go = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, st, (const float*)h32, (const f16*)gamma, (const f16*)W, (f16*)logits,
                           vocab, eps, rpb);
        return hipGetLastError();
    };
    switch (H) {
        case 512: return go(lm_head_f16_kernel<1>);
        case 1024: return go(lm_head_f16_kernel<2>);
        case 2048: return go(lm_head_f16_kernel<4>);
        case 4096: return go(lm_head_f16_kernel<8>);
        case 5120: return go(lm_head_f16_kernel<10>);
        case 8192: return go(lm_head_f16_kernel<16>);
        default: return hipErrorInvalidValue;
    }
}

// end: tok = argmax(logits) when greedy (lowest index among equal maxima, like torch.argmax), pos += 1.  One block.
__global__ __launch_bounds__(1024) void token_end_kernel(const f16* __restrict__ logits, long long* __restrict__ tok,
                                                         int* __restrict__ pos, int vocab, int greedy) {
    __shared__ float bv[16];
    __shared__ int bi[16];
    const int t = threadIdx.x;
    if (greedy) {
        float best = -INFINITY;
        int idx = 0x7fffffff;
        for (int i = t * 8; i < vocab; i += 1024 * 8) {
            if (i + 8 <= vocab) {
                const h8 v = *(const h8*)(logits + i);
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if ((float)v[j] > best) { best = (float)v[j]; idx = i + j; }
            } else {
                for (int j = i; j < vocab; ++j)
                    if ((float)logits[j] > best) { best = (float)logits[j]; idx = j; }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o);
            const int oi = __shfl_xor(idx, o);
            if (ob > best || (ob == best && oi < idx)) { best = ob; idx = oi; }
        }
        if ((t & 63) == 0) { bv[t >> 6] = best; bi[t >> 6] = idx; }
        __syncthreads();
        if (t == 0) {
            for (int w = 1; w < 16; ++w)
                if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
            *tok = idx == 0x7fffffff ? 0 : idx;
        }
    }
    if (t == 0) *pos = *pos + 1;
}

hipError_t token_begin_launch(const void* embed, const void* tok, const void* rope_tab, const int* pos, void* h,
                              void* rope_row, int hidden, int vocab, int max_seq, hipStream_t st) {
    hipLaunchKernelGGL(token_begin_kernel, dim3((hidden / 8 + 255) / 256), dim3(256), 0, st, (const f16*)embed,
                       (const long long*)tok, (const float*)rope_tab, pos, (f16*)h, (float*)rope_row, hidden, vocab, max_seq);
    return hipGetLastError();
}

int token_begin_norm_blocks(int hidden) { return (hidden / 8 + 255) / 256; }

hipError_t token_begin_norm_launch(const void* embed, const void* tok, const void* rope_tab, const int* pos, void* h,
                                   void* rope_row, const void* gamma, void* hnorm, float* ssq_out, int hidden, int vocab,
                                   int max_seq, hipStream_t st) {
    hipLaunchKernelGGL(token_begin_norm_kernel, dim3(token_begin_norm_blocks(hidden)), dim3(256), 0, st, (const f16*)embed,
                       (const long long*)tok, (const float*)rope_tab, pos, (float*)h, (float*)rope_row, (const f16*)gamma,
                       (f16*)hnorm, ssq_out, hidden, vocab, max_seq);
    return hipGetLastError();
}

hipError_t residual_norm_launch(const void* h, const void* add, const void* gamma, void* h_out, void* hnorm, float* ssq_out,
                                int hidden, hipStream_t st) {
    hipLaunchKernelGGL(residual_norm_kernel, dim3(token_begin_norm_blocks(hidden)), dim3(256), 0, st, (const float*)h,
                       (const f16*)add, (const f16*)gamma, (float*)h_out, (f16*)hnorm, ssq_out, hidden);
    return hipGetLastError();
}

hipError_t rmsnorm_f32_launch(const void* x, const void* gamma, void* y, int m, int H, float eps, hipStream_t st) {
    hipLaunchKernelGGL(rmsnorm_f32_kernel, dim3(m), dim3(256), 0, st, (const float*)x, (const f16*)gamma, (f16*)y, H, eps);
    return hipGetLastError();
}

hipError_t token_end_launch(const void* logits, void* tok, int* pos, int vocab, int greedy, hipStream_t st) {
    hipLaunchKernelGGL(token_end_kernel, dim3(1), dim3(1024), 0, st, (const f16*)logits, (long long*)tok, pos, vocab, greedy);
    return hipGetLastError();
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

unsigned long long* g_attn_dbg = nullptr;   // lab only: per-wave phase stamps
size_t attn_workspace_bytes(int n_heads, int S) { return S > 1 ? ((size_t)n_heads * S * kAttnRec + n_heads) * 4 : 0; }

hipError_t rope_attn_decode_launch(const void* q, const void* k, const void* v, const void* cs, const void* sn, void* kc,
                                   void* vc, const int* pos, const int* out_pos, void* out, void* ws, int n_heads,
                                   int n_kv, int max_seq, int S, int tab_rows, hipStream_t st, bool k_ft_layout,
                                   const float* alibi_slopes) {
    const size_t smem = (size_t)(max_seq + 16) * 4 + 16 * 128 * 4 + 64 * 4 + 4 * 4 + 3 * 128 * 2;
    // prefetch 256 positions per head whatever the split: PRE runs of 16 on each of the 4*S waves
    int dh = 1;
    auto launch = [&](auto kern) -> hipError_t {
        if (smem > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
            if (e != hipSuccess) return e;
        }
        if (n_heads > 4095 || n_kv > 4095 || S > 15) return hipErrorInvalidValue;
        const uint32_t packed = (uint32_t)n_heads | ((uint32_t)n_kv << 12) | ((uint32_t)S << 24) | ((tab_rows == 1 ? 1u : 0u) << 28);
        hipLaunchKernelGGL(kern, dim3(n_heads * S * dh), dim3(256), smem, st, pos, out_pos, (const f16*)q, (const f16*)k,
                           (const f16*)v, (const float*)cs, packed, (const float*)sn, (f16*)kc, (f16*)vc, (f16*)out, (float*)ws,
                           max_seq, g_attn_dbg, alibi_slopes);
        return hipGetLastError();
    };
    // measured on the 7B decode step (contexts 64..192): 682 / 689 / 688 tokens/s with 1 / 2 / 4 blocks per head
    static const int dh_env = getenv("QEFT_ATTN_DH") ? atoi(getenv("QEFT_ATTN_DH")) : 2;   // A/B switch: 1, 2 or 4 (4: 797 vs 799 tokens/s)
    if (k_ft_layout) {     // the reference's single_query_attention boundary: one block per head
        if (S != 1) return hipErrorInvalidValue;
        if (alibi_slopes) return launch(rope_attn_decode_kernel<4, 1, true, false, true>);
        return launch(rope_attn_decode_kernel<4, 1, true>);
    }
    if (g_attn_dbg) {      // lab: the stamped variants
        if (S == 1 && dh_env == 2) {
            dh = 2;
            return launch(rope_attn_decode_kernel<4, 2, false, true>);
        }
        if (S == 1) return launch(rope_attn_decode_kernel<4, 1, false, true>);
        if (S == 2) return launch(rope_attn_decode_kernel<2, 1, false, true>);
        return launch(rope_attn_decode_kernel<1, 1, false, true>);
    }
    if (S == 1 && dh_env == 4) {
        dh = 4;
        return launch(rope_attn_decode_kernel<4, 4>);
    }
    if (S == 1 && dh_env == 2) {
        dh = 2;
        return launch(rope_attn_decode_kernel<4, 2>);
    }
    if (S == 1) return launch(rope_attn_decode_kernel<4>);
    if (S == 2) return launch(rope_attn_decode_kernel<2>);
    return launch(rope_attn_decode_kernel<1>);
}

// ---- single-query attention for head sizes other than 128 (round 4; the `qeft_cuda.single_query_attention` boundary only).
// The reference instantiates its FasterTransformer kernel for Dh = 32 .. 256 (ft_attention.cpp:110-181,
// decoder_masked_multihead_attention.cu:30-59); the decode engine's kernel above is built around Dh = 128 (Llama-2).  This one
// takes any Dh % 8 == 0, 8 <= Dh <= 256: one block per head, the reference's cache layouts (keys [n_kv][Dh/8][L][8], values
// [n_kv][L][Dh]), no rotary (the shim rotates q / k with torch ops first, as it does for partial / GPT-J rotary at Dh = 128),
// optional ALiBi.  The block of a group's first head appends k / v at *pos; every block takes position *pos from the call's
// own k / v (not from the cache), so no block depends on another's stores.  Correctness first: scores one position per
// thread, P.V one (position class, dim) per thread, two LDS reductions.
__global__ __launch_bounds__(256) void sqa_generic_kernel(const f16* __restrict__ q, const f16* __restrict__ k, const f16* __restrict__ v,
                                                          f16* __restrict__ kc, f16* __restrict__ vc, const int* __restrict__ pos_ptr,
                                                          const float* __restrict__ alibi, f16* __restrict__ out, int n_heads, int n_kv,
                                                          int max_seq, int D) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem_raw[];
    float* qs = (float*)smem_raw;                 // [D] scaled q
    float* knew = qs + 256;                       // [D]
    float* vnew = knew + 256;                     // [D]
    float* red = vnew + 256;                      // [256]
    float* prob = red + 256;                      // [max_seq]
    const int h = blockIdx.x, t = threadIdx.x;
    const int grp = n_heads / n_kv, hk = h / grp;
    const int pos = *pos_ptr, L = pos + 1;
    if (pos < 0 || pos >= max_seq) return;
    f16* kch = kc + (size_t)hk * max_seq * D;
    f16* vch = vc + (size_t)hk * max_seq * D;
    const float scale = __builtin_amdgcn_rsqf((float)D);
    if (t < D) {
        const f16 kv = k[hk * D + t], vv = v[hk * D + t];
        qs[t] = (float)q[h * D + t] * scale;
        knew[t] = (float)kv;
        vnew[t] = (float)vv;
        if (h % grp == 0) {
            kch[((size_t)(t >> 3) * max_seq + pos) * 8 + (t & 7)] = kv;
            vch[(size_t)pos * D + t] = vv;
        }
    }
    __syncthreads();
    const float slope = alibi ? alibi[h] : 0.f;
    float lmax = -3.0e38f;
    for (int p = t; p < L; p += 256) {
        float sdot = 0.f;
        if (p == pos) {
            for (int d = 0; d < D; ++d) sdot += qs[d] * knew[d];
        } else {
            for (int c = 0; c < D / 8; ++c) {
                const h8 kk = *(const h8*)(kch + ((size_t)c * max_seq + p) * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) sdot += qs[c * 8 + e] * (float)kk[e];
            }
        }
        sdot += slope * (float)(p - pos);
        prob[p] = sdot;
        lmax = fmaxf(lmax, sdot);
    }
    red[t] = lmax;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (t < s) red[t] = fmaxf(red[t], red[t + s]);
        __syncthreads();
    }
    const float m = red[0];
    __syncthreads();
    float lsum = 0.f;
    for (int p = t; p < L; p += 256) {
        const float e = __expf(prob[p] - m);
        prob[p] = e;
        lsum += e;
    }
    red[t] = lsum;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (t < s) red[t] += red[t + s];
        __syncthreads();
    }
    const float inv = 1.f / red[0];
    __syncthreads();
    // P.V: thread = (position class pcl = t / D, dim d = t % D); NG classes share the positions
    const int NG = 256 / D > 0 ? 256 / D : 1;
    float o = 0.f;
    const int d = t % D, pcl = t / D;
    if (pcl < NG)
        for (int p = pcl; p < L; p += NG) o += prob[p] * (p == pos ? vnew[d] : (float)vch[(size_t)p * D + d]);
    red[t] = pcl < NG ? o : 0.f;
    __syncthreads();
    if (t < D) {
        float acc = 0.f;
        for (int g = 0; g < NG; ++g) acc += red[g * D + t];
        out[h * D + t] = (f16)(acc * inv);
    }
}

hipError_t sqa_generic_launch(const void* q, const void* k, const void* v, void* kc, void* vc, const int* pos, const float* alibi,
                              void* out, int n_heads, int n_kv, int max_seq, int head_dim, hipStream_t st) {
    const size_t smem = (size_t)(4 * 256 + max_seq) * 4;
    if (smem > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute((const void*)sqa_generic_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(sqa_generic_kernel, dim3(n_heads), dim3(256), smem, st, (const f16*)q, (const f16*)k, (const f16*)v, (f16*)kc, (f16*)vc,
                       pos, alibi, (f16*)out, n_heads, n_kv, max_seq, head_dim);
    return hipGetLastError();
}

}  // namespace qeft
