// LAB ONLY (round 4, rejected: profiles/r04_attn_oproj_fused.txt): decode attention and o_proj in ONE launch.  Parity-green against the
// two launches while it sat behind the C ABI (KV cache and attention output bit-equal, h32 / y_norm within the GEMV tolerance), but
// 14.2 us in the kernel + 3.8 us around it against 9.5 us for the two launches it replaces, so it never shipped.  Built and run by
// tools/gpu_ao_lab.sh (tools/attn_oproj_lab.hip); kept because the phase table it prints is the evidence.
//
// Why this pair and no other: the round-4 persistent-layer experiment (profiles/r04_persistent_edge.txt) showed that prefetching a
// linear's weights across an in-launch dependency buys nothing when the linear is big -- the prefetched bytes still have to be
// processed after x arrives, at about the rate they would have streamed.  o_proj is the exception twice over: its whole share of a
// block (one 16-row set = 33 KB at hidden 4096) fits IN FLIGHT at once, and its producer, the single-token attention, is a
// latency-bound kernel on 64 of the 256 CUs (5.9 us per launch for a few hundred KB).  So one grid of 256 blocks does both:
//   * every GEMV wave first requests ALL of its o_proj operands (weight ring by LDS-DMA, scale words, outlier rows, residual / gamma);
//   * blocks [0, n_heads * DH) run the attention body (decode_attn.h, unchanged) on waves 0..3 while their waves 4..7 alone carry
//     the block's o_proj share; the body's outputs go to their positions in o_proj's x vector by agent-scope (write-through)
//     stores, and once those are acknowledged the block counts itself in on a device-resident counter;
//   * wave 0 of every block polls that ONE word (not the data: 256 blocks polling the same 16 KB of payload from memory was the
//     first version, and cost more than the launch it saved), invalidates, and the block stages x through its XCD's L2 exactly as
//     a stand-alone GEMV does, runs its steps on operands that have long landed, and finishes with gemv_v3's o_proj epilogue:
//     h32 += y, y_norm = fp16(h * gamma), one partial sum of squares per block.
// Every spin is bounded (status word, give-up code).  Counter and sequence word live on the device (hipGraph-replay safe).
// Reference counterparts: ft_attention.cpp:110-181 (single_query_attention) and gemv_cuda_qeft.cu:75-222 behind
// QuantLinear.forward_outlier_out_proj (qlinear.py:273-299); the fusion itself has none.
#include <hip/hip_runtime.h>

#include "decode_attn.h"
#include "gemv_v3.h"

// lab hooks (tools/attn_oproj_lab.hip): phase time stamps of wave 0 of every block; nothing in the product build
#ifndef AO_STAMP
#define AO_STAMP_DECL
#define AO_STAMP(i)
#define AO_STAMP_FLUSH()
#define AO_LAB_ARGS
#endif

namespace qeft {

constexpr int AO_NW = 8;
constexpr int AO_D = 12;                    // ring slots per wave: D - 1 >= the loads of a wave of an attention block at K <= 5120 (4 GEMV waves, one row set): all in flight while the attention runs
constexpr int AO_MAX_RSC = 2;               // row sets per block (hidden 4096: 1; 5120: 2 on 64 of the 256 blocks)
constexpr uint32_t AO_ST_TIMEOUT = 0x7300;

struct AoArgs {
    // attention (rope_attn_decode_body)
    const int* pos; const int* out_pos; const f16* q; const f16* k; const f16* v; const float* cs; const float* sn;
    uint32_t heads_kv_s_tab; f16* kc; f16* vc; int max_seq; int n_attn_blocks;
    // o_proj
    const uint8_t* qw; const uint8_t* szp; const uint8_t* ow;
    float* h32; const f16* gamma_out; f16* ynorm; float* ssq_out;
    int K, nsets, nblk;
    // edge
    f16* xatt;                              // [K] the attention output = o_proj's x (written and read inside the launch)
    uint32_t* state;                        // [0] sequence word, [1] finish counter, [2] status (first give-up code), [3] attention blocks counted in
    uint32_t timeout_ticks;
    AO_LAB_ARGS
};

struct AoLds { uint32_t attn, xs, szl, owl, epl, red, ring, total; };
__host__ __device__ inline AoLds ao_lds(int K, int max_seq) {
    AoLds L; uint32_t o = 0;
    L.attn = o; o += ((uint32_t)rope_attn_smem_bytes(max_seq) + 1023u) / 1024u * 1024u;
    L.xs = o;   o += (uint32_t)v3_x_bytes(K);
    L.szl = o;  o += (uint32_t)AO_MAX_RSC * (uint32_t)v3_sz_bytes(K >> 7);
    L.owl = o;  o += (uint32_t)AO_MAX_RSC * 4096u;
    L.epl = o;  o += 1024u;
    L.red = o;  o += 2048u;                 // [RSC][NW][16] floats
    L.ring = o; o += (uint32_t)AO_NW * AO_D * 1024u;
    L.total = o;
    return L;
}

__device__ __forceinline__ void ao_dma16_nt(const void* gsrc, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int PRE, int DH>
__global__ __launch_bounds__(AO_NW * 64) void attn_oproj_kernel(AoArgs a) {
    constexpr int NW = AO_NW, D = AO_D;
    extern __shared__ __attribute__((aligned(1024))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nl = lane & 15, kc = lane >> 4;
    const int K = a.K, nfull = (K >> 7) - 1;
    const AoLds L = ao_lds(K, a.max_seq);
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    uint8_t* const xs = smem + L.xs;
    const uint8_t* const szl = smem + L.szl;
    const uint8_t* const owl = smem + L.owl;
    const uint8_t* const epl = smem + L.epl;
    float* const red = (float*)(smem + L.red);
    uint32_t seq;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(seq) : "s"(a.state) : "memory");
    AO_STAMP_DECL
    AO_STAMP(0);
    const int pos = *a.pos;
    const bool pos_ok = pos >= 0 && pos < a.max_seq;           // (grid-uniform; the engine never launches outside the cache)

    // ---- roles.  An attention block leaves waves 0..3 to the attention body and runs its o_proj share on waves 4..7 alone, which
    //      request the operands WHILE the attention runs (a wave's loads return in order: weights requested by the attention waves
    //      would sit in front of their K/V loads, and requested after the body they put the whole o_proj behind the attention).
    const bool is_attn = (int)blockIdx.x < a.n_attn_blocks;
    const int NWg = is_attn ? 4 : NW;                           // the waves of this block that run GEMV steps
    const int gw = is_attn ? wave - 4 : wave;                   // this wave's index among them (< 0: none)

    // ---- this block's o_proj share: row set lb, and -- hidden > 4096 -- one of the nsets - nblk extra sets, dealt from the LAST
    //      block down so that the attention blocks (the first ones) keep the short share
    const int s0 = v3_xcd_block(blockIdx.x, a.nblk);
    const int s1 = a.nblk + (a.nblk - 1 - (int)blockIdx.x);
    const int RS = s1 < a.nsets ? 2 : 1;
    auto set_of = [&](int rs) { return rs == 0 ? s0 : s1; };
    const V3Geom G{K, 128, K >> 7, nfull, K >> 7, a.nsets};
    const int SZB = v3_sz_bytes(G.ngroups), SPS = SZB >> 10;
    const int nsw = gw >= 0 ? (nfull - gw + NWg - 1) / NWg : 0;
    const uint32_t set_bytes = (uint32_t)K * 8u;
    const uint32_t lane_off = v3_w_lane_off(G, nl, kc);
    const uint32_t ring0 = __builtin_amdgcn_readfirstlane(lds0 + L.ring + (uint32_t)wave * (D * 1024u));
    const uint8_t* const ringp = smem + L.ring + (size_t)wave * (D * 1024) + (size_t)lane * 16;
    // issue cursor: load c of the wave's sequence (step-major over the RS row sets)
    const uint8_t* is_ptr = a.qw + (size_t)(gw < 0 ? 0 : gw) * 256u + lane_off;
    const size_t set_off0 = (size_t)s0 * set_bytes, set_off1 = (size_t)(RS > 1 ? s1 : s0) * set_bytes;
    int is_rs = 0, is_left = nsw;
    auto issue_next = [&](uint32_t slot_lds) {
        if (is_left <= 0) return;
        ao_dma16_nt(is_ptr + (is_rs == 0 ? set_off0 : set_off1), slot_lds);
        if (++is_rs == RS) {
            is_rs = 0;
            is_ptr += NWg * 256;
            --is_left;
        }
    };
    auto prefetch = [&]() {
        // block-constant operands: scale words, outlier rows, residual / gamma of the block's rows (a flat list of 1 KB pieces)
        const int PPS = SPS + 4, total = RS * PPS + 1;
        for (int t = gw; t < total; t += NWg) {
            const int rs = t / PPS, j = t - rs * PPS;
            if (rs >= RS) {
                const uint8_t* src;
                if (lane < 32) {
                    const int l = lane < 4 * RS ? lane : 0;
                    src = (const uint8_t*)a.h32 + ((size_t)set_of(l >> 2) * 16 + (size_t)(l & 3) * 4) * 4;
                } else {
                    const int l0 = (lane - 32) & 15, l = l0 < 2 * RS ? l0 : 0;
                    src = (const uint8_t*)a.gamma_out + ((size_t)set_of(l >> 1) * 16 + (size_t)(l & 1) * 8) * 2;
                }
                v3_dma16(src, __builtin_amdgcn_readfirstlane(lds0 + L.epl));
            } else if (j < SPS) {
                v3_dma16(a.szp + v3_sz_off(G, set_of(rs), j, lane), __builtin_amdgcn_readfirstlane(lds0 + L.szl + (uint32_t)rs * SZB + ((uint32_t)j << 10)));
            } else {
                v3_dma16(a.ow + v3_ow_off(set_of(rs), j - SPS, lane), __builtin_amdgcn_readfirstlane(lds0 + L.owl + (uint32_t)rs * 4096u + ((uint32_t)(j - SPS) << 10)));
            }
        }
        for (int d = 0; d < D - 1; ++d) issue_next(ring0 + (uint32_t)d * 1024u);
    };

    // ---- every GEMV wave requests its operands; the attention body runs beside them (two block barriers inside, S == 1)
    if (gw >= 0) prefetch();
    AO_STAMP(1);
    if (is_attn && pos_ok) {
        if (wave < 4) {
            rope_attn_decode_body<PRE, DH>(smem + L.attn, (int)blockIdx.x, a.pos, a.out_pos, a.q, a.k, a.v, a.cs, a.heads_kv_s_tab, a.sn, a.kc,
                                           a.vc, nullptr, nullptr, a.max_seq, nullptr, nullptr, a.xatt);
        } else {
            __builtin_amdgcn_s_barrier();       // the body's two block barriers (rotary staged; partials written)
            __builtin_amdgcn_s_barrier();
        }
    }

    // ---- edge.  Wave 0 of an attention block: its 64 output elements are on their way (write-through stores); once they are
    //      acknowledged, count the block in (a launch that skipped the body counts too: the counter stays in step with the
    //      sequence word).  Wave 0 of every block: wait for the count, drop what this XCD's L2 holds (lines of x part-written
    //      from here), then all 8 waves fetch x into LDS as any GEMV stages it -- through the L2, once per XCD from memory.
    AO_STAMP(2);
    if (wave == 0) {
        if (is_attn) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add(a.state + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const uint32_t target = (seq + 1u) * (uint32_t)a.n_attn_blocks;
        const long long t0 = wall_clock64();
        for (unsigned spins = 0;; ++spins) {
            const uint32_t c = __builtin_amdgcn_readfirstlane(__hip_atomic_load(a.state + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            if ((int)(c - target) >= 0) break;
            if ((spins & 7u) == 7u) {
                if (wall_clock64() - t0 > (long long)a.timeout_ticks ||
                    __hip_atomic_load(a.state + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                    if (lane == 0) atomicCAS(a.state + 2, 0u, AO_ST_TIMEOUT | (uint32_t)(blockIdx.x & 0xff));
                    break;
                }
            }
            __builtin_amdgcn_s_sleep(4);
        }
        AO_STAMP(3);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");            // released; this wave's staging and ring have landed
    AO_STAMP(4);
    {
        const int xbytes = K * 2;                               // a multiple of 256
        for (int p = wave; p * 1024 < xbytes; p += NW) {
            const int off = p * 1024 + lane * 16;
            v3_dma16((const uint8_t*)a.xatt + (off < xbytes ? off : xbytes - 16), __builtin_amdgcn_readfirstlane(lds0 + L.xs + (uint32_t)p * 1024u));
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");            // x and every wave's staging are in LDS
    AO_STAMP(5);

    // ---- the GEMV steps (gemv_v3.h's arithmetic; ring slots in LDS, software-pipelined as tools/gemv_chain_lab.h)
    uint32_t MAGIC = 0x64006400u, NEG1024 = 0xE400E400u;
    asm volatile("" : "+v"(MAGIC), "+v"(NEG1024));
    const v3h8 c8 = __builtin_bit_cast(v3h8, u32x4{NEG1024, NEG1024, NEG1024, NEG1024});
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    float acc[AO_MAX_RSC];
#pragma unroll
    for (int r = 0; r < AO_MAX_RSC; ++r) acc[r] = 0.f;
    float acc_tail = 0.f, acc_out = 0.f;
    const uint8_t* xa = xs + kc * 64;
    if (gw == nfull % NWg) {                    // the fp16 outlier columns [K - 128, K)
        const v3h8* px = (const v3h8*)(xa + (size_t)nfull * 256);
        v3h8 xo[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) xo[jj] = px[jj];
#pragma unroll
        for (int rs = 0; rs < AO_MAX_RSC; ++rs)
            if (rs < RS) {
                f32x4 Pm = z4;
                const uint8_t* prow = owl + rs * 4096 + nl * 256;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                    Pm = __builtin_amdgcn_mfma_f32_16x16x32_f16(xo[jj], *(const v3h8*)(prow + (((kc * 4 + jj) ^ nl) & 15) * 16), Pm, 0, 0, 0);
                if (rs == RS - 1) acc_out = Pm[0]; else acc[rs] = Pm[0];
            }
    }
    uint32_t slot = 0, refill = D - 1;
    if (nsw > 0) {
        struct XF { v3h8 f[4]; float lo, hi; };
        const uint8_t* xp = xa + (size_t)gw * 256;
        const uint8_t* sp = szl + (size_t)gw * 64 + nl * 4;
        const int xstep = NWg * 256, sstep = NWg * 64;
        XF XA, XB;
        auto load_xf = [&](XF& X, const uint8_t* ptr) {
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
        u32x4 wv = *(const u32x4*)(ringp + (size_t)slot * 1024);
        uint32_t szw = *(const uint32_t*)sp;
        f32x4 pPlo = z4, pPhi = z4;
        uint32_t pszw = 0;
        auto fold = [&](float& dst, float lo, float hi) {
            const h2 sz2 = as_h2(pszw);
            dst = dst + ((float)sz2[0] * ((pPlo[0] + lo) + 0.0625f * (pPhi[0] + hi)) + (float)sz2[1] * ((lo + hi) * -0.0009765625f));
        };
        auto step = [&](XF& C, XF& N, bool more) {
            f32x4 B0 = z4, B1 = z4;
#pragma unroll
            for (int rs = 0; rs < AO_MAX_RSC; ++rs) {
                if (rs < RS) {
                    const bool last = rs == RS - 1;
                    if (rs == 0 && more) load_xf(N, xp + xstep);
                    issue_next(ring0 + refill * 1024u);
                    refill = slot;
                    slot = slot + 1 == D ? 0 : slot + 1;
                    u32x4 bf[4];
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const uint32_t q = wv[w], t = q >> 8;
                        bf[0][w] = (q & 0x000f000fu) | MAGIC;
                        bf[1][w] = (q & 0x00f000f0u) | MAGIC;
                        bf[2][w] = (t & 0x000f000fu) | MAGIC;
                        bf[3][w] = (t & 0x00f000f0u) | MAGIC;
                    }
                    f32x4 Plo = __builtin_amdgcn_mfma_f32_16x16x32_f16(C.f[0], __builtin_bit_cast(v3h8, bf[0]), z4, 0, 0, 0);
                    f32x4 Phi = __builtin_amdgcn_mfma_f32_16x16x32_f16(C.f[1], __builtin_bit_cast(v3h8, bf[1]), z4, 0, 0, 0);
                    Plo = __builtin_amdgcn_mfma_f32_16x16x32_f16(C.f[2], __builtin_bit_cast(v3h8, bf[2]), Plo, 0, 0, 0);
                    Phi = __builtin_amdgcn_mfma_f32_16x16x32_f16(C.f[3], __builtin_bit_cast(v3h8, bf[3]), Phi, 0, 0, 0);
                    // the NEXT consume's ring slot: its load is D - 2 issues back (everything when the cursor has run out)
                    if (is_left > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D - 2) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    const u32x4 wv_n = *(const u32x4*)(ringp + (size_t)slot * 1024);
                    const uint32_t szw_n = last ? *(const uint32_t*)(sp + (more ? sstep : 0)) : *(const uint32_t*)(sp + (size_t)(rs + 1) * SZB);
                    if (rs == 0) fold(acc_tail, N.lo, N.hi); else fold(acc[rs - 1], C.lo, C.hi);
                    if (last && more) {
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
                xp += xstep;
                sp += sstep;
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
    if (kc == 0) {
#pragma unroll
        for (int rs = 0; rs < AO_MAX_RSC; ++rs)
            if (rs < RS) red[(rs * NW + wave) * 16 + nl] = rs == RS - 1 ? acc_tail : acc[rs];
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

    AO_STAMP(6);
    // ---- epilogue (gemv_v3.h, PLAIN + residual + gamma_out): wave 0, lane = row of the block
    if (wave == 0) {
        const int rs = lane >> 4, n = lane & 15;
        const bool act = lane < RS * 16;
        float v = 0.f;
        if (act) {
#pragma unroll
            for (int w = 0; w < NW; ++w) v += red[(rs * NW + w) * 16 + n];
        }
        v += ((const float*)epl)[lane];
        const int row = set_of(rs) * 16 + n;
        float sq = 0.f;
        if (act) {
            a.h32[row] = v;
            a.ynorm[row] = (f16)(v * (float)((const f16*)(epl + 512))[lane]);
            sq = v * v;
        }
        sq = wave_sum(sq);
        if (lane == 0) a.ssq_out[blockIdx.x] = sq;
    }
    // ---- the sequence word: the last block to finish bumps it (every block has read it at entry)
    __syncthreads();
    AO_STAMP_FLUSH();
    if (tid == 0) {
        const unsigned done = atomicAdd(a.state + 1, 1u);
        if (done == gridDim.x - 1) {
            a.state[1] = 0;
            __hip_atomic_store(a.state, seq + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---- launch
bool attn_oproj_supported(int n_heads, int n_kv, int max_seq, int hidden, int k_in, int n_out, int group_size) {
    const int nsets = hidden / 16;
    const int nblk = nsets < 256 ? nsets : 256;
    return n_heads >= 1 && n_kv >= 1 && n_heads % n_kv == 0 && n_heads * 2 <= nblk && hidden % 16 == 0 && k_in == n_heads * 128 && k_in % 128 == 0 &&
           k_in >= 256 && k_in <= 8192 && n_out == 128 && group_size == 128 && (nsets + nblk - 1) / nblk <= AO_MAX_RSC && max_seq % 16 == 0 &&
           max_seq >= 16 && ao_lds(k_in, max_seq).total <= 160 * 1024;
}

hipError_t attn_oproj_launch(const void* q, const void* k, const void* v, const void* cs, const void* sn, int tab_rows, void* kc, void* vc,
                             const int* pos, const int* out_pos, int n_heads, int n_kv, int max_seq, const void* qweight, const void* sz_packed,
                             const void* oweight, void* h32, const void* gamma_out, void* ynorm, float* ssq_out, int hidden, int k_in,
                             void* xatt, void* state, hipStream_t st) {
    AoArgs a{};
    a.pos = pos; a.out_pos = out_pos; a.q = (const f16*)q; a.k = (const f16*)k; a.v = (const f16*)v;
    a.cs = (const float*)cs; a.sn = (const float*)sn;
    a.heads_kv_s_tab = (uint32_t)n_heads | ((uint32_t)n_kv << 12) | (1u << 24) | ((tab_rows == 1 ? 1u : 0u) << 28);
    a.kc = (f16*)kc; a.vc = (f16*)vc; a.max_seq = max_seq; a.n_attn_blocks = n_heads * 2;
    a.qw = (const uint8_t*)qweight; a.szp = (const uint8_t*)sz_packed; a.ow = (const uint8_t*)oweight;
    a.h32 = (float*)h32; a.gamma_out = (const f16*)gamma_out; a.ynorm = (f16*)ynorm; a.ssq_out = ssq_out;
    a.K = k_in; a.nsets = hidden / 16;
    a.nblk = a.nsets < 256 ? a.nsets : 256;
    a.xatt = (f16*)xatt; a.state = (uint32_t*)state;
    a.timeout_ticks = 100u * 1000u * 50u;       // 50 ms
    const size_t smem = ao_lds(k_in, max_seq).total;
    auto kern = attn_oproj_kernel<4, 2>;
    if (smem > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(a.nblk), dim3(AO_NW * 64), smem, st, a);
    return hipGetLastError();
}

}  // namespace qeft
