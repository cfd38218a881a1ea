// Weight-stationary skinny GEMM for 17 .. 64 rows (round 4): y[M][N] = x[M][K] . Wdeq^T (+ bias) with ONE pass over the packed
// weights.  The reference serves these sizes with its own tuned tiers (gemm_cuda.cu:952-978: M <= 32 with split-K 2, M <= 64);
// benchmark.py:118,298-299's 64-token prompt lands here.  Before this kernel M = 17 .. 64 went to the 128 x 128 MFMA tiles plus a
// split-K combine (two launches, 15.8 / 21.7 / 24.5 us at M = 64 on the three Llama-2-7B shapes).
//
// Shape of the computation = the decode GEMV's (gemv_v3.h), with the batch rows as A rows of the same MFMAs:
//   * a block owns RSC 16-row sets of W and a CHUNK of 16 MC batch rows (MC = 1, 2; blockIdx.y = chunk); wave w owns the 128-k
//     steps w, w + 8, ..; a wave-wide 16-byte load of the checkpoint layout is the four B fragments of a step (1024 + q nibble
//     trick, bias removed by -1024 MFMAs), unpacked ONCE and used by the MC A fragments;
//   * x is read from global memory COALESCED -- one wave-wide 16-byte load covers 4 batch rows x 256 contiguous bytes (a step's
//     128 k), two or three steps ahead of its use -- and turned into A fragments (16 batch rows x 32 k per MFMA) through a
//     wave-private 4 KB LDS stage per chunk (units XOR-swizzled by the row: conflict-free both ways).  The first version read the
//     fragments directly (lane = row x 32-k chunk: 16-byte pieces 64 bytes apart over 16 rows, 32 cache lines per instruction)
//     and spent its time in the address coalescer: M = 32 on 4096^2 took 15.8 us.  No x is kept in LDS, so any K fits;
//   * scales / scaled_zeros in the CHECKPOINT layout fp16 [K/g][N] are staged raw by LDS-DMA and packed LDS -> LDS (gemv_v3.h's
//     V3_F_SZN phase), the outlier slice comes from the plain oweight [N][128] rows, swizzled in LDS;
//   * every load is compiler-visible: the schedule is fully static (ring slots, row sets and fragment sets are compile-time
//     indices, rounds of lcm-many steps), so hipcc places every s_waitcnt itself.
// Grid: ceil(nsets / RSC) x ceil(M / (16 MC)) blocks, chosen so that about 256 blocks exist (one per CU); blocks of one row-set
// group sit gridDim.x apart in launch order, i.e. on the same XCD when gridDim.x % 8 == 0: the second chunk's weights are L2 hits.
#include <cstdlib>

#include "gemv_v3.h"

namespace qeft {

struct WsArgs {
    const f16* x;            // [m][K]
    const uint8_t* qw;       // int16 [N/4][K]
    const f16* scales;       // fp16 [K/128][N]
    const f16* zeros;        // fp16 [K/128][N] (scaled zeros)
    const uint8_t* ow;       // fp16 [N][128] plain outlier rows (NULL: no outlier slice)
    const f16* bias;         // optional [N]
    f16* y;                  // [m][N]
    int m, K, nsets;         // rows, in_features, N / 16
    int nblk, sets_q, sets_r;    // gridDim.x and the deal of the row sets over it
};

constexpr int WS_NW = 8, WS_D = 4;

__host__ __device__ inline uint32_t ws_red_bytes(int rsc, int mc) { return (uint32_t)rsc * WS_NW * 16 * mc * 16 * 4; }
struct WsLds { uint32_t szraw, szl, owl, red, xst, total; };
__host__ __device__ inline WsLds ws_lds(int K, int rsc, int mc, bool outl) {
    WsLds L; uint32_t o = 0;
    const int ng = K >> 7;
    L.szraw = o; o += 2u * rsc * v3_szraw_bytes(ng);
    L.szl = o;   o += (uint32_t)rsc * v3_sz_bytes(ng);
    L.owl = o;   o += outl ? (uint32_t)rsc * 4096u : 0u;
    // the waves' x stages (per wave and chunk: 16 rows x 256 bytes of one step) and, once the steps are done, the partial sums
    const uint32_t red = ws_red_bytes(rsc, mc), xst = (uint32_t)WS_NW * mc * 4096u;
    L.red = o;
    L.xst = o;   o += red > xst ? red : xst;
    L.total = o;
    return L;
}

template <int N, typename F> __device__ __forceinline__ void ws_for(F&& f) { v3_static_for<0, N>(f); }

template <int RSC, int MC, bool OUTL>
__global__ __launch_bounds__(WS_NW * 64) void gemm_ws_kernel(WsArgs a) {
    constexpr int NW = WS_NW, D = WS_D;
    constexpr int XP = RSC >= 3 ? 2 : (MC == 2 ? 3 : 4), XD = XP - 1;      // x sets: loaded XD steps ahead (three or four row sets per step: one step is time enough, and the registers are needed)
    typedef float accv __attribute__((ext_vector_type(4 * MC)));   // a lane's batch rows 4 kc .. 4 kc + 3 of each 16-row A chunk
    extern __shared__ __attribute__((aligned(1024))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nl = lane & 15, kc = lane >> 4;
    const int K = a.K, nfull = (K >> 7) - (OUTL ? 1 : 0), ngroups = K >> 7;
    const V3Geom G{K, OUTL ? 128 : 0, K >> 7, nfull, ngroups, a.nsets};
    const WsLds L = ws_lds(K, RSC, MC, OUTL);
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    uint8_t* const szl = smem + L.szl;
    const uint8_t* const owl = smem + L.owl;
    float* const red = (float*)(smem + L.red);
    const int SZB = v3_sz_bytes(ngroups), SRB = v3_szraw_bytes(ngroups), SPR = SRB >> 10;
    int set0, RS;
    v3_block_sets(v3_xcd_block(blockIdx.x, a.nblk), a.sets_q, a.sets_r, set0, RS);
    auto set_of = [&](int rs) { return set0 + (rs < RS ? rs : RS - 1); };      // a short block's last slot repeats its last set
    const int row0 = blockIdx.y * (16 * MC);
    const int m = min(16 * MC, a.m - row0);                     // batch rows of this block

    // ---- 1. weight ring: D loads per wave, step-major over the block's row sets (gemv_v3.h step 2)
    const int nsw = (nfull - wave + NW - 1) / NW;
    const uint32_t set_bytes = (uint32_t)K * 8u;
    const uint8_t* const wbase = a.qw + v3_w_set_off(G, set0);
    const uint32_t lane_off = v3_w_lane_off(G, nl, kc);
    const uint32_t step0 = min((uint32_t)wave * 256u, v3_last_step_off(G));
    const uint32_t step_last = step0 + (uint32_t)(nsw > 0 ? nsw - 1 : 0) * (NW * 256u);
    const uint32_t short_back = RS < RSC ? set_bytes : 0u;
    // Every block reads the SAME x rows: were all CUs to walk the steps in the same order, they would all pull the same few cache
    // lines of x from the same L2 channels at the same moment.  Each block therefore starts its waves' step sequences at an offset
    // of its own and wraps around (the sum over k does not care about the order): rot = the wave's first step index.
    const uint32_t rot = nsw > 0 ? (uint32_t)(blockIdx.x * 3 + blockIdx.y * 5 + wave) % (uint32_t)nsw : 0u;
    u32x4 ring[D];
    uint32_t p_step = step0 + rot * (NW * 256u);
    auto issue = [&](u32x4& b, auto rs_tag) {
        constexpr int rs = decltype(rs_tag)::value;
        const uint32_t off = p_step + (uint32_t)rs * set_bytes - (rs == RSC - 1 ? short_back : 0u);
        b = __builtin_nontemporal_load((const u32x4*)(wbase + off + lane_off));
        if (rs == RSC - 1) p_step = p_step >= step_last ? step0 : p_step + NW * 256u;
    };
    ws_for<D>([&](auto d) {
        issue(ring[d], std::integral_constant<int, decltype(d)::value % RSC>{});
        __builtin_amdgcn_sched_barrier(0);
    });

    // ---- 2. staging by LDS-DMA, BEHIND the ring's first loads (they are what the stream waits for): raw scale / zero rows of the
    //         block's sets, outlier rows.  The only hidden loads of the kernel, drained here, in front of everything hipcc counts.
    for (int rs = 0; rs < RSC; ++rs) {
        const int set = set_of(rs);
        for (int t = wave; t < 2 * SPR; t += NW) {
            const int arr = t >= SPR ? 1 : 0, j = t - arr * SPR;
            v3_dma16((const uint8_t*)(arr ? a.zeros : a.scales) + v3_szn_off(G, set, j, lane),
                     __builtin_amdgcn_readfirstlane(lds0 + L.szraw + (uint32_t)(arr * RSC + rs) * SRB + ((uint32_t)j << 10)));
        }
        if (OUTL && wave >= NW - 4)
            v3_dma16(a.ow + v3_ow_off(set, wave - (NW - 4), lane), __builtin_amdgcn_readfirstlane(lds0 + L.owl + (uint32_t)rs * 4096u + ((uint32_t)(wave - (NW - 4)) << 10)));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- 3. the staged rows are in LDS: pack (scale | scaled_zero << 16) words, [rs][group][16 rows]
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    {
        const uint8_t* const sraw = smem + L.szraw;
        const uint8_t* const zraw = sraw + (size_t)RSC * SRB;
        for (int rs = 0; rs < RSC; ++rs)
            for (int q = tid; q < ngroups * 4; q += NW * 64) {
                const u32x2 sv = *(const u32x2*)(sraw + (size_t)rs * SRB + (size_t)q * 8);
                const u32x2 zv = *(const u32x2*)(zraw + (size_t)rs * SRB + (size_t)q * 8);
                *(u32x4*)(szl + (size_t)rs * SZB + (size_t)q * 16) =
                    u32x4{(sv[0] & 0xffffu) | (zv[0] << 16), (sv[0] >> 16) | (zv[0] & 0xffff0000u),
                          (sv[1] & 0xffffu) | (zv[1] << 16), (sv[1] >> 16) | (zv[1] & 0xffff0000u)};
            }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");          // LDS only: the ring stays in flight

    // ---- 4. steps
    accv acc[RSC];
    ws_for<RSC>([&](auto r) { acc[decltype(r)::value] = accv{}; });
    uint32_t MAGIC = 0x64006400u, NEG1024 = 0xE400E400u;
    asm volatile("" : "+v"(MAGIC), "+v"(NEG1024));
    const v3h8 c8 = __builtin_bit_cast(v3h8, u32x4{NEG1024, NEG1024, NEG1024, NEG1024});
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    // coalesced x loads: instruction q of chunk c covers batch rows 16 c + 4 q .. + 3 (this lane: row 4 q + kc, 16-byte unit nl of the
    // step's 256 bytes); rows past the end re-read the last row
    const uint8_t* xrow[MC][4];
    ws_for<MC>([&](auto c) {
        constexpr int cc = decltype(c)::value;
#pragma unroll
        for (int q = 0; q < 4; ++q) xrow[cc][q] = (const uint8_t*)a.x + (size_t)(row0 + min(16 * cc + 4 * q + kc, m - 1)) * K * 2 + nl * 16;
    });
    // the wave's stage: row r's unit u sits at r * 256 + ((u ^ r) & 15) * 16
    uint8_t* const xst = smem + L.xst + (size_t)wave * (MC * 4096);
    // raw (as loaded) -> A fragments of the MFMAs: lane (batch row nl, 32-k chunk kc), fragment j = unit 4 kc + j
    auto to_frags = [&](const v3h8 (&raw)[MC][4], v3h8 (&fr)[MC][4]) {
        ws_for<MC>([&](auto c) {
            constexpr int cc = decltype(c)::value;
#pragma unroll
            for (int q = 0; q < 4; ++q) *(v3h8*)(xst + cc * 4096 + (4 * q + kc) * 256 + ((nl ^ (4 * q + kc)) & 15) * 16) = raw[cc][q];
        });
        ws_for<MC>([&](auto c) {
            constexpr int cc = decltype(c)::value;
#pragma unroll
            for (int j = 0; j < 4; ++j) fr[cc][j] = *(const v3h8*)(xst + cc * 4096 + nl * 256 + (((4 * kc + j) ^ nl) & 15) * 16);
        });
    };
    auto join = [&](const f32x4 (&p)[MC]) {
        if constexpr (MC == 1) return p[0];
        else return __builtin_shufflevector(p[0], p[1], 0, 1, 2, 3, 4, 5, 6, 7);
    };

    // the fp16 outlier columns [K - 128, K): one MFMA step per row set and chunk, by the wave whose turn step nfull would be
    if (OUTL && wave == nfull % NW) {
        v3h8 xraw[MC][4], xo[MC][4];
        ws_for<MC>([&](auto c) {
            constexpr int cc = decltype(c)::value;
#pragma unroll
            for (int q = 0; q < 4; ++q) xraw[cc][q] = *(const v3h8*)(xrow[cc][q] + (size_t)nfull * 256);
        });
        to_frags(xraw, xo);
        ws_for<RSC>([&](auto rs_tag) {
            constexpr int rs = decltype(rs_tag)::value;
            const uint8_t* prow = owl + rs * 4096 + nl * 256;
            v3h8 bo[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) bo[jj] = *(const v3h8*)(prow + (((kc * 4 + jj) ^ nl) & 15) * 16);
            f32x4 P[MC];
            ws_for<MC>([&](auto c) {
                constexpr int cc = decltype(c)::value;
                P[cc] = z4;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) P[cc] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xo[cc][jj], bo[jj], P[cc], 0, 0, 0);
            });
            acc[rs] = acc[rs] + join(P);
        });
    }

    if (nsw > 0) {
        v3h8 xr[XP][MC][4];
        const uint32_t xoff0 = (uint32_t)wave * 256u;
        const uint32_t xoff_last = xoff0 + (uint32_t)(nsw - 1) * (NW * 256u);
        uint32_t xq = xoff0 + rot * (NW * 256u);                // byte offset (inside a row) of the step the next x load reads
        auto load_x = [&](v3h8 (&o)[MC][4]) {
            ws_for<MC>([&](auto c) {
                constexpr int cc = decltype(c)::value;
#pragma unroll
                for (int q = 0; q < 4; ++q) o[cc][q] = *(const v3h8*)(xrow[cc][q] + xq);
            });
            xq = xq >= xoff_last ? xoff0 : xq + NW * 256u;
        };
        ws_for<XD>([&](auto d) { load_x(xr[decltype(d)::value]); });
        const uint8_t* const sp0 = szl + (size_t)wave * 64 + nl * 4;
        const uint8_t* const sp_last = sp0 + (size_t)(nsw - 1) * (NW * 64);
        const uint8_t* sp = sp0 + (size_t)rot * (NW * 64);
        // -1024 S_lo, -1024 S_hi of the step's fragments per chunk (S = the sum of x over the low- / high-nibble positions): whole
        // MFMA results, the C operand of the step's product MFMAs (gemv_v3.h step 4) -- the products come out without the 1024
        // bias and the fold is three FMAs per value; zB = -1024 sum(x) of the step for the zero-point term
        f32x4 A0[MC], A1[MC];
        accv zB = accv{};
        auto bias_sums = [&](const v3h8 (&x4)[MC][4]) {
            ws_for<MC>([&](auto c) {
                constexpr int cc = decltype(c)::value;
                A0[cc] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x4[cc][0], c8, z4, 0, 0, 0);
                A1[cc] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x4[cc][1], c8, z4, 0, 0, 0);
                A0[cc] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x4[cc][2], c8, A0[cc], 0, 0, 0);
                A1[cc] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x4[cc][3], c8, A1[cc], 0, 0, 0);
            });
            zB = join(A0) + join(A1);
        };
        auto splat = [&](float v) {
            if constexpr (MC == 1) return accv{v, v, v, v};
            else return accv{v, v, v, v, v, v, v, v};
        };
        // one (step, row set): xc = the step's fragments, xld = the set the fragments of the step XD ahead are loaded into
        v3h8 xf[MC][4];                                         // the current step's A fragments
        auto consume = [&](u32x4& slot, v3h8 (&xraw)[MC][4], v3h8 (&xld)[MC][4], auto rs_tag, bool more_steps) {
            constexpr int rs = decltype(rs_tag)::value;
            if (rs == 0) {
                to_frags(xraw, xf);
                load_x(xld);
                bias_sums(xf);
            }
            v3h8 (&xc)[MC][4] = xf;
            const uint32_t szw = *(const uint32_t*)(sp + (size_t)rs * SZB);
            const u32x4 wv = slot;
            u32x4 bf[4];
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const uint32_t v = wv[w], t = v >> 8;
                bf[0][w] = (v & 0x000f000fu) | MAGIC;
                bf[1][w] = (v & 0x00f000f0u) | MAGIC;
                bf[2][w] = (t & 0x000f000fu) | MAGIC;
                bf[3][w] = (t & 0x00f000f0u) | MAGIC;
            }
            issue(slot, std::integral_constant<int, (rs + D) % RSC>{});      // the slot's next load
            f32x4 Plo[MC], Phi[MC];
            ws_for<MC>([&](auto c) {
                constexpr int cc = decltype(c)::value;
                Plo[cc] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xc[cc][0], __builtin_bit_cast(v3h8, bf[0]), A0[cc], 0, 0, 0);
                Phi[cc] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xc[cc][1], __builtin_bit_cast(v3h8, bf[1]), A1[cc], 0, 0, 0);
                Plo[cc] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xc[cc][2], __builtin_bit_cast(v3h8, bf[2]), Plo[cc], 0, 0, 0);
                Phi[cc] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xc[cc][3], __builtin_bit_cast(v3h8, bf[3]), Phi[cc], 0, 0, 0);
            });
            const h2 sz2 = as_h2(szw);
            acc[rs] = __builtin_elementwise_fma(splat((float)sz2[0]), __builtin_elementwise_fma(splat(0.0625f), join(Phi), join(Plo)), acc[rs]);
            acc[rs] = __builtin_elementwise_fma(splat((float)sz2[1] * -0.0009765625f), zB, acc[rs]);
            if (rs == RSC - 1) sp = sp >= sp_last ? sp0 : sp + NW * 64;
        };
        // the unrolled round: a whole number of ring turns and of fragment-set turns
        constexpr int US0 = v3_unroll_steps(D, RSC);
        constexpr int US = US0 * XP / v3_gcd(US0, XP);
        static_assert((US * RSC) % D == 0 && US % XP == 0, "a round returns every ring slot and fragment set to its role");
        int i = 0;
        auto round = [&](bool guarded) {
            v3_static_for<0, US>([&](auto u_tag) {
                constexpr int u = decltype(u_tag)::value;
                if (!guarded || i + u < nsw) {
                    const bool more = i + u + 1 < nsw;
                    v3_static_for<0, RSC>([&](auto rs_tag) {
                        constexpr int rs = decltype(rs_tag)::value;
                        constexpr int c = u * RSC + rs;
                        consume(ring[c % D], xr[u % XP], xr[(u + XD) % XP], rs_tag, more);
                        __builtin_amdgcn_sched_barrier(0);
                    });
                }
            });
        };
        for (; i + US <= nsw; i += US) round(false);
        if (i < nsw) round(true);
    }

    // ---- 5. combine the waves: red[rs][wave][batch row][W row] (the region the x stages used: every wave is done with its own first)
    constexpr int RB = 16 * MC;
    __syncthreads();
    ws_for<RSC>([&](auto r) {
        constexpr int rs = decltype(r)::value;
        ws_for<MC>([&](auto c) {
            constexpr int cc = decltype(c)::value;
#pragma unroll
            for (int j = 0; j < 4; ++j) red[((size_t)(rs * NW + wave) * RB + 16 * cc + 4 * kc + j) * 16 + nl] = acc[rs][4 * cc + j];
        });
    });
    __syncthreads();
    // wave w finishes batch rows w, w + 8, ..; lane = (row set lane / 16, W row lane % 16)
    const int N = a.nsets * 16;
    for (int i = wave; i < m; i += NW) {
        for (int rs = lane >> 4; rs < RS; rs += 4) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) v += red[((size_t)(rs * NW + w) * RB + i) * 16 + nl];
            const int row = (set0 + rs) * 16 + nl;
            if (a.bias) v += (float)a.bias[row];
            a.y[(size_t)(row0 + i) * N + row] = (f16)v;
        }
    }
}

// ---- launch
bool gemm_ws_supported(int m, int n, int k, int group_size, int n_out) {
    return m > 16 && m <= 64 && n >= 16 && n % 16 == 0 && k % 128 == 0 && k >= 256 && group_size == 128 && (n_out == 0 || n_out == 128);
}

template <int RSC, int MC, bool OUTL>
static hipError_t ws_go(const WsArgs& a, int chunks, hipStream_t st) {
    auto kern = gemm_ws_kernel<RSC, MC, OUTL>;
    const size_t smem = ws_lds(a.K, RSC, MC, OUTL).total;
    if (smem > 160 * 1024) return hipErrorInvalidValue;
    if (smem > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(a.nblk, chunks), dim3(WS_NW * 64), smem, st, a);
    return hipGetLastError();
}

// Geometry.  Candidates: MC = 1, 2 fragments (16 / 32 batch rows per block) and 1 .. 6 row sets per block; the blocks should fill
// the 256 CUs in ONE round (a second round doubles the time: 4096^2 at M = 48 with 258 blocks took 16.2 us against 9.9).  Among
// those the time follows the length of a wave's dependent sequence -- (K / 1024) steps x (per-step transposition + RSC consumes),
// a two-fragment consume costing ~1.7 x a one-fragment one -- not the bytes: measured on the three Llama-2-7B shapes
// (tools/gpu_ws_sweep.sh, profiles/r04_gemm_ws.txt) t ~ 7.0 + 0.8 RSC (MC = 1) and 8.5 + 1.4 RSC (MC = 2) us at K = 4096, with a
// step up of 2 - 3 us from five row sets on (one-step x prefetch, more registers).  The plan takes the cheapest candidate.
void gemm_ws_plan(int m, int nsets, int k, bool outl, int& mc, int& rsc, int& nblk, int& chunks) {
    double best = 1e30;
    mc = 2; rsc = 1; nblk = nsets; chunks = (m + 31) / 32;
    static const int f_mc = getenv("QEFT_WS_MC") ? atoi(getenv("QEFT_WS_MC")) : 0, f_rsc = getenv("QEFT_WS_RSC") ? atoi(getenv("QEFT_WS_RSC")) : 0;     // lab
    if (f_mc >= 1 && f_mc <= 2 && f_rsc >= 1 && f_rsc <= 6) {
        mc = f_mc; chunks = (m + 16 * mc - 1) / (16 * mc);
        nblk = (nsets + f_rsc - 1) / f_rsc; rsc = (nsets + nblk - 1) / nblk;
        if (ws_lds(k, rsc, mc, outl).total <= 160 * 1024) return;
    }
    for (int c_mc = 1; c_mc <= 2; ++c_mc) {
        const int c_chunks = (m + 16 * c_mc - 1) / (16 * c_mc);
        for (int want = 1; want <= 6; ++want) {
            const int c_nblk = (nsets + want - 1) / want, c_rsc = (nsets + c_nblk - 1) / c_nblk;
            if (ws_lds(k, c_rsc, c_mc, outl).total > 160 * 1024) continue;          // (does not fit a CU's LDS)
            const int blocks = c_nblk * c_chunks, rounds = (blocks + 255) / 256;
            const double cost = rounds * (c_mc == 1 ? 7.0 + 0.8 * c_rsc + (c_rsc >= 5 ? 3.0 : 0.0) : 8.5 + 1.4 * c_rsc + (c_rsc >= 5 ? 2.0 : 0.0));
            if (cost < best) { best = cost; mc = c_mc; rsc = c_rsc; nblk = c_nblk; chunks = c_chunks; }
        }
    }
}

template <int MC, bool OUTL>
static hipError_t ws_rsc(const WsArgs& a, int rsc, int chunks, hipStream_t st) {
    switch (rsc) {
        case 1: return ws_go<1, MC, OUTL>(a, chunks, st);
        case 2: return ws_go<2, MC, OUTL>(a, chunks, st);
        case 3: return ws_go<3, MC, OUTL>(a, chunks, st);
        case 4: return ws_go<4, MC, OUTL>(a, chunks, st);
        case 5: return ws_go<5, MC, OUTL>(a, chunks, st);
        case 6: return ws_go<6, MC, OUTL>(a, chunks, st);
    }
    return hipErrorInvalidValue;
}

hipError_t gemm_ws_launch(const void* x, const void* qweight, const void* scales, const void* zeros, const void* oweight, const void* bias,
                          void* y, int m, int n, int k, int n_out, hipStream_t st) {
    WsArgs a{};
    a.x = (const f16*)x; a.qw = (const uint8_t*)qweight; a.scales = (const f16*)scales; a.zeros = (const f16*)zeros;
    a.ow = (const uint8_t*)oweight; a.bias = (const f16*)bias; a.y = (f16*)y;
    a.m = m; a.K = k; a.nsets = n / 16;
    int mc, rsc, chunks;
    gemm_ws_plan(m, a.nsets, k, n_out > 0, mc, rsc, a.nblk, chunks);
    a.sets_q = a.nsets / a.nblk;
    a.sets_r = a.nsets % a.nblk;
    g_last_variant = "gemm_ws";
    const bool outl = n_out > 0;
    if (mc == 1) return outl ? ws_rsc<1, true>(a, rsc, chunks, st) : ws_rsc<1, false>(a, rsc, chunks, st);
    return outl ? ws_rsc<2, true>(a, rsc, chunks, st) : ws_rsc<2, false>(a, rsc, chunks, st);
}

}  // namespace qeft
