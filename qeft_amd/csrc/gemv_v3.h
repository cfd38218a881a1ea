// Decode GEMV, third formulation ("v3"): the production GEMV -- the decode engine's launches (one packed operand per launch,
// derived buffers of qeft_amd/fuse.py) and, since round 3, the reference's own entry points gemv_4bit / gemv_4bit_qeft /
// QuantLinear.forward for m = 1..7 on the operands as the checkpoint holds them (see the list at the end of this comment).
//
// Same MFMA mapping as gemv_w4_mfma.h (one wave-wide 16 B/lane load = 16 rows x 128 k = four B fragments of
// v_mfma_f32_16x16x32_f16, x as the A operand from LDS), but the per-block prologue is gone.  Round 1 measured the launch as
// T = 4.3 us + bytes / 5.9 TB/s with no wasted traffic: the fixed cost was ~450 prologue instructions every one of the
// 256-768 blocks executed (x pre-scaling, per-step correction sums, RMSNorm / SiLU on the staged vector, the outlier-slab
// de-interleave) in front of a staging barrier.  Here:
//
//   * x is consumed RAW.  Nibble pairs still become 1024 + q / 1024 + 16 q with one v_and_or_b32 each, but instead of
//     pre-scaling x and pre-computing per-step sums, the bias is removed on the otherwise idle matrix pipe: with the
//     constant fragment c = -1024,
//         P_lo = x_lo.(1024 + q),  A_lo = x_lo.c = -1024 S_lo        (k with low nibbles,  2 MFMAs each)
//         P_hi = x_hi.(1024 + 16 q), A_hi = x_hi.c = -1024 S_hi      (k with high nibbles, 2 MFMAs each)
//         acc += s * ((P_lo + A_lo) + (P_hi + A_hi) / 16) + sz * (-(A_lo + A_hi) / 1024)
//     8 MFMAs + ~8 VALU per step instead of 4 MFMAs + LDS-resident sums; fp32 accumulation as before.
//   * everything a block stages -- x, its (scale | scaled_zero) words, its fp16 outlier rows, the epilogue's residual /
//     gamma values, the producer's partial sums of squares -- reaches LDS by LDS-DMA (global_load_lds, 16 B/lane, no VGPR
//     round trip, no VALU) issued in front of the weight ring; there is NO compiler-visible global load in front of the
//     ring, and every kernel argument is read in one batch of scalar loads behind one wait.
//   * the transforms moved to the PRODUCER's epilogue, where they touch 16-64 values per block instead of K:
//       RMSNorm:  the producer of h also stores fp16(h * gamma) and its block's partial sum of squares; the consumer
//                 multiplies its outputs by rsqrt(sum / K + eps)  (W.(h gamma rs) = rs W.(h gamma), as in round 1);
//       SiLU*up:  PAIR mode: the operand's rows are stored pair-interleaved (8 gate rows, 8 up rows per 16-row MFMA set;
//                 a derived buffer built once at load time, qeft_amd/fuse.py), so the block that finishes a set holds
//                 both halves and stores silu(gate) * up directly.
//   * one operand per launch: q|k|v are concatenated at load time, so the kernel has no per-part logic.
//   * the outlier step takes its B fragments from the PLAIN oweight [N, r] rows (64 contiguous bytes per lane, gathered by
//     the DMA into an XOR-swizzled LDS image) -- or, for the reference's entries, from oweight_interleaved (below).
//
// Round 3:
//   * the block's row-set count is a template parameter (RSC): ring slot, accumulator and scale row of every (step, row set)
//     are fixed in the code (the run-time bookkeeping of round 2's loop was a third of its instructions); 8-wave blocks
//     (4-wave blocks for launches of more than 256 blocks), ring depth 2 / 4 / 6 chosen by the launcher (gemv_v3.hip);
//   * checkpoint-layout operands, selected by run-time flags (V3_F_*), all handled LDS -> LDS in the phase between the staging
//     barrier and the first step, while the first weight loads are in flight:
//       V3_F_SZN     scales / scaled_zeros fp16 [K/g][N] staged raw (32 B per group and row set) and packed into the words above;
//       V3_F_OWIL    the outlier slab from oweight_interleaved [N/2][2r] (qlinear.py:70-79), de-interleaved by v_perm_b32 in the
//                    outlier step;
//       V3_F_GATHER  x[:, reorder_ids] (qlinear.py:275): ids and raw rows by DMA, the gather in LDS;
//     and MB = 2 instantiations for m = 2..7 batch rows (the A rows of the same MFMAs; x rows 16 bytes askew in LDS).
//
// Address arithmetic lives in __host__ __device__ functions shared with a host-side enumerator
// (qeft_gemv_v3_check_extents, capi.hip) that walks every block / wave / lane / piece / step of a configuration and
// checks each access against the operand sizes on the CPU (tests/test_gemv_v3_extents.py): the unconditional, clamped
// loads this design relies on are exactly where the round-1 faults came from.
#pragma once
#include "qeft_common.h"

namespace qeft {

constexpr int V3_NW = 8;        // waves per block (4 for launches of more than 256 blocks); wave w owns the 128-k steps w, w + NW, ...
constexpr int V3_NW_MAX = 16;   // LDS sizing of the per-wave partial sums (the lab still instantiates 16-wave blocks)
constexpr int V3_MAX_RS = 4;    // 16-row sets per block (LDS is carved for rs_cap <= 4)
constexpr int V3_MAX_SSQ = 512; // partial sums of squares a consumer accepts
constexpr int V3_MODE_PLAIN = 0;
constexpr int V3_MODE_PAIR = 1; // rows pair-interleaved: set g = gate rows [8g, 8g+8) then up rows [8g, 8g+8); y[8g + i] = silu(gate) * up
constexpr int V3_MAX_M = 16;    // batch rows of one launch: the A rows of one MFMA (the reference's gemv entries serve m = 1..7, gemv_cuda_qeft.cu:433-466;
                                // 8..16 rows reach this kernel from the GEMM entries, capi.hip gemm_impl)
// run-time flags of a launch, packed beside nblk / rs_cap in one preloaded dword (V3_KERNEL_ARGS)
constexpr uint32_t V3_F_PERCH = 1u << 24;   // one group per row (group size == K)
constexpr uint32_t V3_F_XN = 1u << 25;      // x is fp32 h, xn_gamma its gamma: the whole RMSNorm in this launch
constexpr uint32_t V3_F_SZN = 1u << 26;     // scales / scaled_zeros in their CHECKPOINT layout fp16 [K/g][N] (szp = scales, xn_gamma = zeros)
constexpr uint32_t V3_F_OWIL = 1u << 27;    // outlier slice from oweight_interleaved [N/2][256] (pack_oweight, qlinear.py:70-79)
constexpr uint32_t V3_F_GATHER = 1u << 31;  // x[:, ids] (qlinear.py:275), ids int32 [K] in the tail
constexpr int V3_F_M_SHIFT = 19;            // bits 19..22: m - 1 (rs_cap <= 4 sits in bits 16..18)

struct V3Geom {
    int K, n_out, nsteps, nfull, ngroups, nsets;   // nsets = N / 16
};

struct V3Args {
    const f16* x;           // [K] fp16, consumed as it is (xn_gamma == NULL), or the fp32 vector h [K] (xn_gamma != NULL)
    const f16* xn_gamma;    // optional: x is fp32 h and the launch itself stages fp16(h * xn_gamma) and applies rsqrt(mean h^2 + eps)
                            // to its row sums -- the whole RMSNorm on the consumer (tensor-parallel path: h comes out of an all-reduce)
    const uint8_t* qw;      // int16 [N/4][K] checkpoint layout
    const uint8_t* szp;     // u32 [N/16][ngroups][16] (scale | scaled_zero << 16), qeft_pack_scales
    const uint8_t* ow;      // fp16 [N][128] plain outlier rows (unused when n_out == 0)
    const float* ssq_in;    // optional: n_ssq_in partial sums of squares of the vector x was derived from (x = v * gamma)
    const float* residual;  // optional (PLAIN): the fp32 residual stream; y32 = acc + residual (fp32 out, may alias)
    const f16* gamma_out;   // optional (PLAIN, with residual): ynorm = fp16(y * gamma_out), ssq_out[block] = sum of y^2 over its rows
    V3Geom g;
    int rs_cap, nblk;       // LDS row sets per block, grid size (read from here, not from the dispatch packet)
    int sets_q, sets_r;     // nsets / grid, nsets % grid (host-side division: the kernel deals sets without dividing)
    int n_ssq_in;
    float eps;
    const f16* bias;        // optional [N] (PAIR: in the interleaved row order)
    f16* y;                 // PLAIN without residual: [N]; PAIR: [N/2]
    float* y32;             // PLAIN with residual: [N]
    f16* ynorm;
    float* ssq_out;
    long long* dbg;         // lab hook (V3_STAMP_FLUSH): per block 8 x 100 MHz time stamps of wave 0; NULL and never read in the product
    int bits;               // 4 (0 is taken as 4), or 3: qw is the 3-bit extension layout int32 [N/16][nfull * 192]
    // ---- the reference's entry points (gemv_4bit[_qeft], QuantLinear.forward for < 8 rows): operands as the checkpoint holds them
    const f16* scales;      // with `zeros`: fp16 [K/g][N] each, used when szp == NULL (staged raw, packed in LDS)
    const f16* zeros;
    const uint8_t* ow_il;   // oweight_interleaved fp16 [N/2][256], used when ow == NULL and n_out > 0
    const int* ids;         // optional reorder_ids int32 [K]: the launch consumes x[:, ids]
    int m;                  // batch rows 1..16 (0 is taken as 1); x is [m][K], y [m][N]; m > 1: PLAIN, no residual / ssq_in / xn
    int nw;                 // waves per block chosen by the launcher (LDS sizing)
    bool xg;                // m > 1: the lanes read their x fragments from global memory (no x rows in LDS): any K with 16 rows
};

// What the kernel receives.  The first 16 dwords of the kernel-argument segment -- the four operand pointers and the
// geometry the prologue needs before it can issue its DMA pieces and ring loads -- are plain leading parameters so that
// they can be PRELOADED into SGPRs at wave launch (-mllvm -amdgpu-kernarg-preload-count=16: no scalar-memory round trip
// between a wave's first instruction and its first load); everything only the later phases read travels in V3Tail.
struct V3Tail {
    const float* ssq_in;
    const float* residual;
    const f16* gamma_out;
    const f16* bias;
    f16* y;
    float* y32;
    f16* ynorm;
    float* ssq_out;
    long long* dbg;
    const int* ids;
    int n_out, nsteps, nsets, n_ssq_in;
    float eps;
};
inline V3Tail v3_tail(const V3Args& a) {
    return V3Tail{a.ssq_in, a.residual, a.gamma_out, a.bias, a.y, a.y32, a.ynorm, a.ssq_out, a.dbg, a.ids, a.g.n_out, a.g.nsteps, a.g.nsets, a.n_ssq_in, a.eps};
}
inline uint32_t v3_flags(const V3Args& a) {
    const int m = a.m > 0 ? a.m : 1;
    return ((a).g.ngroups == 1 && (a).g.K > 128 ? V3_F_PERCH : 0u) | ((a).xn_gamma ? V3_F_XN : 0u) | (!a.szp ? V3_F_SZN : 0u) |
           (!a.ow && a.g.n_out > 0 ? V3_F_OWIL : 0u) | (a.ids ? V3_F_GATHER : 0u) | ((uint32_t)(m - 1) << V3_F_M_SHIFT);
}
// 13 dwords (14 user SGPRs are available for preloading next to the kernarg pointer): five pointers, K, and two packed words:
// nblk | rs_cap << 16 | per-channel << 24 | xn << 25 and sets_q | sets_r << 16 (nblk, sets_r < 65536: gemv_v3_launch checks);
// the step and group counts follow from K and the flags
// (checkpoint-layout launches: the scales pointer travels in the szp slot, scaled_zeros in the xn_gamma slot, oweight_interleaved in ow's)
#define V3_KERNEL_ARGS(a) (a).qw, (a).x, ((a).szp ? (a).szp : (const uint8_t*)(a).scales), ((a).ow ? (a).ow : (a).ow_il), \
    ((a).szp ? (a).xn_gamma : (a).zeros), (a).g.K, (uint32_t)(a).nblk | ((uint32_t)(a).rs_cap << 16) | qeft::v3_flags(a), \
    (uint32_t)(a).sets_q | ((uint32_t)(a).sets_r << 16), qeft::v3_tail(a)

// ---- LDS carve-up (bytes); every DMA-filled region is a whole number of 1 KB pieces
__host__ __device__ constexpr int v3_x_bytes(int K) { return (K * 2 + 1023) / 1024 * 1024; }
__host__ __device__ constexpr int v3_sz_bytes(int ngroups) { return (ngroups * 64 + 1023) / 1024 * 1024; }   // per row set
__host__ __device__ constexpr int v3_xf_bytes(int K) { return (K * 4 + 1023) / 1024 * 1024; }   // fp32 h of an xn launch
__host__ __device__ constexpr int v3_szraw_bytes(int ngroups) { return (ngroups * 32 + 1023) / 1024 * 1024; }   // per row set and array (checkpoint-layout scales)
// batch rows of x sit XS bytes apart: whole pieces, plus 16 bytes when there are several rows so that the A-fragment reads of
// different batch rows (same k) fall into different banks
__host__ __device__ constexpr int v3_x_stride(int K, int m) { return v3_x_bytes(K) + (m > 1 ? 16 : 0); }
// per-wave partial sums: m == 1: [rs_cap][NW_MAX][16] floats; m > 1: [rs_cap][nw][8 or 16 batch rows][16]
__host__ __device__ constexpr size_t v3_red_bytes(int rs_cap, int m = 1, int nw = V3_NW_MAX) {
    return m > 1 ? ((size_t)rs_cap * nw * (m > 8 ? 16 : 8) * 16 * 4 + 1023) / 1024 * 1024 : ((size_t)rs_cap * V3_NW_MAX * 16 * 4 + 64 + 1023) / 1024 * 1024;
}
struct V3Lds {              // byte offsets of the regions inside the block's dynamic LDS
    uint32_t xs, szl, owl, epl, ssql, red, xf, xg, szraw, idsl, xraw, total;
};
__host__ __device__ inline V3Lds v3_lds(int K, int ngroups, int n_out, int rs_cap, int m, int nw, bool xn, bool szn, bool gather, bool xg = false) {
    V3Lds L;
    uint32_t o = 0;
    L.xs = o;   o += xg ? 0u : ((uint32_t)m * v3_x_stride(K, m) + 1023u) / 1024u * 1024u;   // [m][K] fp16 as the MFMAs read it (xg: x stays in global memory)
    L.szl = o;  o += (uint32_t)rs_cap * v3_sz_bytes(ngroups);                               // [rs_cap][groups][16] u32
    L.owl = o;  o += n_out > 0 ? (uint32_t)rs_cap * 4096u : 0u;                             // [rs_cap] 4 KB outlier slab (swizzled)
    L.epl = o;  o += 1024;                                                                  // epilogue operands
    L.ssql = o; o += 2048;                                                                  // ssq_in
    L.red = o;  o += (uint32_t)v3_red_bytes(rs_cap, m, nw);
    L.xf = o;   o += xn ? (uint32_t)v3_xf_bytes(K) : 0u;                                    // xn: fp32 h
    L.xg = o;   o += xn ? (uint32_t)v3_x_bytes(K) : 0u;                                     //     its gamma
    L.szraw = o; o += szn ? 2u * rs_cap * v3_szraw_bytes(ngroups) : 0u;                     // [2 arrays][rs_cap][groups][16] fp16
    L.idsl = o; o += gather ? (uint32_t)v3_xf_bytes(K) : 0u;                                // reorder_ids int32 [K]
    L.xraw = o; o += gather ? (uint32_t)m * v3_x_bytes(K) : 0u;                             // x rows before the gather
    L.total = o;
    return L;
}
__host__ __device__ inline size_t v3_smem_bytes(int K, int ngroups, int n_out, int rs_cap, bool xn = false, int m = 1, int nw = V3_NW_MAX,
                                                bool szn = false, bool gather = false) {
    return v3_lds(K, ngroups, n_out, rs_cap, m, nw, xn, szn, gather).total;
}

// ---- source byte offsets of every load (relative to the operand's base)
// weights of (set g, lane row nl, 32-k chunk kc): row-group base, and the lane's offset inside it (+ 256 per step)
__host__ __device__ inline size_t v3_w_set_off(const V3Geom& G, int g) { return (size_t)g * 4 * G.K * 2; }
__host__ __device__ inline uint32_t v3_w_lane_off(const V3Geom& G, int nl, int kc) {
    return (uint32_t)(nl >> 2) * (uint32_t)G.K * 2u + (uint32_t)(kc >> 1) * 128u + (uint32_t)(nl & 3) * 32u + (uint32_t)(kc & 1) * 16u;
}
__host__ __device__ inline uint32_t v3_last_step_off(const V3Geom& G) { return (uint32_t)G.K * 2u - 256u; }
// 3-bit extension layout (oracle.pack_w3): int32 [N/16][nfull][64 lanes][3] -- a 16-row x 128-k step is 768 contiguous bytes,
// lane L = (row L & 15, 32-k chunk L >> 4) holds 12 of them; a row set is nfull consecutive steps
__host__ __device__ inline size_t v3w3_set_off(const V3Geom& G, int g) { return (size_t)g * G.nfull * 768; }
__host__ __device__ inline uint32_t v3w3_last_step_off(const V3Geom& G) { return (uint32_t)(G.nfull > 0 ? G.nfull - 1 : 0) * 768u; }
// x piece i: the lane's 16 source bytes, clamped (the LDS destination is lane-linear; clamped lanes fill padding)
__host__ __device__ inline uint32_t v3_x_off(const V3Geom& G, int piece, int lane) {
    const uint32_t o = (uint32_t)piece * 1024u + (uint32_t)lane * 16u, last = (uint32_t)G.K * 2u - 16u;
    return o < last ? o : last;
}
// xn launches: piece i of the fp32 vector h (K * 4 bytes); its gamma uses v3_x_off
__host__ __device__ inline uint32_t v3_xf_off(const V3Geom& G, int piece, int lane) {
    const uint32_t o = (uint32_t)piece * 1024u + (uint32_t)lane * 16u, last = (uint32_t)G.K * 4u - 16u;
    return o < last ? o : last;
}
// scale piece j of set g: the set's words are one contiguous run of ngroups * 64 bytes
__host__ __device__ inline size_t v3_sz_off(const V3Geom& G, int g, int j, int lane) {
    const uint32_t o = (uint32_t)j * 1024u + (uint32_t)lane * 16u, last = (uint32_t)G.ngroups * 64u - 16u;
    return (size_t)g * G.ngroups * 64 + (o < last ? o : last);
}
// outlier piece j of set g: lane = (row 4j + lane/16, LDS chunk lane%16 <- global chunk (lane%16) ^ row)
__host__ __device__ inline size_t v3_ow_off(int g, int j, int lane) {
    const int row = 4 * j + (lane >> 4), cc = (lane & 15) ^ row;
    return ((size_t)(g * 16 + row) * 128 + cc * 8) * 2;
}
// outlier piece j of set g from oweight_interleaved [N/2][256] fp16: the set's 8 interleaved rows (512 B each), two per piece.
// LDS position p (16-byte chunk 0..31 of interleaved row R) <- source chunk ((p >> 1) ^ R) << 1 | (p & 1): the 32-byte pairs a
// lane reads per MFMA are XOR-swizzled by the row so that the 8 rows of a read fall into different banks
__host__ __device__ inline size_t v3_owil_off(int g, int j, int lane) {
    const int R = 2 * j + (lane >> 5), p = lane & 31, c = (((p >> 1) ^ R) << 1) | (p & 1);
    return ((size_t)(g * 8 + R) * 256 + (size_t)c * 8) * 2;
}
// checkpoint-layout scales / scaled_zeros fp16 [ngroups][N]: piece j of set g = groups 32 j .. 32 j + 31, two lanes (8 rows each) per group
__host__ __device__ inline size_t v3_szn_off(const V3Geom& G, int g, int j, int lane) {
    const int grp = 32 * j + (lane >> 1), gc = grp < G.ngroups ? grp : G.ngroups - 1;
    return ((size_t)gc * G.nsets * 16 + (size_t)g * 16 + (size_t)(lane & 1) * 8) * 2;
}
// epilogue operands, one piece: lanes [0, 4 RS) = residual (4 floats of the block's rows each), lanes [16, 16 + 2 RS) =
// gamma_out (8 halves each); every other lane re-reads lane 0's / lane 16's vector.  Byte offset into the respective vector.
__host__ __device__ inline size_t v3_epi_off(int set0, int RS, int lane) {
    if (lane < 16) return ((size_t)set0 * 16 + (size_t)(lane < 4 * RS ? lane : 0) * 4) * 4;         // residual (fp32)
    const int l = lane - 16;
    return ((size_t)set0 * 16 + (size_t)(l < 2 * RS ? l : 0) * 8) * 2;                               // gamma_out (fp16)
}

// Row sets [0, nsets) dealt to the blocks as evenly as possible: the first r blocks get q + 1 (q = nsets / nblk, r = nsets % nblk)
__host__ __device__ inline void v3_block_sets(int b, int q, int r, int& set0, int& cnt) {
    set0 = b * q + (b < r ? b : r);
    cnt = q + (b < r ? 1 : 0);
}
// Blocks are dealt round-robin over the 8 XCDs (b and b + 8 share an L2); neighbouring sets share scale / outlier lines.
__host__ __device__ inline int v3_xcd_block(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

#if defined(__HIPCC__)
typedef _Float16 v3h8 __attribute__((ext_vector_type(8)));

// LDS-DMA of one 1 KB piece: lane's 16 bytes at gsrc -> LDS[lds_dst + 16 * lane].  Hidden from hipcc's s_waitcnt
// bookkeeping on purpose (guide 5.7): the pieces are OLDER than every ring load, so the compiler's counted waits for the
// ring cover them, and the one wait that matters (before the staging barrier) is placed by hand.
__device__ __forceinline__ void v3_dma16(const void* gsrc, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// Lab hooks (tools/gemv_v3_lab.hip defines them before including this file; the product compiles them away): in-kernel time
// stamps of wave 0 of every block, written through V3Args::dbg
#if !defined(V3_STAMP)
#define V3_STAMP_DECL
#define V3_STAMP(i) do { } while (0)
#define V3_STAMP_VALUE(v) do { } while (0)
#define V3_STAMP_FLUSH() do { } while (0)
#endif
#define V3_LAUNCH_THREADS(nw) ((nw) * 64)

// The value a lane accumulates per row set: batch row 0 only (MB == 1: D row 0 = register 0 of the lanes kc == 0), or the four
// D rows 4 kc .. 4 kc + 3 of the lane (MB == 2: batch rows 0..7 live in the lanes kc < 2, rows 8..15 in the others)
template <int MB> struct V3Val { typedef float type; };
template <> struct V3Val<2> { typedef f32x4 type; };
template <int MB> __device__ __forceinline__ typename V3Val<MB>::type v3_pick(const f32x4& v) {
    if constexpr (MB == 1) return v[0]; else return v;
}
template <int MB> __device__ __forceinline__ typename V3Val<MB>::type v3_fma_s(float a, typename V3Val<MB>::type b, typename V3Val<MB>::type c) {
    if constexpr (MB == 1) return __builtin_fmaf(a, b, c); else return __builtin_elementwise_fma(f32x4{a, a, a, a}, b, c);
}
template <int MB> __device__ __forceinline__ typename V3Val<MB>::type v3_zero() {
    if constexpr (MB == 1) return 0.f; else return f32x4{0.f, 0.f, 0.f, 0.f};
}

// compile-time loop over 0 .. N-1 (indices as integral constants: ring slots, row sets and accumulators stay in registers)
template <int I, int N, typename F> __device__ __forceinline__ void v3_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        v3_static_for<I + 1, N>(f);
    }
}
__host__ __device__ constexpr int v3_gcd(int a, int b) { return b == 0 ? a : v3_gcd(b, a % b); }
// steps per unrolled round: even (the x fragments alternate between two register sets) and a whole number of ring turns
__host__ __device__ constexpr int v3_unroll_steps(int D, int RSC) {
    const int t = D / v3_gcd(D, RSC);
    return t % 2 == 0 ? t : 2 * t;
}

// NW waves per block (8, or 16 for launches of one block per CU: twice the instruction streams per SIMD for the same bytes);
// wave w owns the 128-k steps w, w + NW, ...
// MB: 1 = one batch row (the decode engine, and m = 1 at the reference's entry points); 2 = up to V3_MAX_M batch rows.
// RSC: 16-row sets per block, a COMPILE-TIME constant (round 3): every (step, row set) of the wave's sequence then has its ring
//      slot, accumulator and scale row fixed in the code -- the run-time bookkeeping of round 2's loop (which set is next, 0 / 1
//      factors in front of four accumulators, three branches per consume) was a third of its ~95 instructions per 1 KB of
//      weights.  A launch's blocks hold RSC sets, except at most RSC - 1 blocks that hold one fewer (gemv_v3_blocks): those
//      stream their last set twice and drop the copy's results.
// FL: the run-time flags (V3_F_*: per-channel scales, consumer-side norm, checkpoint-layout operands, gather) are honoured; false
//     for the plain launches of the decode engine, which then carry no trace of those paths.
template <int NW, int D, bool OUTL, int MODE, int BITS = 4, int MB = 1, int RSC = 1, bool FL = true, bool XG = false>
__global__ __launch_bounds__(V3_LAUNCH_THREADS(NW)) void gemv_v3_kernel(const uint8_t* qw, const f16* x_in, const uint8_t* szp, const uint8_t* ow,
                                                          const f16* xn_gamma, int K_, uint32_t nblk_rscap_flags, uint32_t setsq_setsr,
                                                          V3Tail a) {
    static_assert(BITS == 4 || BITS == 3, "4-bit checkpoint layout or the 3-bit extension layout (oracle: pack_w3)");
    static_assert(MB == 1 || (MODE == V3_MODE_PLAIN && BITS == 4), "several batch rows: plain 4-bit launches only");
    static_assert(RSC >= 1 && RSC <= V3_MAX_RS && D >= 2, "row sets per block 1..4, at least two loads in flight per wave");
    static_assert(FL || MB == 1, "the batch-row launches always come with flags");
    static_assert(!XG || (MB == 2 && D >= 4), "x fragments from global memory: batch-row launches with a deep ring");
    typedef typename V3Val<MB>::type val_t;
    // the leading parameters arrive in SGPRs (kernarg preload); the tail is one batch of scalar loads issued here and waited
    // for once, behind the ring issue (the pin below) -- argument loads that hipcc leaves next to their first use each cost a
    // dependent scalar-memory round trip in the middle of the stream
    // (FL == false: the launcher guarantees that no flag is set -- the decoding and the never-taken transform branches cost a
    //  plain launch ~0.12 us of prologue, profiles/r03_gemv_lab.txt)
    const bool per_channel_f = FL && (nblk_rscap_flags & V3_F_PERCH) != 0, XN = FL && MB == 1 && (nblk_rscap_flags & V3_F_XN) != 0;
    const bool SZN = FL && (nblk_rscap_flags & V3_F_SZN) != 0, OWIL = FL && OUTL && (nblk_rscap_flags & V3_F_OWIL) != 0;
    const bool GATHER = FL && (nblk_rscap_flags & V3_F_GATHER) != 0;
    const int m = MB == 1 ? 1 : (int)((nblk_rscap_flags >> V3_F_M_SHIFT) & 15u) + 1;
    const int nblk = (int)(nblk_rscap_flags & 0xffffu);
    const int sets_q = (int)(setsq_setsr & 0xffffu), sets_r = (int)(setsq_setsr >> 16);
    // (nsets from the preloaded words: the checkpoint-layout scale pieces need N before the tail has arrived)
    V3Geom G{K_, a.n_out, a.nsteps, (K_ >> 7) - (OUTL ? 1 : 0), per_channel_f ? 1 : (K_ >> 7), sets_q * nblk + sets_r};
    const int ssq_n = a.n_ssq_in;
    const float eps = a.eps;
    const f16* const bias = a.bias;
    f16* const yout = a.y;
    float* const y32 = a.y32;
    f16* const ynorm = a.ynorm;
    float* const ssq_out = a.ssq_out;
    const uint8_t* const xptr = (const uint8_t*)x_in;
    const uint8_t* const ssq_in = (const uint8_t*)a.ssq_in;
    const uint8_t* const residual = (const uint8_t*)a.residual;
    const uint8_t* const gamma_out = (const uint8_t*)a.gamma_out;
    const uint8_t* const idsp = (const uint8_t*)a.ids;
    asm volatile("" ::"s"(G.K), "s"(G.nfull), "s"(G.ngroups), "s"(nblk), "s"(sets_q), "s"(sets_r), "s"(xptr), "s"(qw),
                 "s"(szp), "s"(ow), "s"(xn_gamma));

    extern __shared__ __attribute__((aligned(1024))) uint8_t smem[];
    const V3Lds L = v3_lds(G.K, G.ngroups, OUTL ? 128 : 0, RSC, m, NW, XN, SZN, GATHER, XG);
    const int XB = v3_x_bytes(G.K), SZB = v3_sz_bytes(G.ngroups), XS = v3_x_stride(G.K, m);
    uint8_t* const xs = smem + L.xs;                                  // [m][K] fp16 raw (rows XS apart)
    uint8_t* const szl = smem + L.szl;                                // [RSC][SZB / 64][16] u32
    uint8_t* const owl = smem + L.owl;                                // [RSC] 4 KB: 16 plain rows (chunks ^ row) or 8 interleaved rows
    uint8_t* const epl = smem + L.epl;                                // [64 lanes][16 B]: residual | gamma_out of the block's rows
    float* const ssql = (float*)(smem + L.ssql);                      // [512] ssq_in
    float* const red = (float*)(smem + L.red);                        // [RSC][NW][16], or [RSC][NW][8 or 16][16] (MB == 2)
    uint8_t* const xf32 = smem + L.xf;                                // xn launches: [K] fp32 h, then its gamma [K] fp16
    uint8_t* const xg = smem + L.xg;
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nl = lane & 15, kc = lane >> 4;
    int set0, RS;                                                     // RS = RSC, or RSC - 1 in a short block
    v3_block_sets(v3_xcd_block(blockIdx.x, nblk), sets_q, sets_r, set0, RS);
    auto set_of = [&](int rs) { return set0 + (rs < RS ? rs : RS - 1); };      // a short block's last slot repeats its last set
    V3_STAMP_DECL
    V3_STAMP(0);

    // ---- 1. staging by LDS-DMA.  x: the waves take pieces w, w + NW, ..; per row set the scale words (piece j = wave) and
    //         the outlier rows (the last 4 waves, one piece each).  No VGPR destination, no VALU on the data, nothing to
    //         wait for until the barrier below.  (What only the epilogue reads is requested behind the ring, step 2b.)
    const int PX = XB >> 10, SPS = SZB >> 10;
    const int SRB = v3_szraw_bytes(G.ngroups), SPR = SRB >> 10;
    auto stage_all = [&]() {
        if (XG) {
            // x fragments come straight from global memory (step 4)
        } else if (!XN) {
            // (a gathering launch stages the rows as they are into xraw; step 3b writes xs)
            const uint32_t xdst = lds0 + (GATHER ? L.xraw : L.xs), xstr = GATHER ? (uint32_t)XB : (uint32_t)XS;
            for (int i = 0; i < m; ++i)
                for (int p = wave; p < PX; p += NW)
                    v3_dma16(xptr + (size_t)i * G.K * 2 + v3_x_off(G, p, lane), __builtin_amdgcn_readfirstlane(xdst + (uint32_t)i * xstr + ((uint32_t)p << 10)));
        } else {               // fp32 h (2 PX pieces) and its gamma (PX pieces) go to their own regions; xs is written in step 3b
            const int PF = v3_xf_bytes(G.K) >> 10;
            for (int p = wave; p < PF + PX; p += NW) {
                if (p < PF)
                    v3_dma16(xptr + v3_xf_off(G, p, lane), __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)xf32 + ((uint32_t)p << 10)));
                else
                    v3_dma16((const uint8_t*)xn_gamma + v3_x_off(G, p - PF, lane),
                             __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)xg + ((uint32_t)(p - PF) << 10)));
            }
        }
        {
    #pragma unroll
            for (int rs = 0; rs < RSC; ++rs) {
                const int set = set_of(rs);
                if (!SZN) {
                    for (int j = wave; j < SPS; j += NW)
                        v3_dma16(szp + v3_sz_off(G, set, j, lane), __builtin_amdgcn_readfirstlane(lds0 + L.szl + (uint32_t)rs * SZB + ((uint32_t)j << 10)));
                } else {        // checkpoint layout: scales (array 0, the szp slot) and scaled_zeros (array 1, the xn_gamma slot), packed in step 3b
                    for (int t = wave; t < 2 * SPR; t += NW) {
                        const int arr = t >= SPR ? 1 : 0, j = t - arr * SPR;
                        v3_dma16((arr ? (const uint8_t*)xn_gamma : szp) + v3_szn_off(G, set, j, lane),
                                 __builtin_amdgcn_readfirstlane(lds0 + L.szraw + (uint32_t)(arr * RSC + rs) * SRB + ((uint32_t)j << 10)));
                    }
                }
                if (OUTL && wave >= NW - 4)
                    v3_dma16(ow + (OWIL ? v3_owil_off(set, wave - (NW - 4), lane) : v3_ow_off(set, wave - (NW - 4), lane)),
                             __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)owl + (uint32_t)rs * 4096u + ((uint32_t)(wave - (NW - 4)) << 10)));
            }
        }
    };
    stage_all();

    // ---- 2. weight stream: ring of D loads per wave, branch-free, oldest first.  The wave's work is the sequence
    //         c = 0 .. nsw * RSC - 1 of (step wave + NW (c / RSC), row set c % RSC): STEP-major, so the x fragments and the bias
    //         sums of a step are fetched / computed once and serve all RSC row sets; load c goes to ring slot c % D.  The D
    //         issues past the end re-read the wave's last step (a valid address).
    const int nsw = (G.nfull - wave + NW - 1) / NW;
    constexpr uint32_t STEPB = BITS == 3 ? 768u : 256u;              // bytes from one 128-k step of a row (group) to the next
    const uint32_t set_bytes = BITS == 3 ? (uint32_t)G.nfull * 768u : (uint32_t)G.K * 8u;   // one 16-row set
    const uint8_t* const wbase = qw + (BITS == 3 ? v3w3_set_off(G, set0) : v3_w_set_off(G, set0));         // wave-uniform
    const uint32_t lane_off = BITS == 3 ? (uint32_t)lane * 12u : v3_w_lane_off(G, nl, kc);
    const uint32_t step0 = min((uint32_t)wave * STEPB, BITS == 3 ? v3w3_last_step_off(G) : v3_last_step_off(G));
    const uint32_t step_last = step0 + (uint32_t)(nsw > 0 ? nsw - 1 : 0) * (NW * STEPB);
    const uint32_t short_back = RS < RSC ? set_bytes : 0u;           // a short block's last slot: back to its last real set
    typedef typename std::conditional<BITS == 3, u32x3_u, u32x4>::type ring_t;
    ring_t ring[D];
    uint32_t p_step = step0;                                         // uniform byte offset of the step the next issue reads
    // issue of load number c (its row set is a compile-time constant; the step advances behind a step's last row set)
    auto issue = [&](ring_t& b, auto rs_tag) {
        constexpr int rs = decltype(rs_tag)::value;
        const uint32_t off = p_step + (uint32_t)rs * set_bytes - (rs == RSC - 1 ? short_back : 0u);
        b = __builtin_nontemporal_load((const ring_t*)(wbase + off + lane_off));
        if (rs == RSC - 1) p_step = min(p_step + NW * STEPB, step_last);
    };
    v3_static_for<0, D>([&](auto d) {
        issue(ring[d], std::integral_constant<int, decltype(d)::value % RSC>{});
        __builtin_amdgcn_sched_barrier(0);
    });

    asm volatile("" ::"s"(G.nsteps), "s"(ssq_n), "s"(ssq_in), "s"(residual), "s"(gamma_out), "s"(eps), "s"(bias), "s"(yout), "s"(y32),
                 "s"(ynorm), "s"(ssq_out), "s"(idsp));
    V3_STAMP(1);
    // ---- 1b. a gathering launch: reorder_ids (tail operand, so behind the ring) -> LDS, K * 4 bytes
    if (GATHER) {
        const int PI = v3_xf_bytes(G.K) >> 10;
        for (int p = wave; p < PI; p += NW)
            v3_dma16(idsp + v3_xf_off(G, p, lane), __builtin_amdgcn_readfirstlane(lds0 + L.idsl + ((uint32_t)p << 10)));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // ---- 3. staged data complete: this wave's pieces are older than its D ring loads
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D) : "memory");
    // ---- 2b. epilogue-only operands, requested BEHIND the ring (they must not delay the weight stream): wave 1 the
    //          residual / gamma_out values of the block's rows, waves 2 and 3 the producer's partial sums of squares.  They
    //          complete before the vmcnt(0) every wave executes in front of the final barrier.
    if (MB == 1 && MODE == V3_MODE_PLAIN && residual && wave == 1) {
        const bool g_lane = lane >= 16 && gamma_out != nullptr;
        v3_dma16((g_lane ? gamma_out : residual) + v3_epi_off(set0, RS, g_lane ? lane : (lane < 16 ? lane : 0)),
                 __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)epl));
    }
    if (MB == 1 && ssq_in && (wave == 2 || wave == 3) && (wave - 2) * 256 < ssq_n) {       // <= 2 pieces of 256 floats
        const int v = min((wave - 2) * 64 + lane, (ssq_n - 1) >> 2);             // clamped 16-byte vector of the array
        v3_dma16(ssq_in + (size_t)v * 16, __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)ssql + (uint32_t)(wave - 2) * 1024u));
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // ---- 3b. LDS -> LDS transforms of the staged data, while the first weight loads are still on their way:
    //          xn:      xs = fp16(h * gamma) (the producers' rounding, qeft_residual_norm), sum of h^2 per wave -> LDS
    //          gather:  xs[i][k] = xraw[i][ids[k]]  (QuantLinear.forward_outlier_out_proj's index_select, qlinear.py:275)
    //          szn:     szl[rs][group][row] = scale | scaled_zero << 16 from the two checkpoint-layout images
    float xn_ss = 0.f;
    if (XN) {
        for (int e = tid * 4; e < G.K; e += NW * 64 * 4) {
            const f32x4 hv = *(const f32x4*)(xf32 + (size_t)e * 4);
            const h4 gv = *(const h4*)(xg + (size_t)e * 2);
            h4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j] = mul_f32_to_f16(hv[j], (float)gv[j]);
                xn_ss += hv[j] * hv[j];
            }
            *(h4*)(xs + (size_t)e * 2) = o;
        }
        xn_ss = wave_sum(xn_ss);
        if (lane == 0) ssql[wave] = xn_ss;             // the ssq_in region is free in an xn launch
    }
    if (GATHER) {
        const uint8_t* const idsl = smem + L.idsl;
        const uint8_t* const xraw = smem + L.xraw;
        for (int e = tid * 4; e < G.K; e += NW * 64 * 4) {
            const u32x4 id4 = *(const u32x4*)(idsl + (size_t)e * 4);
            for (int i = 0; i < m; ++i) {
                const uint8_t* row = xraw + (size_t)i * XB;
                uint32_t v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = *(const uint16_t*)(row + (size_t)min(id4[j], (uint32_t)G.K - 1u) * 2);
                *(u32x2*)(xs + (size_t)i * XS + (size_t)e * 2) = u32x2{v[0] | (v[1] << 16), v[2] | (v[3] << 16)};
            }
        }
    }
    if (SZN) {
        const uint8_t* const sraw = smem + L.szraw;
        const uint8_t* const zraw = sraw + (size_t)RSC * SRB;
        for (int rs = 0; rs < RSC; ++rs)
            for (int q = tid; q < G.ngroups * 4; q += NW * 64) {     // q = (group, 4 rows)
                const u32x2 sv = *(const u32x2*)(sraw + (size_t)rs * SRB + (size_t)q * 8);
                const u32x2 zv = *(const u32x2*)(zraw + (size_t)rs * SRB + (size_t)q * 8);
                *(u32x4*)(szl + (size_t)rs * SZB + (size_t)q * 16) =
                    u32x4{(sv[0] & 0xffffu) | (zv[0] << 16), (sv[0] >> 16) | (zv[0] & 0xffff0000u),
                          (sv[1] & 0xffffu) | (zv[1] << 16), (sv[1] >> 16) | (zv[1] & 0xffff0000u)};
            }
    }
    if (XN || GATHER || SZN) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // LDS only: __syncthreads() would drain the ring
    V3_STAMP(2);

    // ---- 4. steps.  acc[rs]: lanes kc == 0 hold row nl of row set rs (batch row 0 = D row 0, register 0); MB == 2: every lane
    //         holds the batch rows 4 kc .. 4 kc + 3 (D rows) of row nl
    val_t acc[RSC];
    v3_static_for<0, RSC>([&](auto r) { acc[r] = v3_zero<MB>(); });
    uint32_t MAGIC = 0x64006400u, NEG1024 = 0xE400E400u;
    asm volatile("" : "+v"(MAGIC), "+v"(NEG1024));
    const v3h8 c8 = __builtin_bit_cast(v3h8, u32x4{NEG1024, NEG1024, NEG1024, NEG1024});
    // this lane's 32-k chunk of a step: four 16-byte slots; A row = lane & 15 = batch row (rows >= m re-read row m - 1)
    // (XG: the same 64 bytes per lane and step from the x rows in global memory -- L2 hits, every block reads all of x either way)
    const uint8_t* xa = XG ? xptr + (size_t)min(nl, m - 1) * G.K * 2 + kc * 64 : xs + kc * 64 + (MB == 1 ? 0 : (size_t)min(nl, m - 1) * XS);
    const bool per_channel = G.ngroups == 1;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};

    // fp16 outlier columns [K - 128, K): one MFMA step per row set, B fragments from the swizzled LDS rows; the wave whose
    // turn step `nfull` would be takes it, before its ring steps
    if (OUTL && wave == G.nfull % NW) {
        const v3h8* px = (const v3h8*)(xa + (size_t)G.nfull * 256);
        v3h8 xo[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) xo[jj] = px[jj];
        // interleaved slab: lane (row nl, chunk kc) finds its row in interleaved row R = (nl / 8) * 4 + nl % 4, as the even
        // (nl % 8 < 4) or odd halves of every dword; MFMA jj needs the 32 bytes of pair q = 4 kc + jj, stored at pair q ^ R
        const int R = ((nl >> 3) << 2) | (nl & 3);
        const uint32_t sel = (nl & 4) ? 0x07060302u : 0x05040100u;
        v3_static_for<0, RSC>([&](auto rs_tag) {
            constexpr int rs = decltype(rs_tag)::value;
            f32x4 P = z4;
            if (!OWIL) {
                const uint8_t* prow = owl + rs * 4096 + nl * 256;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                    P = __builtin_amdgcn_mfma_f32_16x16x32_f16(xo[jj], *(const v3h8*)(prow + (((kc * 4 + jj) ^ nl) & 15) * 16), P, 0, 0, 0);
            } else {
                const uint8_t* prow = owl + rs * 4096 + R * 512;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const uint8_t* pp = prow + (((kc * 4 + jj) ^ R) & 15) * 32;
                    // (the two halves in the order kc & 1, !(kc & 1): the lanes of a 16-byte read group then cover all banks)
                    const u32x4 h0 = *(const u32x4*)(pp + (kc & 1) * 16), h1 = *(const u32x4*)(pp + ((kc & 1) ^ 1) * 16);
                    const u32x4 lo4 = (kc & 1) ? h1 : h0, hi4 = (kc & 1) ? h0 : h1;
                    const u32x4 bw = {__builtin_amdgcn_perm(lo4[1], lo4[0], sel), __builtin_amdgcn_perm(lo4[3], lo4[2], sel),
                                      __builtin_amdgcn_perm(hi4[1], hi4[0], sel), __builtin_amdgcn_perm(hi4[3], hi4[2], sel)};
                    P = __builtin_amdgcn_mfma_f32_16x16x32_f16(xo[jj], __builtin_bit_cast(v3h8, bw), P, 0, 0, 0);
                }
            }
            acc[rs] = acc[rs] + v3_pick<MB>(P);
        });
    }

    if (nsw > 0) {
        // x fragments of consecutive steps (the roles rotate step by step): two sets from LDS; XG: four, loaded from global memory
        // three steps ahead (an L2 round trip under load is ~1 us, a step a few hundred ns)
        constexpr int XP = XG ? 4 : 2, XD = XP - 1;
        v3h8 xr[XP][4];
        // -1024 S_lo, -1024 S_hi of the current step's x fragments (S = the sum of x over the low- / high-nibble positions), kept
        // as the whole MFMA result: they are the C operand of the step's product MFMAs, so P comes out as sum x q with the
        // 1024 bias already gone and the fold never touches them; cB = -sum(x) of the step for the zero-point term
        f32x4 clo = z4, chi = z4;
        f32x4 nlo = z4, nhi = z4;                               // the next step's, computed one step ahead
        val_t cB = v3_zero<MB>();
        // LDS addresses of the wave's current step: x fragments (256 B per step) and the set-0 scale words (64 B per step / group)
        const uint8_t* xp = xa + (size_t)wave * 256;
        const uint8_t* sp = szl + (size_t)(per_channel ? 0 : wave) * 64 + nl * 4;
        const uint32_t s_stride = per_channel ? 0u : NW * 64u;
        auto load_x = [&](v3h8 (&o)[4], const uint8_t* p) {
            const v3h8* px = (const v3h8*)p;
#pragma unroll
            for (int w = 0; w < 4; ++w) o[w] = px[w];
        };
        auto bias_sums = [&](const v3h8 (&x4)[4], f32x4& lo, f32x4& hi) {
            lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(x4[0], c8, z4, 0, 0, 0);
            hi = __builtin_amdgcn_mfma_f32_16x16x32_f16(x4[1], c8, z4, 0, 0, 0);
            lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(x4[2], c8, lo, 0, 0, 0);
            hi = __builtin_amdgcn_mfma_f32_16x16x32_f16(x4[3], c8, hi, 0, 0, 0);
        };
        auto zero_term = [&](const f32x4& lo, const f32x4& hi) { return v3_pick<MB>(lo) + v3_pick<MB>(hi); };      // -1024 sum(x)
        // XG: address of the fragments of the step XD ahead, clamped to the wave's last step
        const uint8_t* const xlast = xp + (size_t)(nsw - 1) * (NW * 256);
        const uint8_t* xq = xp;
        load_x(xr[0], xp);
        if constexpr (XG) {
            v3_static_for<1, XD>([&](auto d) {
                xq = xq + NW * 256 <= xlast ? xq + NW * 256 : xlast;
                load_x(xr[decltype(d)::value], xq);
            });
            xq = xq + NW * 256 <= xlast ? xq + NW * 256 : xlast;
        }
        uint32_t szw = *(const uint32_t*)sp;                   // scale word of the next consume
        bias_sums(xr[0], clo, chi);
        cB = zero_term(clo, chi);
        // Software pipeline: the products of a (step, row set) are folded into the accumulators one consume LATER, behind
        // the MFMAs of the next one -- nothing reads an MFMA result right behind the MFMA (that wait was ~10 % of a launch).
        val_t pv_lo = v3_zero<MB>(), pv_hi = v3_zero<MB>(), pv_B = v3_zero<MB>();
        uint32_t pv_szw = 0;
        auto fold = [&](val_t& dst) {           // three FMAs per value (MB == 2: six packed ones for the lane's four batch rows)
            const h2 sz2 = as_h2(pv_szw);
            dst = v3_fma_s<MB>((float)sz2[0], v3_fma_s<MB>(0.0625f, pv_hi, pv_lo), dst);
            dst = v3_fma_s<MB>((float)sz2[1] * -0.0009765625f, pv_B, dst);
        };
        // one (step, row set): xc = the step's fragments, xnx = the next step's (fetched at the step's first row set, their bias
        // sums formed behind its last); prev_rs = the row set of the previous consume (whose products are folded here)
        auto consume = [&](ring_t& slot, v3h8 (&xc)[4], v3h8 (&xnx)[4], v3h8 (&xld)[4], auto rs_tag, bool more_steps) {
            constexpr int rs = decltype(rs_tag)::value;
            constexpr int prev_rs = (rs + RSC - 1) % RSC;
            if constexpr (XG) {
                if (rs == 0) load_x(xld, xq);                                    // the fragments of the step XD ahead (xld: the previous step's set)
            } else {
                if (rs == 0) load_x(xnx, xp + (more_steps ? NW * 256 : 0));      // the next step's fragments: in flight during this step
            }
            // the next consume's scale word: the next row set of this step, or set 0 of the next step
            const uint32_t szw_n = rs + 1 < RSC ? *(const uint32_t*)(sp + (size_t)(rs + 1) * SZB)
                                                : *(const uint32_t*)(sp + (more_steps ? s_stride : 0u));
            val_t lo, hi;
            const ring_t wv = slot;
            {
                // fragment j = pair j of every word w: k = 8j + 2w, +1 (w = 0..3) -- the 8 consecutive k of x slot j
                u32x4 bf[4];
                if (BITS == 4) {
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const uint32_t v = wv[w], t = v >> 8;
                        bf[0][w] = (v & 0x000f000fu) | MAGIC;    // 1024 + q
                        bf[1][w] = (v & 0x00f000f0u) | MAGIC;    // 1024 + 16 q
                        bf[2][w] = (t & 0x000f000fu) | MAGIC;
                        bf[3][w] = (t & 0x00f000f0u) | MAGIC;
                    }
                } else {
                    // 3-bit: pair e = 4 j + w (k = 2e, 2e + 1 of the lane's chunk, natural order) sits in word e / 5 at bits
                    // 3 (e % 5) of each half-word, pair 15 in bits 15 / 31 of the three words.  x is consumed raw here, so every
                    // field is shifted down to bits 0..2 (1024 + q for all of them: one scale class, no pre-scaled x)
                    uint32_t ext[16];
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        const uint32_t v = wv[i];
#pragma unroll
                        for (int f = 0; f < 5; ++f) ext[5 * i + f] = ((v >> (3 * f)) & 0x00070007u) | MAGIC;
                    }
                    ext[15] = (((wv[0] >> 15) & 0x00010001u) | ((wv[1] >> 14) & 0x00020002u) | ((wv[2] >> 13) & 0x00040004u)) | MAGIC;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int w = 0; w < 4; ++w) bf[j][w] = ext[4 * j + w];
                }
                f32x4 Plo = __builtin_amdgcn_mfma_f32_16x16x32_f16(xc[0], __builtin_bit_cast(v3h8, bf[0]), clo, 0, 0, 0);
                f32x4 Phi = __builtin_amdgcn_mfma_f32_16x16x32_f16(xc[1], __builtin_bit_cast(v3h8, bf[1]), chi, 0, 0, 0);
                Plo = __builtin_amdgcn_mfma_f32_16x16x32_f16(xc[2], __builtin_bit_cast(v3h8, bf[2]), Plo, 0, 0, 0);
                Phi = __builtin_amdgcn_mfma_f32_16x16x32_f16(xc[3], __builtin_bit_cast(v3h8, bf[3]), Phi, 0, 0, 0);
                if (rs == RSC - 1) bias_sums(xnx, nlo, nhi);    // the NEXT step's sums ride behind this step's last products
                fold(acc[prev_rs]);                             // the PREVIOUS (step, row set): its MFMAs retired long ago
                if (BITS == 3) {
                    lo = v3_pick<MB>(Plo) + v3_pick<MB>(Phi);
                    hi = v3_zero<MB>();
                } else {
                    lo = v3_pick<MB>(Plo);
                    hi = v3_pick<MB>(Phi);
                }
            }
            pv_lo = lo; pv_hi = hi; pv_B = cB; pv_szw = szw;
            szw = szw_n;
            issue(slot, std::integral_constant<int, (rs + D) % RSC>{});      // the slot's next load: number c + D of the sequence
            if (rs == RSC - 1) {                                // step done
                clo = nlo;
                chi = nhi;
                cB = zero_term(nlo, nhi);
                if (more_steps) { xp += NW * 256; sp += s_stride; }
                if constexpr (XG) xq = xq + NW * 256 <= xlast ? xq + NW * 256 : xlast;
            }
        };
        // The unrolled round: US steps = US * RSC consumes = a whole number of ring turns, an even number of steps.
        constexpr int US0 = v3_unroll_steps(D, RSC);
        constexpr int US = XG ? US0 * XP / v3_gcd(US0, XP) : US0;
        static_assert((US * RSC) % D == 0 && US % XP == 0, "a round must return every ring slot and every fragment set to its role");
        int i = 0;                                              // step counter of the wave
        auto round = [&](bool guarded) {
            v3_static_for<0, US>([&](auto u_tag) {
                constexpr int u = decltype(u_tag)::value;
                if (!guarded || i + u < nsw) {
                    const bool more = i + u + 1 < nsw;
                    v3_static_for<0, RSC>([&](auto rs_tag) {
                        constexpr int rs = decltype(rs_tag)::value;
                        constexpr int c = u * RSC + rs;
                        consume(ring[c % D], xr[u % XP], xr[(u + 1) % XP], xr[(u + XD) % XP], rs_tag, more);
                        __builtin_amdgcn_sched_barrier(0);
                    });
                }
            });
        };
        for (; i + US <= nsw; i += US) round(false);
        if (i < nsw) round(true);
        fold(acc[RSC - 1]);                    // the last (step, row set)
    }
    V3_STAMP(3);
    // the producer's partial sums of squares: waves 2 and 3 requested them (step 2b) and are the ones that know, through their
    // own waits, that they have landed -- each sums its piece HERE, in the slack before the final barrier, and leaves one float;
    // behind the barrier every wave reads two floats instead of reducing 512 (that reduction was ~0.15 us of every q|k|v and
    // gate|up launch's tail)
    float* const ssq_part = red + RSC * V3_NW_MAX * 16;              // the 64 spare bytes of the red region
    if (MB == 1 && ssq_in && (wave == 2 || wave == 3) && (wave - 2) * 256 < ssq_n) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        typedef float v3f4 __attribute__((ext_vector_type(4)));
        const v3f4 p0 = ((const v3f4*)ssql)[(wave - 2) * 64 + lane];
        const int b0 = (wave - 2) * 256 + 4 * lane;
        float sp_ = (b0 < ssq_n ? p0[0] : 0.f) + (b0 + 1 < ssq_n ? p0[1] : 0.f) + (b0 + 2 < ssq_n ? p0[2] : 0.f) + (b0 + 3 < ssq_n ? p0[3] : 0.f);
        sp_ = wave_sum(sp_);
        if (lane == 0) ssq_part[wave - 2] = sp_;
    }
    if constexpr (MB == 1) {
        if (kc == 0)
            v3_static_for<0, RSC>([&](auto r) { red[(decltype(r)::value * NW + wave) * 16 + nl] = acc[r]; });
    } else {
        const int RB = m > 8 ? 16 : 8;          // batch rows kept per (row set, wave)
        if (kc < 2 || m > 8) {                  // batch rows 4 kc + j
            v3_static_for<0, RSC>([&](auto r) {
#pragma unroll
                for (int j = 0; j < 4; ++j) red[((size_t)(decltype(r)::value * NW + wave) * RB + 4 * kc + j) * 16 + nl] = acc[r][j];
            });
        }
    }

    // ---- 5. combine the waves, finish the norm, fused epilogues (a short block's last slot is a copy: rs < RS only)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // step 2b's pieces (a wave without ring steps never waited)
    __syncthreads();
    V3_STAMP(5);
    if constexpr (MB == 2) {
        // y[i][row] for the m batch rows: wave w takes batch rows w, w + NW; lane = (row set lane / 16, row lane % 16)
        const int rs = lane >> 4, RB = m > 8 ? 16 : 8;
        for (int i = wave; i < m; i += NW)
            if (rs < RS) {
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) v += red[((size_t)(rs * NW + w) * RB + i) * 16 + nl];
                const int row = (set0 + rs) * 16 + nl;
                if (bias) v += (float)bias[row];
                yout[(size_t)i * G.nsets * 16 + row] = (f16)v;
            }
        return;
    }
    float rs_norm = 1.f;
    if (XN) {
        float sx = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) sx += ssql[w];
        rs_norm = __builtin_amdgcn_rsqf(sx * (1.f / (float)G.K) + eps);
    } else if (ssq_in) {               // the two pieces' sums (waves 2 / 3, in front of the barrier)
        const float s = ssq_part[0] + (ssq_n > 256 ? ssq_part[1] : 0.f);
        rs_norm = __builtin_amdgcn_rsqf(s * (1.f / (float)G.K) + eps);
    }
    auto row_sum = [&](int rs, int n) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) v += red[(rs * NW + w) * 16 + n];
        return v * rs_norm;
    };
    if (MODE == V3_MODE_PAIR) {
        // set g: gate rows 0..7, up rows 8..15 of the same 8 outputs -> act = silu(gate) * up, rounded like the unfused
        // sequence (gate and up to fp16 first, qeft_silu_mul arithmetic)
        if (tid < RS * 8) {
            const int rs = tid >> 3, n = tid & 7;
            float gv = row_sum(rs, n), uv = row_sum(rs, n + 8);
            if (bias) {
                gv += (float)bias[(set0 + rs) * 16 + n];
                uv += (float)bias[(set0 + rs) * 16 + 8 + n];
            }
            const f16 g16 = (f16)gv, u16 = (f16)uv;
            const f16 r16 = mul_f32_to_f16(silu_f32((float)g16), (float)u16);
            V3_STAMP_VALUE(r16);
            yout[(set0 + rs) * 8 + n] = r16;
        }
        V3_STAMP_FLUSH();
        return;
    }
    float sq = 0.f;
    if (tid < RS * 16) {
        const int row = set0 * 16 + tid;
        float v = row_sum(tid >> 4, tid & 15);
        if (bias) v += (float)bias[row];
        V3_STAMP_VALUE(v);
        if (residual) {
            v += ((const float*)epl)[tid];                          // lanes [0, 16) x 4 floats = the block's RS * 16 rows
            y32[row] = v;
            if (gamma_out) {
                ynorm[row] = mul_f32_to_f16(v, (float)((const f16*)(epl + 256))[tid]);
                sq = v * v;
            }
        } else {
            yout[row] = (f16)v;
        }
    }
    if (residual && gamma_out && wave == 0) {   // RS * 16 <= 64: the block's rows all sit in wave 0
        sq = wave_sum(sq);
        if (lane == 0) ssq_out[blockIdx.x] = sq;
    }
        V3_STAMP_FLUSH();
}
#endif  // __HIPCC__

}  // namespace qeft
