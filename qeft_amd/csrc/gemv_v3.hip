// Launch logic of the v3 decode GEMV (gemv_v3.h) and the host-side enumerator of its address arithmetic.
#include <cstdlib>

#include "gemv_v3.h"
#include "gemv_v3_dispatch.h"

namespace qeft {

static int ceil_div(int a, int b) { return (a + b - 1) / b; }
hipError_t gemv_v3_dispatch_plain(const V3Args& a, int mode, size_t smem, int depth, hipStream_t st);     // gemv_v3_plain.hip

// Blocks for `nsets` 16-row sets: one set per block while that leaves the 256 CUs two blocks each at most; wide launches
// run about 256 k long-lived blocks of ~3 sets (the ring then never drains between sets and the staging is paid once per
// block).  QEFT_GEMV_BLOCKS overrides (lab).  Never more than V3_MAX_RS sets per block.
int gemv_v3_blocks(int nsets) {
    static int forced = -1;
    if (forced < 0) {
        const char* e = getenv("QEFT_GEMV_BLOCKS");
        forced = e ? atoi(e) : 0;
    }
    int nblk;
    if (forced > 0) nblk = nsets < forced ? nsets : forced;
    else if (nsets <= 256) nblk = nsets;
    else if (nsets < 512) nblk = ceil_div(nsets, 2);     // the fullest CU carries two sets either way: one block of two (staging paid
                                                         // once) rather than two blocks (round 2, Llama-2-13B down_proj, 320 sets: 12.0 -> 10.8 us)
    else {
        int k = (nsets + 384) / 768;
        if (k < 1) k = 1;
        nblk = 256 * k;
    }
    if (ceil_div(nsets, nblk) > V3_MAX_RS) nblk = ceil_div(nsets, V3_MAX_RS);
    // even the load: with RS = ceil(nsets / nblk) sets on the fullest block, ceil(nsets / RS) blocks do the same work with
    // (almost) every block full -- gate|up of Llama-2-7B: 459 blocks of 3 sets instead of 512 of 3 or 2 (12.0 -> 11.8 us)
    if (forced <= 0 && nsets >= 512) nblk = ceil_div(nsets, ceil_div(nsets, nblk));
    return nblk;
}

// What the v3 kernel takes: whole 128-k steps, the checkpoint's r = 128 (or no outlier slice), group 128 or per-channel.
bool gemv_v3_ok(int K, int G, int n_out) { return K % 128 == 0 && K >= 128 && (n_out == 0 || (n_out == 128 && K > 128)) && (G == 128 || G == K); }

static int env_int(const char* name) {
    const char* e = getenv(name);
    return e ? atoi(e) : 0;
}

// Geometry of a launch: blocks, row sets per block, waves per block, LDS bytes.  Returns false when the configuration does
// not fit the chip's 160 KB of LDS per block (many batch rows of a long x: the caller splits the batch).
static bool gemv_v3_plan(V3Args& a, int& nw, size_t& smem) {
    const int nblk = gemv_v3_blocks(a.g.nsets);
    a.rs_cap = ceil_div(a.g.nsets, nblk);
    a.nblk = nblk;
    a.sets_q = a.g.nsets / nblk;
    a.sets_r = a.g.nsets % nblk;
    if (a.m < 1) a.m = 1;
    static const int f_nw = env_int("QEFT_GEMV_NW");      // lab override: 4 or 8
    // several blocks per CU (more than 256 blocks): 4-wave blocks, one wave per SIMD each
    nw = f_nw == 4 || f_nw == 8 ? f_nw : (nblk > 256 && a.m <= 1 ? 4 : V3_NW);
    if (a.m > 1) nw = V3_NW;
    // one row set per block and many steps per wave (down_proj: K = 11008, 85 full steps): 12 waves share the steps 8 / 7 instead of
    // 11 / 10: down_proj 7.42 -> 7.33 us (profiles/r04_gemv_lab.txt; the same file holds the correction sums tabulated up front per
    // wave instead of 4 MFMAs per step: +0.4 us on o_proj, +1.0 us on down_proj, not adopted); QEFT_GEMV_NW12=0 keeps 8 (A/B)
    static const int no_nw12 = getenv("QEFT_GEMV_NW12") && atoi(getenv("QEFT_GEMV_NW12")) == 0;
    if (!no_nw12 && !f_nw && a.m <= 1 && nblk <= 256 && a.rs_cap == 1 && a.g.nfull >= 64) nw = 12;
    a.nw = nw;
    smem = v3_lds(a.g.K, a.g.ngroups, a.g.n_out, a.rs_cap, a.m, nw, a.xn_gamma != nullptr && a.szp != nullptr, a.szp == nullptr, a.ids != nullptr, a.xg).total;
    return smem <= 160 * 1024 && nblk < 65536;       // (nblk, sets_r share dwords with rs_cap, sets_q)
}

// Batch rows one launch of this configuration can take (0: not even one): LDS holds m rows of x (twice with a gather)
int gemv_v3_max_rows(V3Args a, int m_want) {
    for (int m = m_want < V3_MAX_M ? m_want : V3_MAX_M; m >= 1; --m) {
        int nw;
        size_t smem;
        a.m = m;
        if (gemv_v3_plan(a, nw, smem)) return m;
    }
    return 0;
}

hipError_t gemv_v3_launch(V3Args a, int mode, hipStream_t st) {
    int nw;
    size_t smem;
    if (a.m > V3_MAX_M || !gemv_v3_plan(a, nw, smem)) return hipErrorInvalidValue;
    if (a.m > 1 && (mode != V3_MODE_PLAIN || a.bits == 3 || a.residual || a.ssq_in || a.xn_gamma)) return hipErrorInvalidValue;
    if (a.xg && (a.m <= 1 || a.ids)) return hipErrorInvalidValue;      // x from global memory: batch-row launches without a gather
    if (!a.szp && (a.xn_gamma || !a.scales || !a.zeros)) return hipErrorInvalidValue;     // (the zeros pointer uses xn_gamma's slot)
    static const int f_d = env_int("QEFT_GEMV_DEPTH");     // lab override: 2, or 4 (= the deep form of the block's RSC: 4 or 6 loads)
    // Ring depth (tools/gemv_v3_lab.hip, interleaved timing, profiles/r03_gemv_lab.txt): where a CU holds ONE block and a wave has
    // at least 8 loads to make, 4 loads in flight per wave (6 with three or four row sets per block: two whole steps) --
    // q|k|v 7.06 (6) / 7.44 (4) / 7.69 us (2), down_proj 7.18 (4) / 7.59 (2) (the batch-row launches likewise: 8.65 vs 8.81); short launches (o_proj: 4 loads per wave) and
    // two blocks per CU (gate|up) keep 2: 4.58 vs 4.78, 10.97 vs 11.07 / 11.50.
    const int loads_per_wave = ceil_div(a.g.nfull, nw) * a.rs_cap;
    const int depth = a.xg ? 4 : f_d == 2 || f_d == 4 ? f_d : (a.nblk <= 256 && loads_per_wave >= 8 ? 4 : 2);     // (xg: the deep-ring instantiations only)
    const bool w3 = a.bits == 3;
    g_last_variant = a.m > 1 ? (a.xg ? "gemv_v3_mb_xg" : "gemv_v3_mb") : mode == V3_MODE_PAIR ? (w3 ? "gemv_v3_w3_pair" : "gemv_v3_pair") : (w3 ? "gemv_v3_w3" : "gemv_v3");
    // plain launches (no run-time flag, one batch row): the instantiations without the flag paths (gemv_v3_plain.hip)
    if (a.m <= 1 && (v3_flags(a) & (V3_F_PERCH | V3_F_XN | V3_F_SZN | V3_F_OWIL | V3_F_GATHER)) == 0)
        return gemv_v3_dispatch_plain(a, mode, smem, depth, st);
    return w3 ? launch_b<3, true>(a, mode, a.nblk, smem, depth, st) : launch_b<4, true>(a, mode, a.nblk, smem, depth, st);
}

// ---- host-side enumeration of every address the kernel can form for a configuration (no GPU involved).
// Walks all blocks x waves x lanes x {staging pieces, ring issues incl. the clamped ones past the end, epilogue operands,
// output rows} with the SAME __host__ __device__ functions the kernel uses and counts accesses that leave their operand.
// n_rows_have: the number of rows the operands really hold (== G.nsets * 16 unless the caller is the negative control).
long long gemv_v3_count_out_of_range(const V3Geom& G, int n_rows_have, int n_ssq_in, bool xn, int bits) {
    long long bad = 0;
    const size_t qw_bytes = bits == 3 ? (size_t)(n_rows_have / 16) * G.nfull * 768 : (size_t)(n_rows_have / 4) * G.K * 2, sz_bytes = (size_t)(n_rows_have / 16) * G.ngroups * 64,
                 ow_bytes = (size_t)n_rows_have * 128 * 2, res_bytes = (size_t)n_rows_have * 4, gam_bytes = (size_t)n_rows_have * 2;
    const int nblk = gemv_v3_blocks(G.nsets);
    const int XB = v3_x_bytes(G.K), SZB = v3_sz_bytes(G.ngroups);
    for (int b = 0; b < nblk; ++b) {
        int set0, RS;
        v3_block_sets(v3_xcd_block(b, nblk), G.nsets / nblk, G.nsets % nblk, set0, RS);
        if (RS < 1 || RS > V3_MAX_RS || set0 < 0 || set0 + RS > G.nsets) { ++bad; continue; }
        for (int lane = 0; lane < 64; ++lane) {
            const int nl = lane & 15, kc = lane >> 4;
            for (int p = 0; p < (XB >> 10); ++p) bad += (size_t)v3_x_off(G, p, lane) + 16 > (size_t)G.K * 2;      // x, or an xn launch's gamma
            if (xn)
                for (int p = 0; p < (v3_xf_bytes(G.K) >> 10); ++p) bad += (size_t)v3_xf_off(G, p, lane) + 16 > (size_t)G.K * 4;
            bad += v3_epi_off(set0, RS, lane) + 16 > (lane < 16 ? res_bytes : gam_bytes);
            if (n_ssq_in > 0)
                for (int w = 2; w < 4; ++w)
                    if ((w - 2) * 256 < n_ssq_in) {
                        int v = (w - 2) * 64 + lane;
                        if (v > ((n_ssq_in - 1) >> 2)) v = (n_ssq_in - 1) >> 2;
                        bad += (size_t)v * 16 + 16 > (size_t)((n_ssq_in + 3) / 4 * 4) * 4;     // the array is padded to 4 floats
                    }
            for (int rs = 0; rs < RS; ++rs) {
                for (int j = 0; j < (SZB >> 10); ++j) bad += v3_sz_off(G, set0 + rs, j, lane) + 16 > sz_bytes;
                if (G.n_out > 0)
                    for (int j = 0; j < 4; ++j) bad += v3_ow_off(set0 + rs, j, lane) + 16 > ow_bytes;
                if (kc == 0) bad += (set0 + rs) * 16 + nl >= n_rows_have;
            }
            for (int NW : {4, 8, 12, 16})               // every wave count the kernel is (or was) instantiated with
                for (int wave = 0; wave < NW; ++wave) {
                    const int nsw = (G.nfull - wave + NW - 1) / NW;
                    const uint32_t stepb = bits == 3 ? 768u : 256u, last = bits == 3 ? v3w3_last_step_off(G) : v3_last_step_off(G);
                    uint32_t step0 = (uint32_t)wave * stepb;
                    if (step0 > last) step0 = last;
                    const size_t set_off = bits == 3 ? v3w3_set_off(G, set0) : v3_w_set_off(G, set0);
                    const size_t set_bytes = bits == 3 ? (size_t)G.nfull * 768 : (size_t)G.K * 8;
                    const uint32_t lane_off = bits == 3 ? (uint32_t)lane * 12u : v3_w_lane_off(G, nl, kc);
                    for (int rs = 0; rs < RS; ++rs)
                        for (int i = 0; i < (nsw > 0 ? nsw : 1); ++i)
                            bad += set_off + (size_t)rs * set_bytes + step0 + (size_t)i * NW * stepb + lane_off + (bits == 3 ? 12 : 16) > qw_bytes;
                }
        }
    }
    return bad;
}

// The same for a launch on the CHECKPOINT-layout operands (the reference's gemv entries): scales / scaled_zeros fp16
// [ngroups][N], oweight_interleaved [N/2][256], reorder_ids, m batch rows.  Source accesses against the operand sizes, and
// every LDS destination / LDS read of the staging transforms against the carve-up (v3_lds).
long long gemv_v3_count_out_of_range_ckpt(const V3Geom& G, int n_rows_have, int m, bool gather) {
    long long bad = 0;
    const size_t sc_bytes = (size_t)G.ngroups * n_rows_have * 2, il_bytes = (size_t)(n_rows_have / 2) * 256 * 2,
                 x_bytes = (size_t)m * G.K * 2, ids_bytes = (size_t)G.K * 4;
    const int nblk = gemv_v3_blocks(G.nsets), rs_cap = ceil_div(G.nsets, nblk);
    const int XB = v3_x_bytes(G.K), SRB = v3_szraw_bytes(G.ngroups), XS = v3_x_stride(G.K, m);
    // (the strided scale rows assume the operand really has G.nsets * 16 columns: a shrunk operand is the negative control)
    // x fragments read by the lanes from global memory (xg launches: m > 1, no gather): lane (batch row nl clamped to m - 1, chunk kc)
    // reads 64 bytes of every 128-k step, the outlier step included
    if (!gather && m > 1)
        for (int lane = 0; lane < 64; ++lane) {
            const int nl = lane & 15, kc = lane >> 4, row = nl < m - 1 ? nl : m - 1;
            for (int step = 0; step < G.nsteps; ++step) bad += (size_t)row * G.K * 2 + (size_t)step * 256 + kc * 64 + 64 > x_bytes;
        }
    for (int nw : {4, 8, 12, 16}) {
        const V3Lds L = v3_lds(G.K, G.ngroups, G.n_out, rs_cap, m, nw, false, true, gather);
        if (L.total > 160 * 1024) continue;     // not launched (gemv_v3_plan)
        for (int b = 0; b < nblk; ++b) {
            int set0, RS;
            v3_block_sets(v3_xcd_block(b, nblk), G.nsets / nblk, G.nsets % nblk, set0, RS);
            for (int lane = 0; lane < 64; ++lane) {
                for (int i = 0; i < m; ++i)
                    for (int p = 0; p < (XB >> 10); ++p) {
                        bad += (size_t)i * G.K * 2 + v3_x_off(G, p, lane) + 16 > x_bytes;
                        const uint32_t dst = (gather ? L.xraw + (uint32_t)i * XB : L.xs + (uint32_t)i * XS) + ((uint32_t)p << 10) + lane * 16;
                        bad += dst + 16 > (gather ? L.total : L.szl);
                    }
                if (gather)
                    for (int p = 0; p < (v3_xf_bytes(G.K) >> 10); ++p) {
                        bad += (size_t)v3_xf_off(G, p, lane) + 16 > ids_bytes;
                        bad += L.idsl + ((uint32_t)p << 10) + lane * 16 + 16 > L.xraw;
                    }
                for (int rs = 0; rs < RS; ++rs) {
                    for (int arr = 0; arr < 2; ++arr)
                        for (int j = 0; j < (SRB >> 10); ++j) {
                            bad += v3_szn_off(G, set0 + rs, j, lane) + 16 > sc_bytes;
                            bad += L.szraw + (uint32_t)(arr * rs_cap + rs) * SRB + ((uint32_t)j << 10) + lane * 16 + 16 > L.idsl;
                        }
                    if (G.n_out > 0)
                        for (int j = 0; j < 4; ++j) bad += v3_owil_off(set0 + rs, j, lane) + 16 > il_bytes;
                }
            }
        }
    }
    return bad;
}

}  // namespace qeft
