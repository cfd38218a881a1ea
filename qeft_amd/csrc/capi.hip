// extern "C" entry points declared in include/qeft_hip.h: argument validation + dispatch.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <stdint.h>

#include "../../include/qeft_hip.h"
#include "qeft_common.h"
#include "gemv_v3.h"

namespace qeft {
hipError_t gemv_v3_launch(V3Args a, int mode, hipStream_t st);
int gemv_v3_max_rows(V3Args a, int m_want);
long long gemv_v3_count_out_of_range_ckpt(const V3Geom& G, int n_rows_have, int m, bool gather);
int gemv_v3_blocks(int nsets);
bool gemv_v3_ok(int K, int G, int n_out);
long long gemv_v3_count_out_of_range(const V3Geom& G, int n_rows_have, int n_ssq_in, bool xn, int bits);
hipError_t token_begin_norm_launch(const void* embed, const void* tok, const void* rope_tab, const int* pos, void* h,
                                   void* rope_row, const void* gamma, void* hnorm, float* ssq_out, int hidden, int vocab,
                                   int max_seq, hipStream_t st);
int token_begin_norm_blocks(int hidden);
hipError_t residual_norm_launch(const void* h, const void* add, const void* gamma, void* h_out, void* hnorm, float* ssq_out,
                                int hidden, hipStream_t st);
hipError_t rmsnorm_f32_launch(const void* x, const void* gamma, void* y, int m, int H, float eps, hipStream_t st);
hipError_t rope_rows_launch(void* x, const void* cs, const void* sn, int T, int H, int row_stride, hipStream_t st);
hipError_t lm_head_f16_launch(const void* h32, const void* gamma, const void* W, void* logits, int H, int vocab, float eps,
                              hipStream_t st);
hipError_t gemv_w4_dispatch(const GemvArgs& a, int m, hipStream_t st);
hipError_t gemv_w4_group_dispatch(GemvGroupArgs g, int nparts, hipStream_t st);
hipError_t gemv_w4_silu_dispatch(const GemvArgs& a, hipStream_t st);
hipError_t gemv_w4_smallm_dispatch(const GemvArgs& a, int m, hipStream_t st);
hipError_t gemv_w3_dispatch(const GemvArgs& a, int m, hipStream_t st);
hipError_t gemv_w3_group_dispatch(GemvGroupArgs g, int nparts, hipStream_t st);
hipError_t gemv_w3_silu_dispatch(const GemvArgs& a, hipStream_t st);
hipError_t expand_w3_launch(const void* q3, void* q4, int N, int K, int n_out, hipStream_t st);
hipError_t rmsnorm_launch(const void* x, const void* add, const void* gamma, void* res_out, void* y, int m, int H,
                          float eps, hipStream_t st);
hipError_t silu_mul_launch(const void* gate, const void* up, void* out, int n, hipStream_t st);
hipError_t rope_attn_decode_launch(const void* q, const void* k, const void* v, const void* cs, const void* sn, void* kc,
                                   void* vc, const int* pos, const int* out_pos, void* out, void* ws, int n_heads,
                                   int n_kv, int max_seq, int S, int tab_rows, hipStream_t st, bool k_ft_layout = false,
                                   const float* alibi_slopes = nullptr);
size_t attn_workspace_bytes(int n_heads, int S);
hipError_t sqa_generic_launch(const void* q, const void* k, const void* v, void* kc, void* vc, const int* pos, const float* alibi,
                              void* out, int n_heads, int n_kv, int max_seq, int head_dim, hipStream_t st);
hipError_t token_begin_launch(const void* embed, const void* tok, const void* rope_tab, const int* pos, void* h,
                              void* rope_row, int hidden, int vocab, int max_seq, hipStream_t st);
hipError_t token_end_launch(const void* logits, void* tok, int* pos, int vocab, int greedy, hipStream_t st);
extern unsigned long long* g_attn_dbg;
hipError_t dequant_w4_launch(const void* qw, const void* scales, const void* zeros, const void* ow, void* out, int N,
                             int K, int G, int n_out, hipStream_t st);
hipError_t pack_oweight_launch(const void* ow, void* il, int N, int R, hipStream_t st);
hipError_t pack_scales_launch(const void* scales, const void* zeros, void* out, int N, int ngroups, hipStream_t st);
extern const void* const kSiluPair64;
bool gemm_w4_pair64_ok(int M, int n2, int K, int G, int n_out);
bool gemm_ws_supported(int m, int n, int k, int group_size, int n_out);                                  // gemm_ws.hip
hipError_t gemm_ws_launch(const void* x, const void* qweight, const void* scales, const void* zeros, const void* oweight, const void* bias,
                          void* y, int m, int n, int k, int n_out, hipStream_t st);
hipError_t gemm_w4_launch(const void* x, const void* qw, const void* scales, const void* zeros, const void* ow,
                          const void* bias, void* y, int M, int N, int K, int G, int n_out, hipStream_t st,
                          void* workspace = nullptr, size_t workspace_bytes = 0, const void* silu_gate = nullptr,
                          bool* fused_epilogue = nullptr, int bits = 4);
int gemm_w3_native_tile(int M, int N, int K, int G, int n_out);
bool gemm_w3_dx_native(int M, int N, int K, int G, int n_out);
int gemm_w4_split(int M, int N, int K, int n_out);
int gemm_v3_split(int M, int N, int K, int n_out);
hipError_t gemm_w4_dx_launch(const void* dy, const void* qw, const void* scales, const void* zeros, const void* ow,
                             void* dx, int M, int N, int K, int G, int n_out, hipStream_t st, void* workspace = nullptr,
                             size_t workspace_bytes = 0, int bits = 4);
int gemm_w4_dx_split(int M, int N, int K);
hipError_t grad_oweight_launch(const void* dy, const void* x, void* dow, int M, int N, int K, int n_out,
                               hipStream_t st);
}  // namespace qeft

static thread_local int g_last_hip_error = 0;
thread_local const char* qeft::g_last_variant = "";
// Few rows: the GEMM entry streams the weights once per 16 rows through the MFMA GEMV instead of 128-row GEMM tiles.
// Measured crossover (tools/bench_gemm.py): one 16-row slice costs ~13 us per 2^24 weights, the
// GEMM at M <= 128 is latency-bound at ~54 us per 4096 of K whatever N is -> the GEMV route wins while
// slices * N <= 16384 (always for one slice).
// With a split-K workspace (qeft_gemm_w4_ws) the GEMM itself is no longer latency-bound at these sizes (M = 128:
// 24 / 31 / 46 us on the three Llama-2-7B shapes) and wins from the second slice on.
static bool small_m_route(int m, int n, bool have_workspace = false) {
    if (m <= 16) return true;
    if (have_workspace) return false;
    const int slices = (m + 15) / 16;
    return m <= 64 && (long long)slices * n <= 16384;
}

static int finish(hipError_t e) {
    if (e == hipSuccess) return QEFT_OK;
    g_last_hip_error = (int)e;
    return QEFT_ERR_LAUNCH;
}

// QEFT_GEMV_V3=0: the reference's gemv entries keep the round-1 kernels (A/B timing, and the parity tests of those kernels)
static bool v3_route_enabled() {
    static const int on = [] { const char* e = getenv("QEFT_GEMV_V3"); return e ? atoi(e) : 1; }();
    return on != 0;
}

// QEFT_GEMM_WS=0: 17 .. 64 rows keep the round-3 routes (split-K GEMM tiles / the round-1 small-M GEMV) (A/B timing)
static bool ws_route_enabled() {
    static const int on = [] { const char* e = getenv("QEFT_GEMM_WS"); return e ? atoi(e) : 1; }();
    return on != 0;
}

// QEFT_GEMV_XG=0: batch rows whose x does not fit the LDS go as several launches / to the split-K GEMM tier (A/B timing)
static int xg_route_mode() {      // 0: never, 1: where the x rows do not fit the LDS, 2 (lab): every batch-row launch without a gather
    static const int on = [] { const char* e = getenv("QEFT_GEMV_XG"); return e ? atoi(e) : 1; }();
    return on;
}
static bool xg_route_enabled() { return xg_route_mode() != 0; }

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

static int check_common(int n, int k, int g, int n_out) {
    if (n <= 0 || k <= 0 || n % 4 != 0 || k % 64 != 0) return QEFT_ERR_SHAPE;
    if (g <= 0 || g % 32 != 0 || k % g != 0) return QEFT_ERR_GROUP;
    if (n_out < 0 || n_out % 32 != 0 || n_out >= k) return QEFT_ERR_SHAPE;
    if (n_out > 0 && n % 8 != 0) return QEFT_ERR_SHAPE;
    return QEFT_OK;
}

extern "C" {

int qeft_abi_version(void) { return 1; }

int qeft_last_hip_error(void) { return g_last_hip_error; }

const char* qeft_last_variant(void) { return qeft::g_last_variant; }

const char* qeft_error_string(int code) {
    switch (code) {
        case QEFT_OK: return "ok";
        case QEFT_ERR_BATCH: return "Unsupported batch size for gemv kernel.";
        case QEFT_ERR_SHAPE: return "unsupported shape (need N % 4 == 0 (8 with outliers), K % 64 == 0, n_out % 32 == 0, n_out < K)";
        case QEFT_ERR_GROUP: return "unsupported group size (need a multiple of 32 that divides K)";
        case QEFT_ERR_NULL: return "required pointer is NULL";
        case QEFT_ERR_LAUNCH: return "HIP kernel launch failed (see qeft_last_hip_error)";
        case QEFT_ERR_ALIGN: return "pointer is not 16-byte aligned";
        default: return "unknown error";
    }
}

// m batch rows (1..16 per launch) through the v3 decode GEMV on the operands as the CHECKPOINT holds them (gemv_v3.h): the
// scales staged raw and packed in LDS (or the sz_packed shadow when the caller has one), the outlier slab from
// oweight_interleaved (the gemv entries) or the plain oweight [n, r] (the GEMM entries), the o_proj gather inside the
// launch, the batch rows as A rows of the same MFMAs.  A batch whose x rows do not fit the block's LDS goes in several
// launches, split evenly; more than `max_launches` of them: not taken (V3_NOT_TAKEN), like a shape the kernel does not serve.
constexpr int V3_NOT_TAKEN = -1000;
// launches v3_rows would make for m rows of this shape on checkpoint-layout operands (0: the kernel does not serve it)
static int v3_rows_launches(int m, int n, int k, int group_size, int n_out) {
    if (m < 1 || !v3_route_enabled() || n <= 0 || n % 16 != 0 || !qeft::gemv_v3_ok(k, group_size, n_out)) return 0;
    qeft::V3Args v{};
    v.g.K = k;
    v.g.n_out = n_out;
    v.g.nsteps = k / 128;
    v.g.nfull = (k - n_out) / 128;
    v.g.ngroups = group_size == k ? 1 : k / 128;
    v.g.nsets = n / 16;
    int mmax = qeft::gemv_v3_max_rows(v, m);
    if (mmax < m && m > 1 && xg_route_enabled()) {
        v.xg = true;
        if (qeft::gemv_v3_max_rows(v, m) >= m) mmax = m;
    }
    return mmax < 1 ? 0 : (m + mmax - 1) / mmax;
}
constexpr int kGemmRowsOnGemvLaunches = 2;      // the GEMM entries put up to 16 rows on the decode GEMV if that takes two launches at most
static int v3_rows(const void* x, const void* qweight, const void* scales, const void* scaled_zeros, const void* oweight_plain,
                   const void* oweight_il, const void* bias, const int* reorder_ids, const void* sz_packed, void* y, int m, int n,
                   int k, int group_size, int n_out, qeft_stream_t stream, int max_launches) {
    if (!v3_route_enabled() || n % 16 != 0 || !qeft::gemv_v3_ok(k, group_size, n_out) || !aligned16(scales) || !aligned16(scaled_zeros) ||
        (reorder_ids && !aligned16(reorder_ids)))
        return V3_NOT_TAKEN;
    qeft::V3Args v{};
    v.g.K = k;
    v.g.n_out = n_out;
    v.g.nsteps = k / 128;
    v.g.nfull = (k - n_out) / 128;
    v.g.ngroups = group_size == k ? 1 : k / 128;
    v.g.nsets = n / 16;
    v.qw = (const uint8_t*)qweight;
    v.szp = (const uint8_t*)sz_packed;
    v.scales = (const qeft::f16*)scales;
    v.zeros = (const qeft::f16*)scaled_zeros;
    v.ow = (const uint8_t*)oweight_plain;
    v.ow_il = (const uint8_t*)oweight_il;
    v.bias = (const qeft::f16*)bias;
    v.ids = reorder_ids;
    int mmax = qeft::gemv_v3_max_rows(v, m);
    if ((mmax < m || xg_route_mode() == 2) && m > 1 && !reorder_ids && xg_route_enabled()) {
        // the x rows do not fit the block's LDS: the lanes read their fragments from global memory instead (one launch, any K)
        v.xg = true;
        const int mx = qeft::gemv_v3_max_rows(v, m);
        if (mx >= m) mmax = mx; else v.xg = false;
    }
    if (mmax < 1) return V3_NOT_TAKEN;
    const int nlaunch = (m + mmax - 1) / mmax, per = (m + nlaunch - 1) / nlaunch;      // 7 rows as 4 + 3, not 6 + 1
    if (nlaunch > max_launches) return V3_NOT_TAKEN;
    for (int m0 = 0; m0 < m; m0 += per) {
        v.m = m - m0 < per ? m - m0 : per;
        v.x = (const qeft::f16*)x + (size_t)m0 * k;
        v.y = (qeft::f16*)y + (size_t)m0 * n;
        if (const hipError_t e = qeft::gemv_v3_launch(v, qeft::V3_MODE_PLAIN, (hipStream_t)stream)) return finish(e);
    }
    return QEFT_OK;
}

static int gemv_fused_impl(const void* x, const void* qweight, const void* scales, const void* scaled_zeros,
                           const void* oweight_il, const void* bias, const int* reorder_ids, const void* residual,
                           const void* sz_packed, void* y, int m, int n, int k, int group_size, int n_out,
                           qeft_stream_t stream, int bits = 4) {
    if (m < 1 || (bits == 4 && m > 7)) return QEFT_ERR_BATCH;
    if (int e = check_common(n, k, group_size, n_out)) return e;
    if (!x || !qweight || !scales || !scaled_zeros || !y || (n_out > 0 && !oweight_il)) return QEFT_ERR_NULL;
    if (!aligned16(x) || !aligned16(qweight) || (n_out > 0 && !aligned16(oweight_il))) return QEFT_ERR_ALIGN;
    if (sz_packed && !aligned16(sz_packed)) return QEFT_ERR_ALIGN;
    if (bits == 4 && !residual) {
        const int r = v3_rows(x, qweight, scales, scaled_zeros, nullptr, oweight_il, bias, reorder_ids, sz_packed, y, m, n, k, group_size,
                              n_out, stream, 1 << 30);
        if (r != V3_NOT_TAKEN) return r;
    }
    qeft::GemvArgs a;
    a.x = (const qeft::f16*)x;
    a.qw = (const uint8_t*)qweight;
    a.scales = (const qeft::f16*)scales;
    a.zeros = (const qeft::f16*)scaled_zeros;
    a.ow_il = (const qeft::f16*)oweight_il;
    a.bias = (const qeft::f16*)bias;
    a.ids = reorder_ids;
    a.residual = (const qeft::f16*)residual;
    a.y = (qeft::f16*)y;
    a.N = n;
    a.K = k;
    a.G = group_size;
    a.n_out = n_out;
    a.xt_aux = nullptr;
    a.xt_eps = 0.f;
    a.sz_blk = (const uint32_t*)sz_packed;
    a.dbg = nullptr;
    a.dbg2 = nullptr;
    a.ow_plain = nullptr;
    a.m_rt = 1;
    if (group_size != k && (group_size & (group_size - 1)) != 0) return QEFT_ERR_GROUP;  // GEMV: power of two or == K
    a.gshift = (group_size == k) ? 31 : __builtin_ctz(group_size);
    if (bits == 3) {
        if (m < 1 || reorder_ids) return QEFT_ERR_SHAPE;   // the gather is the caller's (qlinear.py:275) for w3
        return finish(qeft::gemv_w3_dispatch(a, m, (hipStream_t)stream));
    }
    return finish(qeft::gemv_w4_dispatch(a, m, (hipStream_t)stream));
}

int qeft_gemv_w4_fused(const void* x, const void* qweight, const void* scales, const void* scaled_zeros,
                       const void* oweight_il, const void* bias, const int* reorder_ids, const void* residual,
                       const void* sz_packed, void* y, int m, int n, int k, int group_size, int n_out,
                       qeft_stream_t stream) {
    return gemv_fused_impl(x, qweight, scales, scaled_zeros, oweight_il, bias, reorder_ids, residual, sz_packed, y, m, n,
                           k, group_size, n_out, stream);
}

int qeft_gemv_w4(const void* x, const void* qweight, const void* scales, const void* scaled_zeros, void* y, int m,
                 int n, int k, int group_size, qeft_stream_t stream) {
    return gemv_fused_impl(x, qweight, scales, scaled_zeros, nullptr, nullptr, nullptr, nullptr, nullptr, y, m, n, k,
                           group_size, 0, stream);
}

int qeft_gemv_w4_qeft(const void* x, const void* qweight, const void* scales, const void* scaled_zeros,
                      const void* oweight_il, void* y, int m, int n, int k, int group_size, int n_out,
                      qeft_stream_t stream) {
    return gemv_fused_impl(x, qweight, scales, scaled_zeros, oweight_il, nullptr, nullptr, nullptr, nullptr, y, m, n, k,
                           group_size, n_out, stream);
}

static int gemm_impl(const void* x, const void* qweight, const void* scales, const void* scaled_zeros,
                     const void* oweight, const void* bias, void* y, int m, int n, int k, int group_size, int n_out,
                     qeft_stream_t stream, void* workspace, size_t workspace_bytes, const void* silu_gate = nullptr,
                     bool* fused_epilogue = nullptr) {
    if (fused_epilogue) *fused_epilogue = false;
    if (m < 1) return QEFT_ERR_SHAPE;
    if (!oweight) n_out = 0;
    if (int e = check_common(n, k, group_size, n_out)) return e;
    if (!x || !qweight || !scales || !scaled_zeros || !y) return QEFT_ERR_NULL;
    if (!aligned16(x) || !aligned16(qweight) || (n_out > 0 && !aligned16(oweight))) return QEFT_ERR_ALIGN;
    if (m <= qeft::V3_MAX_M && (n_out > 0) == (oweight != nullptr)) {
        // up to 16 rows: the batch rows ride as A rows of the decode GEMV's MFMAs -- one weight stream (two launches at most where
        // the caller brought the split-K workspace: a long x whose rows do not fit the LDS then falls to the split-K GEMM tier
        // below rather than stream the weights three times; without a workspace three streams still beat an unsplit 128-row tile)
        const int r = v3_rows(x, qweight, scales, scaled_zeros, oweight, nullptr, bias, nullptr, nullptr, y, m, n, k, group_size, n_out,
                              stream, workspace != nullptr && workspace_bytes > 0 && (group_size & (group_size - 1)) == 0
                                          ? kGemmRowsOnGemvLaunches : (1 << 30));      // (per-channel scales: no split-K tier to fall to)
        if (r != V3_NOT_TAKEN) return r;
    }
    if (ws_route_enabled() && (n_out > 0) == (oweight != nullptr) && aligned16(scales) && aligned16(scaled_zeros) &&
        qeft::gemm_ws_supported(m, n, k, group_size, n_out)) {
        // 17 .. 64 rows: the weight-stationary tier (gemm_ws.hip) -- one launch, one pass over the packed weights, no workspace
        return finish(qeft::gemm_ws_launch(x, qweight, scales, scaled_zeros, oweight, bias, y, m, n, k, n_out, (hipStream_t)stream));
    }
    if (small_m_route(m, n, workspace != nullptr && workspace_bytes > 0) && (n_out > 0) == (oweight != nullptr) &&
        !(m <= qeft::V3_MAX_M && v3_route_enabled() && n % 16 == 0 && qeft::gemv_v3_ok(k, group_size, n_out))) {
        // few rows: stream the weights once per 16 rows through the MFMA GEMV instead of 128-row GEMM tiles.
        // (gemm_4bit semantics with oweight == NULL and a non-zero slice -- dead nibbles -- stays on the GEMM kernel.)
        qeft::GemvArgs a{};
        a.x = (const qeft::f16*)x;
        a.qw = (const uint8_t*)qweight;
        a.scales = (const qeft::f16*)scales;
        a.zeros = (const qeft::f16*)scaled_zeros;
        a.ow_il = nullptr;
        a.ow_plain = (const qeft::f16*)oweight;
        a.bias = (const qeft::f16*)bias;
        a.y = (qeft::f16*)y;
        a.N = n;
        a.K = k;
        a.G = group_size;
        a.n_out = n_out;
        a.gshift = (group_size == k) ? 31 : ((group_size & (group_size - 1)) == 0 ? __builtin_ctz(group_size) : 0);
        a.m_rt = 1;
        const hipError_t e = qeft::gemv_w4_smallm_dispatch(a, m, (hipStream_t)stream);
        if (e == hipSuccess) return QEFT_OK;
        if (e != hipErrorNotSupported) return finish(e);
    }
    return finish(qeft::gemm_w4_launch(x, qweight, scales, scaled_zeros, oweight, bias, y, m, n, k, group_size, n_out,
                                       (hipStream_t)stream, workspace, workspace_bytes, silu_gate, fused_epilogue));
}

int qeft_gemm_w4(const void* x, const void* qweight, const void* scales, const void* scaled_zeros,
                 const void* oweight, const void* bias, void* y, int m, int n, int k, int group_size, int n_out,
                 qeft_stream_t stream) {
    return gemm_impl(x, qweight, scales, scaled_zeros, oweight, bias, y, m, n, k, group_size, n_out, stream, nullptr, 0);
}

int qeft_gemm_w4_silu_mul(const void* x, const void* qweight, const void* scales, const void* scaled_zeros,
                          const void* oweight, const void* bias, const void* gate, void* y, int m, int n, int k,
                          int group_size, int n_out, qeft_stream_t stream) {
    if (!gate) return QEFT_ERR_NULL;
    if (n % 8 != 0 || (long long)m * n > 0x7fffffffLL) return QEFT_ERR_SHAPE;
    if (!aligned16(gate) || !aligned16(y)) return QEFT_ERR_ALIGN;
    bool fused = false;       // reported by the launch itself (the variant string is diagnostic only)
    const int e = gemm_impl(x, qweight, scales, scaled_zeros, oweight, bias, y, m, n, k, group_size, n_out, stream, nullptr, 0,
                            gate, &fused);
    if (e != QEFT_OK || fused) return e;
    // a tier without the fused epilogue: the product is in y, finish in place
    return finish(qeft::silu_mul_launch(gate, y, y, m * n, (hipStream_t)stream));
}

int qeft_gemm_w4_gateup_supported(int m, int n2, int k, int group_size, int n_out) {
    if (m < 1 || check_common(n2, k, group_size, n_out) != QEFT_OK || (long long)m * n2 > 0x7fffffffLL) return 0;
    return qeft::gemm_w4_pair64_ok(m, n2, k, group_size, n_out) ? 1 : 0;
}

int qeft_gemm_w4_gateup(const void* x, const void* qweight, const void* scales, const void* scaled_zeros,
                        const void* oweight, const void* bias, void* y, int m, int n2, int k, int group_size, int n_out,
                        qeft_stream_t stream) {
    if (!oweight) n_out = 0;
    if (!qeft_gemm_w4_gateup_supported(m, n2, k, group_size, n_out)) return QEFT_ERR_SHAPE;
    if (!x || !qweight || !scales || !scaled_zeros || !y) return QEFT_ERR_NULL;
    if (!aligned16(x) || !aligned16(qweight) || !aligned16(y) || (n_out > 0 && !aligned16(oweight))) return QEFT_ERR_ALIGN;
    return finish(qeft::gemm_w4_launch(x, qweight, scales, scaled_zeros, oweight, bias, y, m, n2, k, group_size, n_out,
                                       (hipStream_t)stream, nullptr, 0, qeft::kSiluPair64));
}

/* ---- 3-bit extension: GEMM forward / dX on the 3-bit stream itself (the loader-wave tiers), no expansion pass */
int qeft_gemm_w3_supported(int m, int n, int k, int group_size, int n_out) {
    if (m < 1 || n < 16 || k < 128 || group_size < 1 || n_out < 0) return 0;
    return qeft::gemm_w3_native_tile(m, n, k, group_size, n_out) != 0;
}

int qeft_gemm_w3(const void* x, const void* qweight3, const void* scales, const void* scaled_zeros, const void* oweight,
                 const void* bias, void* y, int m, int n, int k, int group_size, int n_out, qeft_stream_t stream) {
    if (!oweight) n_out = 0;
    if (!qeft_gemm_w3_supported(m, n, k, group_size, n_out)) return QEFT_ERR_SHAPE;
    if (!x || !qweight3 || !scales || !scaled_zeros || !y) return QEFT_ERR_NULL;
    if (!aligned16(x) || !aligned16(qweight3) || !aligned16(y) || (n_out > 0 && !aligned16(oweight))) return QEFT_ERR_ALIGN;
    return finish(qeft::gemm_w4_launch(x, qweight3, scales, scaled_zeros, oweight, bias, y, m, n, k, group_size, n_out,
                                       (hipStream_t)stream, nullptr, 0, nullptr, nullptr, 3));
}

int qeft_gemm_w3_dx_supported(int m, int n, int k, int group_size, int n_out) {
    if (m < 1 || n < 16 || k < 128 || group_size < 1 || n_out < 0) return 0;
    return qeft::gemm_w3_dx_native(m, n, k, group_size, n_out) ? 1 : 0;
}

int qeft_gemm_w3_dx(const void* dy, const void* qweight3, const void* scales, const void* scaled_zeros, const void* oweight,
                    void* dx, int m, int n, int k, int group_size, int n_out, qeft_stream_t stream) {
    if (!oweight) n_out = 0;
    if (!qeft_gemm_w3_dx_supported(m, n, k, group_size, n_out)) return QEFT_ERR_SHAPE;
    if (!dy || !qweight3 || !scales || !scaled_zeros || !dx) return QEFT_ERR_NULL;
    if (!aligned16(dy) || (reinterpret_cast<uintptr_t>(qweight3) & 3u) || !aligned16(dx) || (n_out > 0 && !aligned16(oweight))) return QEFT_ERR_ALIGN;
    return finish(qeft::gemm_w4_dx_launch(dy, qweight3, scales, scaled_zeros, oweight, dx, m, n, k, group_size, n_out,
                                          (hipStream_t)stream, nullptr, 0, 3));
}

long long qeft_gemm_w4_workspace_bytes(int m, int n, int k, int n_out) {
    if (m < 1 || n < 1 || k < 1 || n_out < 0 || n_out >= k) return 0;
    if (m <= qeft::V3_MAX_M && k % 128 == 0 && n % 16 == 0 && (n_out == 0 || n_out == 128)) {
        // up to 16 rows of a shape the decode GEMV serves (group size 128 assumed: the larger LDS plan): no workspace if gemm_impl
        // will put them there, the split-K tier's otherwise (three weight streams would cost more than its partial sums)
        const int l = v3_rows_launches(m, n, k, 128, n_out);
        if (l >= 1 && l <= kGemmRowsOnGemvLaunches) return 0;
    } else if (ws_route_enabled() && qeft::gemm_ws_supported(m, n, k, 128, n_out)) {
        return 0;       // (group size 128 assumed, as above: the weight-stationary tier needs no workspace)
    } else if (small_m_route(m, n, true)) {
        return 0;
    }
    int s = qeft::gemm_w4_split(m, n, k, n_out);
    const int s3 = qeft::gemm_v3_split(m, n, k, n_out);       // the loader-wave tier's split (whichever tier the launch takes, it fits)
    if (s3 > s) s = s3;
    return s > 1 ? (long long)s * m * n * 4 : 0;
}

int qeft_gemm_w4_ws(const void* x, const void* qweight, const void* scales, const void* scaled_zeros,
                    const void* oweight, const void* bias, void* y, void* workspace, long long workspace_bytes, int m,
                    int n, int k, int group_size, int n_out, qeft_stream_t stream) {
    if (workspace && !aligned16(workspace)) return QEFT_ERR_ALIGN;
    return gemm_impl(x, qweight, scales, scaled_zeros, oweight, bias, y, m, n, k, group_size, n_out, stream, workspace,
                     workspace && workspace_bytes > 0 ? (size_t)workspace_bytes : 0);
}

static int gemm_dx_impl(const void* dy, const void* qweight, const void* scales, const void* scaled_zeros,
                        const void* oweight, void* dx, int m, int n, int k, int group_size, int n_out,
                        qeft_stream_t stream, void* workspace, size_t workspace_bytes) {
    if (m < 1) return QEFT_ERR_SHAPE;
    if (!oweight) n_out = 0;
    if (int e = check_common(n, k, group_size, n_out)) return e;
    if (n % 8 != 0) return QEFT_ERR_SHAPE;  // dy rows are read 16 bytes at a time
    if (!dy || !qweight || !scales || !scaled_zeros || !dx) return QEFT_ERR_NULL;
    if (!aligned16(dy) || !aligned16(qweight) || !aligned16(dx) || (n_out > 0 && !aligned16(oweight)))
        return QEFT_ERR_ALIGN;
    return finish(qeft::gemm_w4_dx_launch(dy, qweight, scales, scaled_zeros, oweight, dx, m, n, k, group_size, n_out,
                                          (hipStream_t)stream, workspace, workspace_bytes));
}

int qeft_gemm_w4_dx(const void* dy, const void* qweight, const void* scales, const void* scaled_zeros,
                    const void* oweight, void* dx, int m, int n, int k, int group_size, int n_out,
                    qeft_stream_t stream) {
    return gemm_dx_impl(dy, qweight, scales, scaled_zeros, oweight, dx, m, n, k, group_size, n_out, stream, nullptr, 0);
}

long long qeft_gemm_w4_dx_workspace_bytes(int m, int n, int k) {
    if (m < 1 || n < 1 || k < 64 || k % 4 != 0) return 0;
    if (k % 128 == 0 && ((m + 127) / 128) * (k / 128) >= 512) return 0;       // the 128-wide tile: no split
    const int s = qeft::gemm_w4_dx_split(m, n, k);
    return s > 1 ? (long long)s * m * k * 4 : 0;
}

int qeft_gemm_w4_dx_ws(const void* dy, const void* qweight, const void* scales, const void* scaled_zeros,
                       const void* oweight, void* dx, void* workspace, long long workspace_bytes, int m, int n, int k,
                       int group_size, int n_out, qeft_stream_t stream) {
    if (workspace && !aligned16(workspace)) return QEFT_ERR_ALIGN;
    return gemm_dx_impl(dy, qweight, scales, scaled_zeros, oweight, dx, m, n, k, group_size, n_out, stream, workspace,
                        workspace && workspace_bytes > 0 ? (size_t)workspace_bytes : 0);
}

int qeft_grad_oweight(const void* dy, const void* x, void* d_oweight_f32, int m, int n, int k, int n_out,
                      qeft_stream_t stream) {
    if (m < 1 || n <= 0 || k <= 0 || n_out <= 0 || n_out > k) return QEFT_ERR_SHAPE;
    if (!dy || !x || !d_oweight_f32) return QEFT_ERR_NULL;
    return finish(qeft::grad_oweight_launch(dy, x, d_oweight_f32, m, n, k, n_out, (hipStream_t)stream));
}

int qeft_dequant_w4(const void* qweight, const void* scales, const void* scaled_zeros, const void* oweight,
                    void* w_out, int n, int k, int group_size, int n_out, qeft_stream_t stream) {
    if (!oweight) n_out = 0;
    if (int e = check_common(n, k, group_size, n_out)) return e;
    if (!qweight || !scales || !scaled_zeros || !w_out) return QEFT_ERR_NULL;
    if (!aligned16(qweight) || !aligned16(w_out) || (n_out > 0 && !aligned16(oweight))) return QEFT_ERR_ALIGN;
    return finish(qeft::dequant_w4_launch(qweight, scales, scaled_zeros, oweight, w_out, n, k, group_size, n_out,
                                          (hipStream_t)stream));
}

int qeft_pack_scales(const void* scales, const void* scaled_zeros, void* sz_packed, int n, int k, int group_size,
                     qeft_stream_t stream) {
    if (n <= 0 || n % 16 != 0 || k <= 0 || group_size <= 0 || k % group_size != 0) return QEFT_ERR_SHAPE;
    if (group_size != 128 && group_size != k) return QEFT_ERR_GROUP;   // what the MFMA decode GEMV consumes
    if (!scales || !scaled_zeros || !sz_packed) return QEFT_ERR_NULL;
    if (!aligned16(sz_packed)) return QEFT_ERR_ALIGN;
    return finish(qeft::pack_scales_launch(scales, scaled_zeros, sz_packed, n, k / group_size, (hipStream_t)stream));
}

int qeft_pack_oweight(const void* oweight, void* oweight_il, int n, int n_out, qeft_stream_t stream) {
    if (n <= 0 || n % 8 != 0 || n_out <= 0 || n_out % 32 != 0) return QEFT_ERR_SHAPE;
    if (!oweight || !oweight_il) return QEFT_ERR_NULL;
    return finish(qeft::pack_oweight_launch(oweight, oweight_il, n, n_out, (hipStream_t)stream));
}

static int gemv_group_impl(const void* x, const void* norm_gamma, float norm_eps, int nparts,
                           const void* const* qweight, const void* const* scales, const void* const* scaled_zeros,
                           const void* const* oweight_il, const void* const* bias, const void* const* sz_packed,
                           void* const* y, const int* n, int k, int group_size, int n_out, qeft_stream_t stream, int bits) {
    if (nparts < 1 || nparts > 3) return QEFT_ERR_SHAPE;
    if (!x || !qweight || !scales || !scaled_zeros || !y || !n) return QEFT_ERR_NULL;
    if (group_size != k && (group_size & (group_size - 1)) != 0) return QEFT_ERR_GROUP;
    qeft::GemvGroupArgs g{};
    g.x = (const qeft::f16*)x;
    for (int p = 0; p < nparts; ++p) {
        if (int e = check_common(n[p], k, group_size, n_out)) return e;
        if (!qweight[p] || !scales[p] || !scaled_zeros[p] || !y[p] || (n_out > 0 && (!oweight_il || !oweight_il[p])))
            return QEFT_ERR_NULL;
        if (!aligned16(qweight[p]) || (n_out > 0 && !aligned16(oweight_il[p]))) return QEFT_ERR_ALIGN;
        g.qw[p] = (const uint8_t*)qweight[p];
        g.scales[p] = (const qeft::f16*)scales[p];
        g.zeros[p] = (const qeft::f16*)scaled_zeros[p];
        g.ow_il[p] = n_out > 0 ? (const qeft::f16*)oweight_il[p] : nullptr;
        g.bias[p] = bias ? (const qeft::f16*)bias[p] : nullptr;
        g.sz_blk[p] = sz_packed ? (const uint32_t*)sz_packed[p] : nullptr;
        if (g.sz_blk[p] && !aligned16(g.sz_blk[p])) return QEFT_ERR_ALIGN;
        g.y[p] = (qeft::f16*)y[p];
        g.N[p] = n[p];
    }
    if (!aligned16(x)) return QEFT_ERR_ALIGN;
    g.K = k;
    g.G = group_size;
    g.n_out = n_out;
    g.gshift = (group_size == k) ? 31 : __builtin_ctz(group_size);
    if (norm_gamma && !aligned16(norm_gamma)) return QEFT_ERR_ALIGN;
    g.xt_aux = (const qeft::f16*)norm_gamma;
    g.xt_eps = norm_eps;
    if (bits == 3) return finish(qeft::gemv_w3_group_dispatch(g, nparts, (hipStream_t)stream));
    return finish(qeft::gemv_w4_group_dispatch(g, nparts, (hipStream_t)stream));
}

int qeft_gemv_w4_group(const void* x, const void* norm_gamma, float norm_eps, int nparts,
                       const void* const* qweight, const void* const* scales, const void* const* scaled_zeros,
                       const void* const* oweight_il, const void* const* bias, const void* const* sz_packed,
                       void* const* y, const int* n, int k, int group_size, int n_out, qeft_stream_t stream) {
    return gemv_group_impl(x, norm_gamma, norm_eps, nparts, qweight, scales, scaled_zeros, oweight_il, bias, sz_packed, y,
                           n, k, group_size, n_out, stream, 4);
}

int qeft_gemv_w3_group(const void* x, const void* norm_gamma, float norm_eps, int nparts,
                       const void* const* qweight3, const void* const* scales, const void* const* scaled_zeros,
                       const void* const* oweight_il, const void* const* bias, const void* const* sz_packed,
                       void* const* y, const int* n, int k, int group_size, int n_out, qeft_stream_t stream) {
    return gemv_group_impl(x, norm_gamma, norm_eps, nparts, qweight3, scales, scaled_zeros, oweight_il, bias, sz_packed,
                           y, n, k, group_size, n_out, stream, 3);
}

static int gemv_silu_impl(const void* gate, const void* up, const void* qweight, const void* scales,
                          const void* scaled_zeros, const void* oweight_il, const void* bias, const void* residual,
                          const void* sz_packed, void* y, int n, int k, int group_size, int n_out, qeft_stream_t stream,
                          int bits) {
    if (int e = check_common(n, k, group_size, n_out)) return e;
    if (sz_packed && !aligned16(sz_packed)) return QEFT_ERR_ALIGN;
    if (!gate || !up || !qweight || !scales || !scaled_zeros || !y || (n_out > 0 && !oweight_il)) return QEFT_ERR_NULL;
    if (!aligned16(gate) || !aligned16(up) || !aligned16(qweight) || (n_out > 0 && !aligned16(oweight_il)))
        return QEFT_ERR_ALIGN;
    if (group_size != k && (group_size & (group_size - 1)) != 0) return QEFT_ERR_GROUP;
    qeft::GemvArgs a;
    a.x = (const qeft::f16*)gate;
    a.qw = (const uint8_t*)qweight;
    a.scales = (const qeft::f16*)scales;
    a.zeros = (const qeft::f16*)scaled_zeros;
    a.ow_il = (const qeft::f16*)oweight_il;
    a.bias = (const qeft::f16*)bias;
    a.ids = nullptr;
    a.residual = (const qeft::f16*)residual;
    a.y = (qeft::f16*)y;
    a.N = n;
    a.K = k;
    a.G = group_size;
    a.n_out = n_out;
    a.gshift = (group_size == k) ? 31 : __builtin_ctz(group_size);
    a.xt_aux = (const qeft::f16*)up;
    a.xt_eps = 0.f;
    a.sz_blk = (const uint32_t*)sz_packed;
    a.dbg = nullptr;
    a.dbg2 = nullptr;
    a.ow_plain = nullptr;
    a.m_rt = 1;
    if (bits == 3) return finish(qeft::gemv_w3_silu_dispatch(a, (hipStream_t)stream));
    return finish(qeft::gemv_w4_silu_dispatch(a, (hipStream_t)stream));
}

int qeft_gemv_w4_silu(const void* gate, const void* up, const void* qweight, const void* scales,
                      const void* scaled_zeros, const void* oweight_il, const void* bias, const void* residual,
                      const void* sz_packed, void* y, int n, int k, int group_size, int n_out, qeft_stream_t stream) {
    return gemv_silu_impl(gate, up, qweight, scales, scaled_zeros, oweight_il, bias, residual, sz_packed, y, n, k,
                          group_size, n_out, stream, 4);
}

int qeft_gemv_w3_silu(const void* gate, const void* up, const void* qweight3, const void* scales,
                      const void* scaled_zeros, const void* oweight_il, const void* bias, const void* residual,
                      const void* sz_packed, void* y, int n, int k, int group_size, int n_out, qeft_stream_t stream) {
    return gemv_silu_impl(gate, up, qweight3, scales, scaled_zeros, oweight_il, bias, residual, sz_packed, y, n, k,
                          group_size, n_out, stream, 3);
}

int qeft_gemv_w3(const void* x, const void* qweight3, const void* scales, const void* scaled_zeros,
                 const void* oweight_il, const void* bias, const void* residual, const void* sz_packed, void* y, int m,
                 int n, int k, int group_size, int n_out, qeft_stream_t stream) {
    return gemv_fused_impl(x, qweight3, scales, scaled_zeros, oweight_il, bias, nullptr, residual, sz_packed, y, m, n, k,
                           group_size, n_out, stream, 3);
}

int qeft_expand_w3(const void* qweight3, void* qweight4, int n, int k, int n_out, qeft_stream_t stream) {
    if (n <= 0 || k <= 0 || n % 16 != 0 || k % 128 != 0 || n_out < 0 || n_out % 128 != 0 || n_out >= k) return QEFT_ERR_SHAPE;
    if (!qweight3 || !qweight4) return QEFT_ERR_NULL;
    if (!aligned16(qweight4) || (reinterpret_cast<uintptr_t>(qweight3) & 3u)) return QEFT_ERR_ALIGN;
    return finish(qeft::expand_w3_launch(qweight3, qweight4, n, k, n_out, (hipStream_t)stream));
}

int qeft_rmsnorm(const void* x, const void* add, const void* gamma, void* res_out, void* y, int m, int hidden,
                 float eps, qeft_stream_t stream) {
    if (m < 1 || hidden < 8 || hidden % 8 != 0) return QEFT_ERR_SHAPE;
    if (!x || !gamma || !y) return QEFT_ERR_NULL;
    if (!aligned16(x) || !aligned16(y) || !aligned16(gamma) || (add && !aligned16(add)) || (res_out && !aligned16(res_out)))
        return QEFT_ERR_ALIGN;
    return finish(qeft::rmsnorm_launch(x, add, gamma, res_out, y, m, hidden, eps, (hipStream_t)stream));
}

int qeft_rmsnorm_f32(const void* x32, const void* gamma, void* y, int m, int hidden, float eps, qeft_stream_t stream) {
    if (m < 1 || hidden < 8 || hidden % 8 != 0) return QEFT_ERR_SHAPE;
    if (!x32 || !gamma || !y) return QEFT_ERR_NULL;
    if (!aligned16(x32) || !aligned16(y) || !aligned16(gamma)) return QEFT_ERR_ALIGN;
    return finish(qeft::rmsnorm_f32_launch(x32, gamma, y, m, hidden, eps, (hipStream_t)stream));
}

int qeft_rope_rows(void* x, const void* cos_tab, const void* sin_tab, int t, int n_heads, int row_stride,
                   qeft_stream_t stream) {
    if (t < 1 || n_heads < 1 || row_stride < n_heads * 128) return QEFT_ERR_SHAPE;
    if (!x || !cos_tab || !sin_tab) return QEFT_ERR_NULL;
    return finish(qeft::rope_rows_launch(x, cos_tab, sin_tab, t, n_heads, row_stride, (hipStream_t)stream));
}

int qeft_lm_head_f16(const void* h32, const void* gamma, const void* weight, void* logits, int hidden, int vocab, float eps,
                     qeft_stream_t stream) {
    if (vocab < 1 || hidden < 512 || hidden % 512 != 0) return QEFT_ERR_SHAPE;
    if (!h32 || !gamma || !weight || !logits) return QEFT_ERR_NULL;
    if (!aligned16(h32) || !aligned16(gamma) || !aligned16(weight)) return QEFT_ERR_ALIGN;
    hipError_t e = qeft::lm_head_f16_launch(h32, gamma, weight, logits, hidden, vocab, eps, (hipStream_t)stream);
    if (e == hipErrorInvalidValue) return QEFT_ERR_SHAPE;
    return finish(e);
}

int qeft_silu_mul(const void* gate, const void* up, void* out, int n, qeft_stream_t stream) {
    if (n < 8 || n % 8 != 0) return QEFT_ERR_SHAPE;
    if (!gate || !up || !out) return QEFT_ERR_NULL;
    if (!aligned16(gate) || !aligned16(up) || !aligned16(out)) return QEFT_ERR_ALIGN;
    return finish(qeft::silu_mul_launch(gate, up, out, n, (hipStream_t)stream));
}

void qeft_debug_attn_stamps(void* p) { qeft::g_attn_dbg = (unsigned long long*)p; }

int qeft_attn_workspace_bytes(int n_heads, int n_split) {
    if (n_heads < 1 || n_heads > 4096 || n_split < 1 || n_split > 8) return 0;
    return (int)qeft::attn_workspace_bytes(n_heads, n_split);
}

int qeft_rope_attn_decode(const void* q, const void* k, const void* v, const void* cos_tab, const void* sin_tab,
                          int tab_rows, void* k_cache, void* v_cache, const int* pos, const int* out_pos, void* out,
                          void* workspace, int n_split, int n_heads, int n_kv_heads, int max_seq, qeft_stream_t stream) {
    if (tab_rows != 1 && tab_rows < max_seq) return QEFT_ERR_SHAPE;
    if (n_heads < 1 || n_kv_heads < 1 || n_heads % n_kv_heads != 0 || max_seq < 16 || max_seq % 16 != 0 || max_seq > 32768)
        return QEFT_ERR_SHAPE;
    if (n_split != 1 && n_split != 2 && n_split != 4 && n_split != 8) return QEFT_ERR_SHAPE;
    if (!q || !k || !v || !cos_tab || !sin_tab || !k_cache || !v_cache || !pos || !out) return QEFT_ERR_NULL;
    if (n_split > 1 && !workspace) return QEFT_ERR_NULL;
    if (!aligned16(k_cache) || !aligned16(v_cache) || !aligned16(workspace)) return QEFT_ERR_ALIGN;
    return finish(qeft::rope_attn_decode_launch(q, k, v, cos_tab, sin_tab, k_cache, v_cache, pos, out_pos, out, workspace,
                                                n_heads, n_kv_heads, max_seq, n_split, tab_rows, (hipStream_t)stream));
}

static int sqa_impl(const void* q, const void* k, const void* v, const void* cos_tab, const void* sin_tab, int tab_rows, void* k_cache_ft,
                    void* v_cache, const int* pos, void* out, int n_heads, int n_kv_heads, int max_seq, const float* alibi_slopes,
                    qeft_stream_t stream) {
    if (tab_rows != 1 && tab_rows < max_seq) return QEFT_ERR_SHAPE;
    if (n_heads < 1 || n_kv_heads < 1 || n_heads % n_kv_heads != 0 || max_seq < 16 || max_seq % 16 != 0 || max_seq > 32768)
        return QEFT_ERR_SHAPE;
    if (!q || !k || !v || !cos_tab || !sin_tab || !k_cache_ft || !v_cache || !pos || !out) return QEFT_ERR_NULL;
    if (!aligned16(k_cache_ft) || !aligned16(v_cache)) return QEFT_ERR_ALIGN;
    return finish(qeft::rope_attn_decode_launch(q, k, v, cos_tab, sin_tab, k_cache_ft, v_cache, pos, nullptr, out, nullptr,
                                                n_heads, n_kv_heads, max_seq, 1, tab_rows, (hipStream_t)stream, true, alibi_slopes));
}

int qeft_single_query_attention(const void* q, const void* k, const void* v, const void* cos_tab, const void* sin_tab,
                                int tab_rows, void* k_cache_ft, void* v_cache, const int* pos, void* out, int n_heads,
                                int n_kv_heads, int max_seq, qeft_stream_t stream) {
    return sqa_impl(q, k, v, cos_tab, sin_tab, tab_rows, k_cache_ft, v_cache, pos, out, n_heads, n_kv_heads, max_seq, nullptr, stream);
}

int qeft_single_query_attention_generic(const void* q, const void* k, const void* v, void* k_cache_ft, void* v_cache, const int* pos,
                                        void* out, int n_heads, int n_kv_heads, int max_seq, int head_dim, const float* alibi_slopes,
                                        qeft_stream_t stream) {
    if (!q || !k || !v || !k_cache_ft || !v_cache || !pos || !out) return QEFT_ERR_NULL;
    if (n_heads < 1 || n_kv_heads < 1 || n_heads % n_kv_heads != 0 || head_dim < 8 || head_dim > 256 || head_dim % 8 != 0 ||
        max_seq < 1 || max_seq > 32768)
        return QEFT_ERR_SHAPE;
    if (!aligned16(k_cache_ft) || !aligned16(v_cache)) return QEFT_ERR_ALIGN;
    qeft::g_last_variant = "sqa_generic";
    return finish(qeft::sqa_generic_launch(q, k, v, k_cache_ft, v_cache, pos, alibi_slopes, out, n_heads, n_kv_heads, max_seq, head_dim,
                                           (hipStream_t)stream));
}

int qeft_single_query_attention_alibi(const void* q, const void* k, const void* v, const void* cos_tab, const void* sin_tab,
                                      int tab_rows, void* k_cache_ft, void* v_cache, const int* pos, void* out, int n_heads,
                                      int n_kv_heads, int max_seq, const float* alibi_slopes, qeft_stream_t stream) {
    if (!alibi_slopes) return QEFT_ERR_NULL;
    return sqa_impl(q, k, v, cos_tab, sin_tab, tab_rows, k_cache_ft, v_cache, pos, out, n_heads, n_kv_heads, max_seq, alibi_slopes, stream);
}

// ---- v3 decode linear (gemv_v3.h): fills the geometry, validates, launches
static int v3_geom(qeft::V3Geom& G, int n, int k, int group_size, int n_out, int mode) {
    if (mode != qeft::V3_MODE_PLAIN && mode != qeft::V3_MODE_PAIR) return QEFT_ERR_SHAPE;
    if (n <= 0 || n % 16 != 0) return QEFT_ERR_SHAPE;
    if (k <= 0 || group_size <= 0 || k % group_size != 0) return QEFT_ERR_GROUP;
    if (!qeft::gemv_v3_ok(k, group_size, n_out)) return QEFT_ERR_SHAPE;
    G.K = k;
    G.n_out = n_out;
    G.nsteps = k / 128;
    G.nfull = (k - n_out) / 128;
    G.ngroups = group_size == k ? 1 : k / 128;
    G.nsets = n / 16;
    return QEFT_OK;
}

int qeft_decode_linear_blocks(int n_rows) {
    if (n_rows <= 0 || n_rows % 16 != 0) return 0;
    return qeft::gemv_v3_blocks(n_rows / 16);
}

int qeft_decode_linear(const void* x, const void* qweight, const void* sz_packed, const void* oweight, const void* bias,
                       void* y, int n, int k, int group_size, int n_out, int mode, const void* residual,
                       const float* ssq_in, int n_ssq_in, float eps, const void* gamma_out, void* y_norm, float* ssq_out,
                       qeft_stream_t stream) {
    if (!x || !qweight || !sz_packed || !y || (n_out > 0 && !oweight)) return QEFT_ERR_NULL;
    if (!aligned16(x) || !aligned16(qweight) || !aligned16(sz_packed) || (n_out > 0 && !aligned16(oweight))) return QEFT_ERR_ALIGN;
    qeft::V3Args a{};
    if (int e = v3_geom(a.g, n, k, group_size, n_out, mode)) return e;
    if ((residual || gamma_out) && mode != qeft::V3_MODE_PLAIN) return QEFT_ERR_SHAPE;
    if (gamma_out && (!residual || !y_norm || !ssq_out)) return QEFT_ERR_NULL;
    if ((residual && !aligned16(residual)) || (gamma_out && !aligned16(gamma_out))) return QEFT_ERR_ALIGN;
    if (ssq_in && (n_ssq_in < 1 || n_ssq_in > qeft::V3_MAX_SSQ || !aligned16(ssq_in))) return QEFT_ERR_SHAPE;
    a.x = (const qeft::f16*)x;
    a.qw = (const uint8_t*)qweight;
    a.szp = (const uint8_t*)sz_packed;
    a.ow = (const uint8_t*)oweight;
    a.bias = (const qeft::f16*)bias;
    a.residual = (const float*)residual;
    a.y = residual ? nullptr : (qeft::f16*)y;
    a.y32 = residual ? (float*)y : nullptr;
    a.ssq_in = ssq_in;
    a.n_ssq_in = ssq_in ? n_ssq_in : 0;
    a.eps = eps;
    a.gamma_out = (const qeft::f16*)gamma_out;
    a.ynorm = (qeft::f16*)y_norm;
    a.ssq_out = ssq_out;
    return finish(qeft::gemv_v3_launch(a, mode, (hipStream_t)stream));
}

int qeft_decode_linear_w3(const void* x, const void* qweight3, const void* sz_packed, const void* oweight, const void* bias,
                       void* y, int n, int k, int group_size, int n_out, int mode, const void* residual,
                       const float* ssq_in, int n_ssq_in, float eps, const void* gamma_out, void* y_norm, float* ssq_out,
                       qeft_stream_t stream) {
    if (!x || !qweight3 || !sz_packed || !y || (n_out > 0 && !oweight)) return QEFT_ERR_NULL;
    if (!aligned16(x) || ((uintptr_t)qweight3 & 3) != 0 || !aligned16(sz_packed) || (n_out > 0 && !aligned16(oweight))) return QEFT_ERR_ALIGN;
    qeft::V3Args a{};
    if (int e = v3_geom(a.g, n, k, group_size, n_out, mode)) return e;
    if ((residual || gamma_out) && mode != qeft::V3_MODE_PLAIN) return QEFT_ERR_SHAPE;
    if (gamma_out && (!residual || !y_norm || !ssq_out)) return QEFT_ERR_NULL;
    if ((residual && !aligned16(residual)) || (gamma_out && !aligned16(gamma_out))) return QEFT_ERR_ALIGN;
    if (ssq_in && (n_ssq_in < 1 || n_ssq_in > qeft::V3_MAX_SSQ || !aligned16(ssq_in))) return QEFT_ERR_SHAPE;
    a.x = (const qeft::f16*)x;
    a.qw = (const uint8_t*)qweight3;
    a.bits = 3;
    a.szp = (const uint8_t*)sz_packed;
    a.ow = (const uint8_t*)oweight;
    a.bias = (const qeft::f16*)bias;
    a.residual = (const float*)residual;
    a.y = residual ? nullptr : (qeft::f16*)y;
    a.y32 = residual ? (float*)y : nullptr;
    a.ssq_in = ssq_in;
    a.n_ssq_in = ssq_in ? n_ssq_in : 0;
    a.eps = eps;
    a.gamma_out = (const qeft::f16*)gamma_out;
    a.ynorm = (qeft::f16*)y_norm;
    a.ssq_out = ssq_out;
    return finish(qeft::gemv_v3_launch(a, mode, (hipStream_t)stream));
}

int qeft_decode_linear_hnorm(const void* h32, const void* gamma_x, const void* qweight, const void* sz_packed, const void* oweight,
                             const void* bias, void* y, int n, int k, int group_size, int n_out, int mode, float eps,
                             qeft_stream_t stream) {
    if (!h32 || !gamma_x || !qweight || !sz_packed || !y || (n_out > 0 && !oweight)) return QEFT_ERR_NULL;
    if (!aligned16(h32) || !aligned16(gamma_x) || !aligned16(qweight) || !aligned16(sz_packed) || (n_out > 0 && !aligned16(oweight)))
        return QEFT_ERR_ALIGN;
    qeft::V3Args a{};
    if (int e = v3_geom(a.g, n, k, group_size, n_out, mode)) return e;
    a.x = (const qeft::f16*)h32;              // fp32 [k]: the kernel reads it as such when xn_gamma is set
    a.xn_gamma = (const qeft::f16*)gamma_x;
    a.qw = (const uint8_t*)qweight;
    a.szp = (const uint8_t*)sz_packed;
    a.ow = (const uint8_t*)oweight;
    a.bias = (const qeft::f16*)bias;
    a.y = (qeft::f16*)y;
    a.eps = eps;
    return finish(qeft::gemv_v3_launch(a, mode, (hipStream_t)stream));
}

long long qeft_gemv_v3_check_extents(int n, int k, int group_size, int n_out, int n_ssq_in, int shrink_rows) {
    qeft::V3Geom G{};
    if (v3_geom(G, n, k, group_size, n_out, qeft::V3_MODE_PLAIN) != QEFT_OK) return -1;
    if (n_ssq_in < 0 || n_ssq_in > qeft::V3_MAX_SSQ) return -1;
    return qeft::gemv_v3_count_out_of_range(G, n - shrink_rows, n_ssq_in, n_ssq_in == 0, 4) +
           (G.nfull > 0 ? qeft::gemv_v3_count_out_of_range(G, n - shrink_rows, n_ssq_in, false, 3) : 0);
}

long long qeft_gemv_v3_check_extents_ckpt(int n, int k, int group_size, int n_out, int m, int gather, int shrink_rows) {
    qeft::V3Geom G{};
    if (v3_geom(G, n, k, group_size, n_out, qeft::V3_MODE_PLAIN) != QEFT_OK || m < 1 || m > qeft::V3_MAX_M) return -1;
    return qeft::gemv_v3_count_out_of_range_ckpt(G, n - shrink_rows, m, gather != 0);
}

int qeft_token_begin_norm_blocks(int hidden) { return hidden >= 8 ? qeft::token_begin_norm_blocks(hidden) : 0; }

int qeft_token_begin_norm(const void* embed, const void* tok, const void* rope_tab, const int* pos, void* h, void* rope_row,
                          const void* gamma, void* h_norm, float* ssq_out, int hidden, int vocab, int max_seq,
                          qeft_stream_t stream) {
    if (hidden < 8 || hidden % 8 != 0 || vocab < 1 || max_seq < 1) return QEFT_ERR_SHAPE;
    if (!embed || !tok || !h || !gamma || !h_norm || !ssq_out || (rope_row && (!rope_tab || !pos))) return QEFT_ERR_NULL;
    if (!aligned16(embed) || !aligned16(h) || !aligned16(gamma) || !aligned16(h_norm)) return QEFT_ERR_ALIGN;
    return finish(qeft::token_begin_norm_launch(embed, tok, rope_tab, pos, h, rope_row, gamma, h_norm, ssq_out, hidden, vocab,
                                                max_seq, (hipStream_t)stream));
}

int qeft_residual_norm(const void* h, const void* add, const void* gamma, void* h_out, void* h_norm, float* ssq_out,
                       int hidden, qeft_stream_t stream) {
    if (hidden < 8 || hidden % 8 != 0) return QEFT_ERR_SHAPE;
    if (!h || !h_out) return QEFT_ERR_NULL;
    if (gamma && (!h_norm || !ssq_out)) return QEFT_ERR_NULL;
    if (!aligned16(h) || !aligned16(h_out) || (add && !aligned16(add)) || (gamma && (!aligned16(gamma) || !aligned16(h_norm))))
        return QEFT_ERR_ALIGN;
    return finish(qeft::residual_norm_launch(h, add, gamma, h_out, h_norm, ssq_out, hidden, (hipStream_t)stream));
}

int qeft_token_begin(const void* embed, const void* tok, const void* rope_tab, const int* pos, void* h, void* rope_row,
                     int hidden, int vocab, int max_seq, qeft_stream_t stream) {
    if (hidden < 8 || hidden % 8 != 0 || vocab < 1 || max_seq < 1) return QEFT_ERR_SHAPE;
    if (!embed || !tok || !h || (rope_row && (!rope_tab || !pos))) return QEFT_ERR_NULL;
    if (!aligned16(embed) || !aligned16(h)) return QEFT_ERR_ALIGN;
    return finish(qeft::token_begin_launch(embed, tok, rope_tab, pos, h, rope_row, hidden, vocab, max_seq, (hipStream_t)stream));
}

int qeft_token_end(const void* logits, void* tok, int* pos, int vocab, int greedy, qeft_stream_t stream) {
    if (vocab < 1) return QEFT_ERR_SHAPE;
    if (!pos || (greedy && (!logits || !tok))) return QEFT_ERR_NULL;
    if (greedy && !aligned16(logits)) return QEFT_ERR_ALIGN;
    return finish(qeft::token_end_launch(logits, tok, pos, vocab, greedy, (hipStream_t)stream));
}

}  // extern "C"
