// extern "C" entry points declared in include/qeft_hip.h: argument validation + dispatch.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/qeft_hip.h"
#include "qeft_common.h"

namespace qeft {
hipError_t gemv_w4_dispatch(const GemvArgs& a, int m, hipStream_t st);
hipError_t dequant_w4_launch(const void* qw, const void* scales, const void* zeros, const void* ow, void* out, int N,
                             int K, int G, int n_out, hipStream_t st);
hipError_t pack_oweight_launch(const void* ow, void* il, int N, int R, hipStream_t st);
hipError_t gemm_w4_launch(const void* x, const void* qw, const void* scales, const void* zeros, const void* ow,
                          const void* bias, void* y, int M, int N, int K, int G, int n_out, hipStream_t st);
hipError_t gemm_w4_dx_launch(const void* dy, const void* qw, const void* scales, const void* zeros, const void* ow,
                             void* dx, int M, int N, int K, int G, int n_out, hipStream_t st);
hipError_t grad_oweight_launch(const void* dy, const void* x, void* dow, int M, int N, int K, int n_out,
                               hipStream_t st);
}  // namespace qeft

static thread_local int g_last_hip_error = 0;

static int finish(hipError_t e) {
    if (e == hipSuccess) return QEFT_OK;
    g_last_hip_error = (int)e;
    return QEFT_ERR_LAUNCH;
}

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

int qeft_gemv_w4_fused(const void* x, const void* qweight, const void* scales, const void* scaled_zeros,
                       const void* oweight_il, const void* bias, const int* reorder_ids, const void* residual,
                       void* y, int m, int n, int k, int group_size, int n_out, qeft_stream_t stream) {
    if (m < 1 || m > 7) return QEFT_ERR_BATCH;
    if (int e = check_common(n, k, group_size, n_out)) return e;
    if (!x || !qweight || !scales || !scaled_zeros || !y || (n_out > 0 && !oweight_il)) return QEFT_ERR_NULL;
    if (!aligned16(x) || !aligned16(qweight) || (n_out > 0 && !aligned16(oweight_il))) return QEFT_ERR_ALIGN;
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
    if (group_size != k && (group_size & (group_size - 1)) != 0) return QEFT_ERR_GROUP;  // GEMV: power of two or == K
    a.gshift = (group_size == k) ? 31 : __builtin_ctz(group_size);
    return finish(qeft::gemv_w4_dispatch(a, m, (hipStream_t)stream));
}

int qeft_gemv_w4(const void* x, const void* qweight, const void* scales, const void* scaled_zeros, void* y, int m,
                 int n, int k, int group_size, qeft_stream_t stream) {
    return qeft_gemv_w4_fused(x, qweight, scales, scaled_zeros, nullptr, nullptr, nullptr, nullptr, y, m, n, k,
                              group_size, 0, stream);
}

int qeft_gemv_w4_qeft(const void* x, const void* qweight, const void* scales, const void* scaled_zeros,
                      const void* oweight_il, void* y, int m, int n, int k, int group_size, int n_out,
                      qeft_stream_t stream) {
    return qeft_gemv_w4_fused(x, qweight, scales, scaled_zeros, oweight_il, nullptr, nullptr, nullptr, y, m, n, k,
                              group_size, n_out, stream);
}

int qeft_gemm_w4(const void* x, const void* qweight, const void* scales, const void* scaled_zeros,
                 const void* oweight, const void* bias, void* y, int m, int n, int k, int group_size, int n_out,
                 qeft_stream_t stream) {
    if (m < 1) return QEFT_ERR_SHAPE;
    if (!oweight) n_out = 0;
    if (int e = check_common(n, k, group_size, n_out)) return e;
    if (!x || !qweight || !scales || !scaled_zeros || !y) return QEFT_ERR_NULL;
    if (!aligned16(x) || !aligned16(qweight) || (n_out > 0 && !aligned16(oweight))) return QEFT_ERR_ALIGN;
    return finish(qeft::gemm_w4_launch(x, qweight, scales, scaled_zeros, oweight, bias, y, m, n, k, group_size, n_out,
                                       (hipStream_t)stream));
}

int qeft_gemm_w4_dx(const void* dy, const void* qweight, const void* scales, const void* scaled_zeros,
                    const void* oweight, void* dx, int m, int n, int k, int group_size, int n_out,
                    qeft_stream_t stream) {
    if (m < 1) return QEFT_ERR_SHAPE;
    if (!oweight) n_out = 0;
    if (int e = check_common(n, k, group_size, n_out)) return e;
    if (!dy || !qweight || !scales || !scaled_zeros || !dx) return QEFT_ERR_NULL;
    if (!aligned16(dy) || !aligned16(qweight) || !aligned16(dx) || (n_out > 0 && !aligned16(oweight)))
        return QEFT_ERR_ALIGN;
    return finish(qeft::gemm_w4_dx_launch(dy, qweight, scales, scaled_zeros, oweight, dx, m, n, k, group_size, n_out,
                                          (hipStream_t)stream));
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

int qeft_pack_oweight(const void* oweight, void* oweight_il, int n, int n_out, qeft_stream_t stream) {
    if (n <= 0 || n % 8 != 0 || n_out <= 0 || n_out % 32 != 0) return QEFT_ERR_SHAPE;
    if (!oweight || !oweight_il) return QEFT_ERR_NULL;
    return finish(qeft::pack_oweight_launch(oweight, oweight_il, n, n_out, (hipStream_t)stream));
}

}  // extern "C"
