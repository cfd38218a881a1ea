/*
 * qeft_hip.h — C ABI of the MI355X (gfx950) packed-weight quantized linear.
 *
 * Drop-in boundary for the one hot path of xvyaward/qeft: the functions the
 * reference binds in its `qeft_cuda` torch extension (qeft/kernel/qeft_cuda.cpp:10-27)
 * for the W4 group-quantised linear with retained fp16 outlier columns.
 * Plain pointers and sizes only; every pointer is a DEVICE pointer unless
 * noted; `stream` is a hipStream_t passed as void*.  No allocation, no
 * synchronisation and no host<->device copy happens inside any entry point,
 * so every call may be captured into a hipGraph.
 *
 * Buffers (reference checkpoint layout, qeft/qlinear.py:143-173):
 *   qweight       int16 [N/4, K]      4 rows interleaved per 64-k tile, nibble order of
 *                                     pack_intweight (qlinear.py:81-121)
 *   scales        fp16  [K/g, N]
 *   scaled_zeros  fp16  [K/g, N]      = -(zero * scale)
 *   oweight       fp16  [N, r]        retained outlier columns (the LAST r input columns)
 *   oweight_il    fp16  [N/2, 2r]     pack_oweight interleave (qlinear.py:70-79)
 *   x             fp16  [m, K] row-major contiguous;  y fp16 [m, N]
 *
 * Every function returns QEFT_OK (0) or a QEFT_ERR_* code; qeft_error_string()
 * gives the message.  Kernel-launch failures return QEFT_ERR_LAUNCH and the
 * hipError_t is available from qeft_last_hip_error().
 */
#ifndef QEFT_HIP_H
#define QEFT_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define QEFT_OK 0
#define QEFT_ERR_BATCH 1  /* m outside 1..7 on a gemv entry: "Unsupported batch size for gemv kernel."
                             (gemv_cuda_qeft.cu:466, gemv_cuda.cu:466) */
#define QEFT_ERR_SHAPE 2  /* N % 4, K % 64, n_out % 32 ... */
#define QEFT_ERR_GROUP 3  /* group size not a multiple of 32 dividing K */
#define QEFT_ERR_NULL 4   /* required pointer is NULL */
#define QEFT_ERR_LAUNCH 5 /* hip launch error, see qeft_last_hip_error() */
#define QEFT_ERR_ALIGN 6  /* pointer not 16-byte aligned */

typedef void* qeft_stream_t;

int qeft_abi_version(void);
const char* qeft_error_string(int code);
int qeft_last_hip_error(void);
/* Name of the kernel variant the calling thread's most recent compute entry point launched, e.g. "gemv_mfma",
 * "gemm_v2_128x256", "gemm_v2_128x128+splitk", "gemv_smallm", "dx128", "dx64+split", "grad_oweight_mfma",
 * "grad_oweight_mfma_n64".  Static storage;
 * diagnostic only (the parity tests assert it so that every routing tier keeps its coverage). */
const char* qeft_last_variant(void);

/* Kernel reached by the three gemv entries below (qeft_last_variant() names it):
 *   "gemv_v3" (m = 1) / "gemv_v3_mb" (m = 2..7): the round-2 MFMA GEMV of csrc/gemv_v3.h, taken when n % 16 == 0, k % 128 == 0,
 *       group_size in {128, k}, n_out in {0, 128}, no fp16 `residual`, operands 16-byte aligned.  It consumes the operands as the
 *       checkpoint holds them: scales / scaled_zeros [k/g][n] (or the sz_packed shadow when given), oweight_interleaved,
 *       reorder_ids gathered inside the launch.  Batches whose x rows exceed the block's LDS run as several launches.
 *   "gemv_mfma" / "gemv_valu": the round-1 kernels, for every other accepted shape (group sizes 32 / 64 / 256, n_out 32 / 64 /
 *       96, n % 16 != 0, k % 128 != 0, a residual).  QEFT_GEMV_V3=0 in the environment forces them everywhere.
 *
 * Decode GEMV, m in 1..7, no outlier slice.
 * Replaces gemv_4bit(in_feats, kernel, scaling_factors, zeros, m, n, k, group_size)
 * (qeft/kernel/quantization_new/gemv/gemv_cuda.cu:358-525). */
int qeft_gemv_w4(const void* x, const void* qweight, const void* scales, const void* scaled_zeros,
                 void* y, int m, int n, int k, int group_size, qeft_stream_t stream);

/* Decode GEMV with the fp16 outlier slice taken from the interleaved buffer.
 * Replaces gemv_4bit_qeft(in_feats, kernel, scaling_factors, zeros, oweight_interleaved, m, n, k, group_size)
 * (qeft/kernel/quantization_new/gemv/gemv_cuda_qeft.cu:392-513); n_out = oweight_il.size(1)/2 (:424). */
int qeft_gemv_w4_qeft(const void* x, const void* qweight, const void* scales, const void* scaled_zeros,
                      const void* oweight_il, void* y, int m, int n, int k, int group_size, int n_out,
                      qeft_stream_t stream);

/* Fused decode GEMV: everything QuantLinear.forward_* does around the kernel in one launch
 * (qeft/qlinear.py:244-330): optional input gather x[:, reorder_ids] (:275, int32 ids, NULL = none),
 * optional bias add (:268, NULL = none), optional residual add into y (y = acc + residual; NULL = none).
 * oweight_il may be NULL when n_out == 0; sz_packed (see qeft_pack_scales) may be NULL. */
int qeft_gemv_w4_fused(const void* x, const void* qweight, const void* scales, const void* scaled_zeros,
                       const void* oweight_il, const void* bias, const int* reorder_ids,
                       const void* residual, const void* sz_packed, void* y, int m, int n, int k, int group_size,
                       int n_out, qeft_stream_t stream);

/* Optional derived buffer for the decode GEMV (never part of the checkpoint): sz_packed int32 [N/16][K/g][16] with
 * scale | scaled_zero << 16 per (row, group), so the scales of a 16-row block are one contiguous run.  Built once at
 * load time (QuantLinear.set_kernel); entries that take `sz_packed` accept NULL and then read scales / scaled_zeros
 * in their checkpoint layout.  group_size must be 128 or k; n % 16 == 0. */
int qeft_pack_scales(const void* scales, const void* scaled_zeros, void* sz_packed, int n, int k, int group_size,
                     qeft_stream_t stream);

/* Prefill / fine-tune GEMM  y[M,N] = x[M,K] . Wdeq[N,K]^T  on MFMA, fp32 accumulate, fp16 out.
 * Replaces gemm_4bit(in_feats, kernel, scales, zeros) (qeft/kernel/quantization_new/gemm/gemm_cuda.cu:929-1033).
 * With oweight == NULL the INT4 nibbles are used for every column (exactly gemm_4bit);
 * with oweight != NULL the last n_out columns come from the fp16 slice inside the same launch — the fusion
 * gemm_cuda_qeft.cu:1003 intended and qlinear.py:266 does with a second F.linear.  bias may be NULL. */
int qeft_gemm_w4(const void* x, const void* qweight, const void* scales, const void* scaled_zeros,
                 const void* oweight, const void* bias, void* y, int m, int n, int k, int group_size,
                 int n_out, qeft_stream_t stream);

/* qeft_gemm_w4 with the MLP's activation in its epilogue:  y = silu(gate) * (x . Wdeq^T + bias), gate fp16 [m, n] -- the
 * reference's up_proj GEMM followed by act_fn(gate) * up (modeling_llama's LlamaMLP around qlinear.py:244-271) in one
 * launch on the 256-row tier (variant "gemm_v3_256x128+silu"), the GEMM plus qeft_silu_mul in place on the others; the
 * same rounding either way (the product is rounded to fp16 before the activation).  n % 8 == 0; y may not alias gate. */
int qeft_gemm_w4_silu_mul(const void* x, const void* qweight, const void* scales, const void* scaled_zeros,
                          const void* oweight, const void* bias, const void* gate, void* y, int m, int n, int k,
                          int group_size, int n_out, qeft_stream_t stream);

/* gate_proj and up_proj of the MLP as ONE launch: qweight / scales / scaled_zeros / oweight / bias hold the two linears'
 * rows interleaved in blocks of 64 (rows 128 b .. 128 b + 63 = gate rows 64 b .., rows 128 b + 64 .. = up rows 64 b ..: a
 * permutation of whole checkpoint rows, built at load time by qeft_amd/fuse.py, never saved), n2 = both linears' rows;
 * y [m, n2 / 2] = silu(x . Wgate^T + bgate) * (x . Wup^T + bup), the two products rounded to fp16 first (bit-equal to the
 * separate launches).  qeft_gemm_w4_gateup_supported() says whether a shape is taken (n2 % 128 == 0, K >= 384,
 * group a power of two >= 64, n_out % 64 == 0); otherwise use the two linears and qeft_gemm_w4_silu_mul. */
int qeft_gemm_w4_gateup_supported(int m, int n2, int k, int group_size, int n_out);
int qeft_gemm_w4_gateup(const void* x, const void* qweight, const void* scales, const void* scaled_zeros,
                        const void* oweight, const void* bias, void* y, int m, int n2, int k, int group_size, int n_out,
                        qeft_stream_t stream);

/* The same with a scratch buffer for mid-size m: when the 128x128 tiling of [m, n] leaves most of the chip idle the
 * K loop is cut into S parts (fp32 partial tiles in `workspace`, summed in a fixed order by a second small launch --
 * deterministic).  qeft_gemm_w4_workspace_bytes() is the size that enables it for a shape (0: no split would be
 * used); a smaller or NULL workspace silently means fewer parts / none.  16-byte aligned device memory, contents
 * irrelevant, must not be shared by launches that may overlap. */
long long qeft_gemm_w4_workspace_bytes(int m, int n, int k, int n_out);
int qeft_gemm_w4_ws(const void* x, const void* qweight, const void* scales, const void* scaled_zeros,
                    const void* oweight, const void* bias, void* y, void* workspace, long long workspace_bytes, int m,
                    int n, int k, int group_size, int n_out, qeft_stream_t stream);

/* Backward wrt the input: dx[M,K] = dy[M,N] . Wdeq[N,K]; columns K-n_out.. use oweight (SURVEY.md §8a row 7;
 * the mathematically correct form of QuantMatMulQEFT.backward, qlinear.py:30-44). */
int qeft_gemm_w4_dx(const void* dy, const void* qweight, const void* scales, const void* scaled_zeros,
                    const void* oweight, void* dx, int m, int n, int k, int group_size, int n_out,
                    qeft_stream_t stream);
/* The same with a scratch buffer (as qeft_gemm_w4_ws): few output tiles and a long contraction over n -> the n loop is
 * cut into parts, fp32 partials in `workspace`, summed in a fixed order. */
long long qeft_gemm_w4_dx_workspace_bytes(int m, int n, int k);
int qeft_gemm_w4_dx_ws(const void* dy, const void* qweight, const void* scales, const void* scaled_zeros,
                       const void* oweight, void* dx, void* workspace, long long workspace_bytes, int m, int n, int k,
                       int group_size, int n_out, qeft_stream_t stream);

/* Gradient of the trainable outlier slice: d_oweight[N, r] (fp32) = dy[M,N]^T . x[M, K-r:]  (qlinear.py:41-42). */
int qeft_grad_oweight(const void* dy, const void* x, void* d_oweight_f32, int m, int n, int k, int n_out,
                      qeft_stream_t stream);

/* Dense dequantisation Wdeq[N,K] fp16 (validation / backward helper; the role of the uncompiled
 * dequantize_weight_4bit_qeft, qeft/kernel/quantization_new/dequantize/dequantize_cuda_qeft.cu:38-118).
 * oweight (plain [N, r]) may be NULL. */
int qeft_dequant_w4(const void* qweight, const void* scales, const void* scaled_zeros, const void* oweight,
                    void* w_out, int n, int k, int group_size, int n_out, qeft_stream_t stream);

/* Re-derive the interleaved outlier buffer from oweight on device (pack_oweight, qlinear.py:70-79);
 * used after fine-tuning so the GEMV never reads a stale copy (modelutils.py:192 quirk). */
int qeft_pack_oweight(const void* oweight, void* oweight_il, int n, int n_out, qeft_stream_t stream);

/* ---- decode-step helpers (SURVEY.md section 8f "next" rows 1, 3, 4; not part of the packed-weight path) ---- */

/* Up to 3 quantized linears that read the SAME x (q/k/v or gate/up) in one launch, batch 1.
 * Host arrays of `nparts` device pointers; oweight_il / bias entries may be NULL (all-or-none for oweight_il).
 * Equivalent to nparts calls of gemv_4bit_qeft (gemv_cuda_qeft.cu:392-513) on identical in_feats.
 * norm_gamma != NULL: the input is RMS-normalised while it is staged (x * rsqrt(mean x^2 + norm_eps) * gamma, the
 * arithmetic of qeft_rmsnorm / layernorm.cu:26-76), i.e. the decoder's input_layernorm / post_attention_layernorm
 * fused into the projection that consumes it (K <= 16384). */
int qeft_gemv_w4_group(const void* x, const void* norm_gamma, float norm_eps, int nparts,
                       const void* const* qweight, const void* const* scales, const void* const* scaled_zeros,
                       const void* const* oweight_il, const void* const* bias, const void* const* sz_packed,
                       void* const* y, const int* n, int k, int group_size, int n_out, qeft_stream_t stream);

/* down_proj of the decode step in one launch: y = W . (silu(gate) * up) (+ bias) (+ residual), batch 1.
 * The activation is formed while x is staged, rounded to fp16 exactly like qeft_silu_mul (K <= 16384). */
int qeft_gemv_w4_silu(const void* gate, const void* up, const void* qweight, const void* scales,
                      const void* scaled_zeros, const void* oweight_il, const void* bias, const void* residual,
                      const void* sz_packed, void* y, int n, int k, int group_size, int n_out, qeft_stream_t stream);

/* y = rmsnorm(x (+ add)) * gamma, fp32 statistics (role of layernorm_forward_cuda, qeft/kernel/layernorm/layernorm.cu:26-76).
 * If add != NULL the sum x + add is normalised and, if res_out != NULL, also written there (fused residual). */
int qeft_rmsnorm(const void* x, const void* add, const void* gamma, void* res_out, void* y, int m, int hidden,
                 float eps, qeft_stream_t stream);

/* out = silu(gate) * up, n elements (n % 8 == 0). */
int qeft_silu_mul(const void* gate, const void* up, void* out, int n, qeft_stream_t stream);

/* One decode token of one sequence: rotary on q/k at position *pos (device int), append k/v to the caches
 * [n_kv][max_seq][128] and compute softmax(q.K^T/sqrt(128)).V (role of single_query_attention,
 * qeft/kernel/attention/ft_attention.cpp:110-181, neox rotary).  head_dim is 128, max_seq % 16 == 0.
 * cos/sin: fp32 [tab_rows][64], tab_rows >= max_seq (row *pos is used), or tab_rows == 1: the caller has already
 * selected the row of this position (the kernel then has no load that waits for *pos before the rotary).
 * out_pos (optional int32 [n_heads*128]): element i of the attention output is stored at out[out_pos[i]].  With
 * out_pos = inverse of o_proj's reorder_ids the o_proj input gather (qlinear.py:275) costs nothing.
 * n_split in {1, 2, 4, 8}: blocks per head (the context is dealt over them; the last block to finish merges the
 * head, in a fixed order).  n_split > 1 needs `workspace`: qeft_attn_workspace_bytes(n_heads, n_split) bytes of
 * device memory, 16-byte aligned, ZERO before the first call and not shared by launches that may overlap; the
 * kernel leaves it ready for the next call. */
int qeft_attn_workspace_bytes(int n_heads, int n_split);   /* 0 for n_split == 1 or bad arguments */
int qeft_rope_attn_decode(const void* q, const void* k, const void* v, const void* cos_tab, const void* sin_tab,
                          int tab_rows, void* k_cache, void* v_cache, const int* pos, const int* out_pos, void* out,
                          void* workspace, int n_split, int n_heads, int n_kv_heads, int max_seq, qeft_stream_t stream);

/* The same step on the REFERENCE's cache layout, for the `qeft_cuda.single_query_attention` boundary
 * (qeft/kernel/qeft_cuda.cpp:23-26, ft_attention.cpp:110-181; caller ftllama_modeling.py:139-153):
 *   k_cache_ft  fp16 [n_kv][128/8][max_seq][8]   (FasterTransformer key layout, ft_attention.cpp:131-133)
 *   v_cache     fp16 [n_kv][max_seq][128]
 * one sequence, head_dim 128, neox rotary from the cos/sin table (tab_rows as above), one block per head, natural output
 * order.  The Python shim loops over the batch and raises for ALiBi / other head sizes / interleaved rotary. */
int qeft_single_query_attention(const void* q, const void* k, const void* v, const void* cos_tab, const void* sin_tab,
                                int tab_rows, void* k_cache_ft, void* v_cache, const int* pos, void* out, int n_heads,
                                int n_kv_heads, int max_seq, qeft_stream_t stream);
/* qeft_single_query_attention_alibi: the same with the reference's alibi_slopes_ (fp32 [n_heads], ft_attention.cpp:149-154): the
 * head's linear position bias slope * (key position - query position) is added to the scaled score
 * (decoder_masked_multihead_attention_template.hpp:1335-1345). */
int qeft_single_query_attention_alibi(const void* q, const void* k, const void* v, const void* cos_tab, const void* sin_tab,
                                      int tab_rows, void* k_cache_ft, void* v_cache, const int* pos, void* out, int n_heads,
                                      int n_kv_heads, int max_seq, const float* alibi_slopes, qeft_stream_t stream);

/* qeft_single_query_attention_generic (round 4): the same boundary for the other head sizes the reference instantiates
 * (ft_attention.cpp:110-181: 32 .. 256): any head_dim % 8 == 0 in [8, 256], caches k [n_kv][head_dim/8][max_seq][8],
 * v [n_kv][max_seq][head_dim], NO rotary inside (the shim rotates q / k first), optional alibi_slopes (NULL: none). */
int qeft_single_query_attention_generic(const void* q, const void* k, const void* v, void* k_cache_ft, void* v_cache, const int* pos,
                                        void* out, int n_heads, int n_kv_heads, int max_seq, int head_dim, const float* alibi_slopes,
                                        qeft_stream_t stream);

/* ---- v3 decode linear (batch 1): the decode engine's production GEMV (csrc/gemv_v3.h) ------------------------------
 *     y[n] = Wdeq . x (+ bias)                                          (gemv_4bit_qeft, gemv_cuda_qeft.cu:392-513, m = 1)
 * reads the fp16 vector x[k] AS IT IS (no transform of x inside); the decoder's element-wise neighbours sit in the EPILOGUE
 * of the producing launch instead of the prologue of the consuming one (every block stages x by LDS-DMA and starts its
 * weight stream at once):
 *   ssq_in != NULL    x is (v * gamma) of some vector v whose producer also stored n_ssq_in (<= 512) partial sums of v^2
 *                     (readable up to a multiple of 4 floats): y *= rsqrt(sum(ssq_in) / k + eps) -- together the RMSNorm of
 *                     layernorm.cu:26-76 in front of W
 *   mode 1 (PAIR)     the operand's rows are pair-interleaved -- rows [16g, 16g+8) are gate_proj rows [8g, 8g+8), rows
 *                     [16g+8, 16g+16) the same rows of up_proj (a derived buffer, see qeft_amd/fuse.py):
 *                     y[8g + i] = silu(fp16 gate_i) * fp16 up_i, n/2 outputs (the arithmetic of qeft_silu_mul)
 *   residual != NULL  (mode 0) the fp32 residual stream: `residual` and `y` are FLOAT vectors, y = Wx + residual (in place is fine)
 *   gamma_out != NULL (with residual) additionally y_norm = fp16(y * gamma_out) and ssq_out[b] = the sum of y^2 over the rows
 *                     of block b, b < qeft_decode_linear_blocks(n): what the NEXT launch takes as x / ssq_in.
 * Operands: qweight int16 [n/4, k]; sz_packed from qeft_pack_scales (required); oweight PLAIN fp16 [n, 128] (the `oweight`
 * buffer of the checkpoint, not the interleaved copy; NULL when n_out == 0); bias optional.  Several linears that read the
 * same x (q|k|v) are ONE operand: their buffers concatenated along n at load time.
 * Requirements: k % 128 == 0, n_out in {0, 128}, group 128 or k, n % 16 == 0.
 * qeft_gemv_v3_check_extents: CPU-only self-check -- enumerates every address the kernel would form for the configuration
 * and returns how many leave their operand (0 = none; -1 = configuration not accepted); shrink_rows > 0 is the negative
 * control (operands that many rows shorter than the geometry: the count must be positive). */
int qeft_decode_linear_blocks(int n_rows);
int qeft_decode_linear(const void* x, const void* qweight, const void* sz_packed, const void* oweight, const void* bias,
                       void* y, int n, int k, int group_size, int n_out, int mode, const void* residual,
                       const float* ssq_in, int n_ssq_in, float eps, const void* gamma_out, void* y_norm, float* ssq_out,
                       qeft_stream_t stream);
/* qeft_decode_linear_w3: qeft_decode_linear on the 3-bit EXTENSION layout (qweight3 int32 [n/16, ((k - n_out)/128) * 192],
 * see below): 12 bytes per lane and step, every 3-bit field shifted down to bits 0..2 in registers (x stays raw). */
int qeft_decode_linear_w3(const void* x, const void* qweight3, const void* sz_packed, const void* oweight, const void* bias,
                          void* y, int n, int k, int group_size, int n_out, int mode, const void* residual,
                          const float* ssq_in, int n_ssq_in, float eps, const void* gamma_out, void* y_norm, float* ssq_out,
                          qeft_stream_t stream);
/* qeft_decode_linear_hnorm: the same launch with the WHOLE RMSNorm on the consumer -- x is the fp32 vector h [k]; the launch
 * stages fp16(h * gamma_x) (the producers' rounding) and multiplies its row sums by rsqrt(mean(h^2) + eps).  For inputs that
 * no GEMV epilogue produced: the tensor-parallel path's h comes out of an all-reduce (qeft_amd/llama.py).  y fp16 (mode 0: [n],
 * mode 1: [n/2]); no residual / gamma_out forms. */
int qeft_decode_linear_hnorm(const void* h32, const void* gamma_x, const void* qweight, const void* sz_packed, const void* oweight,
                             const void* bias, void* y, int n, int k, int group_size, int n_out, int mode, float eps,
                             qeft_stream_t stream);
long long qeft_gemv_v3_check_extents(int n, int k, int group_size, int n_out, int n_ssq_in, int shrink_rows);
/* The same self-check for the launches the reference's gemv entries make on the CHECKPOINT-layout operands (qeft_gemv_w4,
 * qeft_gemv_w4_qeft, qeft_gemv_w4_fused): scales / scaled_zeros fp16 [k/g][n] staged raw, oweight_interleaved, m = 1..7 batch
 * rows, optional reorder_ids gather; also checks every LDS destination of the staging against the block's LDS carve-up. */
long long qeft_gemv_v3_check_extents_ckpt(int n, int k, int group_size, int n_out, int m, int gather, int shrink_rows);

/* Producer-form helpers for the same scheme.
 * qeft_token_begin_norm: qeft_token_begin + h32 = float(embed[*tok]), h_norm = fp16(embed[*tok] * gamma),
 *                        ssq_out[b] (b < qeft_token_begin_norm_blocks(hidden)) = partial sums of squares.
 * qeft_residual_norm:    h_out(fp32) = h(fp32) (+ add(fp16)); with gamma: h_norm = fp16(h_out * gamma), ssq_out[b] likewise
 *                        (tensor-parallel path: the residual add follows an all-reduce; also the stand-alone reference form
 *                        the fused epilogues are tested against). */
int qeft_token_begin_norm_blocks(int hidden);
int qeft_token_begin_norm(const void* embed, const void* tok, const void* rope_tab, const int* pos, void* h32, void* rope_row,
                          const void* gamma, void* h_norm, float* ssq_out, int hidden, int vocab, int max_seq,
                          qeft_stream_t stream);
int qeft_rmsnorm_f32(const void* x32, const void* gamma, void* y, int m, int hidden, float eps, qeft_stream_t stream); /* qeft_rmsnorm on an fp32 input */
/* qeft_rope_rows (prefill helper): NeoX rotary of the first n_heads heads of every row of x (fp16, rows row_stride elements
 * apart, a head = 128 consecutive elements; row_stride = n_heads * 128 for a plain [t][n_heads][128] tensor, the width of the
 * fused q|k|v output for its q and k heads) in place, cos_tab / sin_tab fp32 [t][64]. */
int qeft_rope_rows(void* x, const void* cos_tab, const void* sin_tab, int t, int n_heads, int row_stride,
                   qeft_stream_t stream);
/* qeft_lm_head_f16 (decode harness, token tail): logits[vocab] (fp16) = weight[vocab][hidden] (fp16, the unquantized lm_head)
 * . fp16(rmsnorm(h32) * gamma) -- the final norm of qeft_rmsnorm_f32 and the head GEMV in one launch, fp32 accumulation.
 * hidden in {512, 1024, 2048, 4096, 5120, 8192}. */
int qeft_lm_head_f16(const void* h32, const void* gamma, const void* weight, void* logits, int hidden, int vocab, float eps,
                     qeft_stream_t stream);
int qeft_residual_norm(const void* h32, const void* add, const void* gamma, void* h32_out, void* h_norm, float* ssq_out,
                       int hidden, qeft_stream_t stream);

/* GEMM forward / dX of a 3-bit layer ON the 3-bit stream (round 3): the loader-wave tiers (256 x 128 tiles from 224 tiles and
 * M >= 1024, the forward also 128 x 128 tiles from 112 tiles and M > 128) stage / load the 12-byte lane records and unpack them in
 * registers -- no 3 -> 4-bit expansion pass.  qeft_gemm_w3[_dx]_supported says whether a shape is taken (k % 128 == 0,
 * n_out % 128 == 0, group a power of two >= 64 (dX: a multiple of 128), n % 16 == 0 (dX: n % 64 == 0, n >= 256)); other shapes
 * go through qeft_expand_w3 and the 4-bit entries.  Arguments as qeft_gemm_w4 / qeft_gemm_w4_dx with qweight3 int32
 * [n/16, ((k - n_out)/128) * 192]. */
int qeft_gemm_w3_supported(int m, int n, int k, int group_size, int n_out);
int qeft_gemm_w3(const void* x, const void* qweight3, const void* scales, const void* scaled_zeros, const void* oweight,
                 const void* bias, void* y, int m, int n, int k, int group_size, int n_out, qeft_stream_t stream);
int qeft_gemm_w3_dx_supported(int m, int n, int k, int group_size, int n_out);
int qeft_gemm_w3_dx(const void* dy, const void* qweight3, const void* scales, const void* scaled_zeros, const void* oweight,
                    void* dx, int m, int n, int k, int group_size, int n_out, qeft_stream_t stream);

/* ---- 3-bit EXTENSION (BASELINE config 5).  The reference cannot pack or run 3 bits (QuantLinear asserts
 * bits == 4, qlinear.py:127; its quantiser can produce them, quant.py:8-10 with maxq = 7), so the layout is this
 * library's own, shaped by the decode GEMV (oracle/qeft_oracle.py: pack_w3 / w3_position):
 *   qweight3  int32 [N/16, ((K - n_out)/128) * 192]   12 bytes per (row, 32-k chunk), 16 rows x 128 k contiguous;
 *                                                      the fp16 outlier columns have no 3-bit fields
 * scales / scaled_zeros / oweight / oweight_il / sz_packed are as for 4 bits.
 * Requirements: N % 16 == 0, K % 128 == 0, n_out % 128 == 0, n_out < K, group size 128 or K.
 * qeft_gemv_w3        y[m,N] = x[m,K] . W^T (+ bias, + residual), any m >= 1 (16 rows per weight pass)
 * qeft_gemv_w3_group / qeft_gemv_w3_silu   as their w4 namesakes
 * qeft_expand_w3      rewrites the stream into the 4-bit checkpoint layout int16 [N/4, K] (dead zero nibbles under
 *                     the fp16 columns) so that qeft_gemm_w4 / qeft_gemm_w4_dx / qeft_dequant_w4 serve 3-bit layers:
 *                     reads 3/8 byte and writes 1/2 byte per weight. */
int qeft_gemv_w3(const void* x, const void* qweight3, const void* scales, const void* scaled_zeros,
                 const void* oweight_il, const void* bias, const void* residual, const void* sz_packed, void* y, int m,
                 int n, int k, int group_size, int n_out, qeft_stream_t stream);
int qeft_gemv_w3_group(const void* x, const void* norm_gamma, float norm_eps, int nparts,
                       const void* const* qweight3, const void* const* scales, const void* const* scaled_zeros,
                       const void* const* oweight_il, const void* const* bias, const void* const* sz_packed,
                       void* const* y, const int* n, int k, int group_size, int n_out, qeft_stream_t stream);
int qeft_gemv_w3_silu(const void* gate, const void* up, const void* qweight3, const void* scales,
                      const void* scaled_zeros, const void* oweight_il, const void* bias, const void* residual,
                      const void* sz_packed, void* y, int n, int k, int group_size, int n_out, qeft_stream_t stream);
int qeft_expand_w3(const void* qweight3, void* qweight4, int n, int k, int n_out, qeft_stream_t stream);

/* Token boundary of the decode loop (main.py:340-371, benchmark.py:293-338).
 * begin: h[hidden] = embed[*tok] (tok: device int64, clamped to the vocabulary) and, if rope_row != NULL,
 *        rope_row[128] = rope_tab[*pos] (rope_tab fp32 [max_seq][cos 64 | sin 64]) for qeft_rope_attn_decode(tab_rows = 1).
 * end:   if greedy: *tok = argmax(logits[vocab]) (lowest index among equal maxima); always *pos += 1. */
int qeft_token_begin(const void* embed, const void* tok, const void* rope_tab, const int* pos, void* h, void* rope_row,
                     int hidden, int vocab, int max_seq, qeft_stream_t stream);
int qeft_token_end(const void* logits, void* tok, int* pos, int vocab, int greedy, qeft_stream_t stream);

/* One-shot all-reduce of the tensor-parallel decode path (csrc/oneshot.hip; no reference counterpart --
 * the reference places whole layers on GPUs, qeft/utils/modelutils.py:21-57).  In-place fp32 sum of t[n] over `world` ranks
 * (one process per GPU) by ONE kernel per rank: every rank writes its partial as 8-byte {value, tag} granules into slot `rank`
 * of every rank's MAILBOX (device memory of the owning rank, mapped into the peers through hipIpcMemHandle: the export / open /
 * close wrappers below), polls its own mailbox and sums the slots in rank order -- bit-identical on every rank.
 *   boxes:  HOST array of `world` device pointers, box[r] = rank r's mailbox as mapped in THIS process (box[rank] = the local
 *           allocation from qeft_oneshot_mailbox_alloc); each qeft_oneshot_mailbox_bytes(world, n) bytes, zeroed once;
 *   seq:    one zeroed uint32 in device memory (the call counter: every rank must make the same sequence of calls);
 *   status: two zeroed uint32 in device memory; status[0] != 0 after a call = a wait gave up (a peer never wrote).
 * hipGraph-capturable (no host state changes per call); world <= qeft_oneshot_max_world(). */
long long qeft_oneshot_mailbox_bytes(int world, int n);
int qeft_oneshot_max_world(void);
/* set-up helpers -- the ONLY entry points that allocate / synchronise: the mailbox is a zeroed hipMalloc block of its own (an
 * IPC handle exports the whole allocation its pointer lives in; a caching allocator's sub-block would arrive at an unknown offset) */
int qeft_oneshot_mailbox_alloc(int world, int n, void** dev_ptr_out);
int qeft_oneshot_mailbox_free(void* dev_ptr);
int qeft_oneshot_ipc_export(void* dev_ptr, void* handle_out /* 64 bytes, host */);
int qeft_oneshot_ipc_open(const void* handle /* 64 bytes, host */, void** dev_ptr_out);
int qeft_oneshot_ipc_close(void* dev_ptr);
int qeft_oneshot_allreduce_f32(void* t, int n, void* const* boxes, int rank, int world, void* seq, void* status,
                               qeft_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* QEFT_HIP_H */
