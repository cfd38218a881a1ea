// Small kernels around the quantized linears in a decode step: RMSNorm,
// rotary + KV-cache append + single-query attention, SiLU*mul.  They exist so the decode-token benchmark is a
// whole Llama step inside one hipGraph; none of them is on the packed-weight path itself.
//   RMSNorm    reference: qeft/kernel/layernorm/layernorm.cu:26-76 (generalT5LayerNorm): fp32 sum of squares,
//              y = x * rsqrt(mean(x^2) + eps) * gamma, one rounding to fp16.
//   attention  reference: qeft/kernel/attention (vendored FT masked MHA, ft_attention.cpp:110-181): rotary
//              (neox style = HF rotate_half), append k/v at `timestep`, softmax(q.K^T/sqrt(d)).V.
#include <cstdlib>

#include "qeft_common.h"
#include "decode_attn.h"      // the attention kernel; its body is a device function (the lab tools/attn_oproj_lab_kernel.h runs it too)

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
        for (int j = 0; j < 8; ++j) o[j] = mul_f32_to_f16((float)v[j] * rs, (float)g[j]);
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
        o[j] = mul_f32_to_f16(silu_f32((float)g[j]), (float)u[j]);
    }
    *(h8*)(out + i) = o;
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
            o[j] = mul_f32_to_f16((float)v[j], (float)g[j]);
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
                o[j] = mul_f32_to_f16(v[j], (float)g[j]);
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
    for (int i = threadIdx.x; i < H; i += 256) y[base + i] = mul_f32_to_f16(x[base + i] * rs, (float)gamma[i]);
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
            xh[c][j] = h2{mul_f32_to_f16(xv[c][2 * j] * rs, (float)g[2 * j]), mul_f32_to_f16(xv[c][2 * j + 1] * rs, (float)g[2 * j + 1])};
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
