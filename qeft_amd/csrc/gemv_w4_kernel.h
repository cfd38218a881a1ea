// Kernel templates of the decode GEMV (design notes: gemv_w4.hip).
#pragma once
#include "qeft_common.h"

namespace qeft {


// One ring slot = the HBM load of one step for one lane: 16 B of nibbles.
struct RingSlot {
    u32x4 w;
};

// LDS carve-up shared by kernel and launcher (bytes, every piece a multiple of 16).
__host__ __device__ constexpr int gemv_red_bytes(int NW, int RGI, int M) { return NW * RGI * 4 * M * 4; }
__host__ __device__ constexpr int gemv_slab_stride(int n_out) { return n_out + 8; }  // halves; +16 B spreads banks
__host__ __device__ constexpr size_t gemv_sz_bytes(int RGI, int K, int G) {   // [groups][rows][2] halves, 16-B padded
    return ((size_t)(K / G) * RGI * 4 * 4 + 15) / 16 * 16;
}
__host__ __device__ constexpr size_t gemv_smem_bytes(int NW, int RGI, int M, int K, int G, int n_out) {
    return (size_t)gemv_red_bytes(NW, RGI, M) + (n_out > 0 ? (size_t)RGI * 4 * gemv_slab_stride(n_out) * 2 : 0) +
           gemv_sz_bytes(RGI, K, G) + (size_t)M * (K / 32) * 8 + (size_t)M * K * 2;
}

// ABL: ablation bits for tools/gemv_lab.hip only (1: no scale/zero loads, 4: no math).  The product uses ABL = 0.
//
// RGI  row-groups covered by one wave-wide 16 B/lane load (1, 2 or 4): the load spans RGI row-groups x 512/RGI k.
// M    batch rows (1..7).
// D    ring depth: steps whose loads are in flight per wave.
// OUTL last n_out columns come from the fp16 slice.
// XG   x is gathered through reorder_ids while it is staged (o_proj).
//
// Arithmetic.  For a 32-k chunk c of row n inside group g the kernel needs  sum_k (q_k*s + sz) * x_k
//   = s * sum_k q_k x_k + sz * sum_k x_k.
// The nibbles are turned into fp16 with the 0x6400 exponent trick of the reference (dequantize.cuh:14-77) but
// the bias is NOT removed per weight: low nibbles become 1024+q, high nibbles 1024+16q.  x is staged in LDS
// with the positions that meet high nibbles pre-multiplied by 1/16 (x'), so
//   P = sum_lo (1024+q) x + sum_hi (1024+16q) x'  =  sum_k q_k x_k + 1024 * sum_k x'_k        (fp32, v_dot2c)
// and with A_c = 1024 * sum x'_k, B_c = sum x_k precomputed once per block and chunk (fp32, in LDS):
//   acc += s * (P - A_c) + sz * B_c.
// 5 integer ops + 4 dot2 per 8 weights instead of 13 + 4; weights are the exact q*s+sz (not rounded to fp16).
// NW   waves per block (4 or 8); wave w owns the steps w, w+NW, ...
// XT   transform applied to x while it is staged (M == 1 only): 0 none, 1 RMSNorm (x * rsqrt(mean x^2 + eps) * gamma,
//      a.xt_aux = gamma[K]), 2 SiLU-gate (silu(x) * up, a.xt_aux = up[K]).  Rounded to fp16 exactly like the
//      stand-alone kernels in decode_aux.hip, so fusing does not change results.
template <int NW, int RGI, int M, int D, bool OUTL, bool XG, int ABL = 0, int XT = 0>
__device__ __forceinline__ void gemv_w4_body(const GemvArgs& a, const int blk) {
    static_assert(XT == 0 || (M == 1 && !XG), "x transforms are for the batch-1 decode engine");
    constexpr int kWaves = NW, kBlock = NW * 64;
    constexpr int LPS = 64 / RGI;        // lanes per row-group within the wave
    constexpr int KSTEP = 512 / RGI;     // k advanced per step
    constexpr int ROWS = RGI * 4;        // rows per block
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    float* red = (float*)smem;                                               // [kWaves][M][ROWS]
    f16* slab = (f16*)(smem + gemv_red_bytes(NW, RGI, M));                        // [ROWS][n_out + 8] outlier slice
    const int slab_stride = gemv_slab_stride(a.n_out);
    uint32_t* szl = (uint32_t*)(slab + (OUTL ? ROWS * slab_stride : 0));      // [K/G][ROWS] (scale | scaled_zero << 16)
    float* corr = (float*)((uint8_t*)szl + gemv_sz_bytes(RGI, a.K, a.G));     // [M][K/32][2] = (A_c, B_c)
    f16* xs = (f16*)(corr + M * (a.K / 32) * 2);                              // [M][K] staged activations x'

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int sub = lane / LPS, lin = lane % LPS;
    const int r = (lane >> 1) & 3;
    const int koff = (lin >> 3) * 64 + (lane & 1) * 32;
    const int rg0 = blk * RGI;
    const int rg = rg0 + sub;
    const int row = rg * 4 + r;
    const int kq = a.K - (OUTL ? a.n_out : 0);      // INT4 columns [0, kq), fp16 outlier slice [kq, K)
    const int nsteps = (a.K + KSTEP - 1) / KSTEP;
    const int nfull = kq / KSTEP;                            // steps [0, nfull) are INT4 in every lane: the ring's work
    const int nsw = (nfull - wave + kWaves - 1) / kWaves;    // ring steps of this wave: wave, wave+NW, ...  (may be 0)

    // ---- 1. oldest loads: activations (and the outlier slab).  Unconditional (clamped) so the compiler's vmcnt
    //         bookkeeping stays exact: they are waited for first, with everything younger left in flight.
    const int xtotal = M * a.K;
    const int xvecs = xtotal / 8;
    u32x4 xst[XG ? 1 : 4];
    (void)xst;
    u32x4 ast[XT ? 4 : 1];
    (void)ast;
    if (!XG) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const size_t e = (size_t)min(p * kBlock + tid, xvecs - 1) * 8;
            xst[p] = *(const u32x4*)(a.x + e);
            if (XT) ast[p] = *(const u32x4*)(a.xt_aux + e);
        }
    }
    constexpr int NIR = (RGI == 1) ? 4 : ROWS / 2;                    // interleaved rows this block touches
    const int slab_vecs = OUTL ? NIR * (2 * a.n_out) / 8 : 0;         // 16-byte pieces of them
    // interleaved rows (rg0/2)*4 .. +NIR are contiguous: [NIR][2*n_out] halves (RGI == 1: only one parity is ours)
    const f16* osrc = OUTL ? a.ow_il + (size_t)(rg0 >> 1) * 4 * (2 * a.n_out) : nullptr;
    u32x4 ost = {0u, 0u, 0u, 0u};
    if (OUTL) ost = *(const u32x4*)(osrc + (size_t)min(tid, slab_vecs - 1) * 8);

    // the block's scales / scaled zeros: [K/G groups][ROWS rows]; a thread fetches the dwords of two adjacent rows
    const int szn = (a.K / a.G) * (ROWS / 2);
    uint32_t sst[4], zst[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int i = min(p * kBlock + tid, szn - 1);
        const uint32_t so = (uint32_t)(i / (ROWS / 2)) * (uint32_t)a.N + rg0 * 4 + (i % (ROWS / 2)) * 2;
        sst[p] = (ABL & 1) ? 0x1c001c00u : *(const uint32_t*)(a.scales + so);
        zst[p] = (ABL & 1) ? 0xa000a000u : *(const uint32_t*)(a.zeros + so);
    }

    // ---- 2. get the weight stream going: first D steps of this wave, branch-free
    const int s_last = max(wave + (nsw - 1) * kWaves, 0);
    RingSlot ring[D];
    const uint8_t* wbase = a.qw + (size_t)rg * a.K * 2;
    auto issue = [&](RingSlot& b, int s) {
        s = min(s, s_last);                              // past the wave's last step: harmless re-read of it
        // clamp inside the row-group: a wave without ring steps (K < one step) still issues its (unused) loads
        const uint32_t woff = min((uint32_t)s * (1024 / RGI) + lin * 16, (uint32_t)a.K * 2 - 16);
        b.w = __builtin_nontemporal_load((const u32x4*)(wbase + woff));
    };
#pragma unroll
    for (int d = 0; d < D; ++d) {
        issue(ring[d], wave + d * kWaves);
        __builtin_amdgcn_sched_barrier(0);   // keep slot order = issue order: slot 0 must be the oldest load
    }

    // ---- 2b. optional x transform on the register-held vectors (covers K <= 4 * 8 * block threads)
    if (XT == 1) {
        float ss = 0.f;
#pragma unroll
        for (int p = 0; p < 4; ++p)
            if (p * kBlock + tid < xvecs) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const h2 t = as_h2(xst[p][j]);
                    ss += (float)t[0] * (float)t[0] + (float)t[1] * (float)t[1];
                }
            }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
        if (lane == 0) red[wave] = ss;
        __syncthreads();
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) tot += red[w];
        const float rs = rsqrtf(tot / (float)a.K + a.xt_eps);
        __syncthreads();   // red is reused by the final reduction
#pragma unroll
        for (int p = 0; p < 4; ++p) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const h2 t = as_h2(xst[p][j]), gm = as_h2(ast[p][j]);
                xst[p][j] = as_u32(h2{mul_f32_to_f16((float)t[0] * rs, (float)gm[0]), mul_f32_to_f16((float)t[1] * rs, (float)gm[1])});
            }
        }
    } else if (XT == 2) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const h2 t = as_h2(xst[p][j]), u = as_h2(ast[p][j]);
                const float g0 = (float)t[0], g1 = (float)t[1];
                xst[p][j] = as_u32(h2{mul_f32_to_f16(silu_f32(g0), (float)u[0]), mul_f32_to_f16(silu_f32(g1), (float)u[1])});
            }
        }
    }

    // ---- 3. stage scales, x' (+ per-chunk sums) and the outlier slab into LDS
    for (int i = tid, p = 0; i < szn; i += kBlock, ++p) {
        uint32_t sv = sst[0], zv = zst[0];
        if (p == 1) { sv = sst[1]; zv = zst[1]; }
        if (p == 2) { sv = sst[2]; zv = zst[2]; }
        if (p == 3) { sv = sst[3]; zv = zst[3]; }
        if (p >= 4) {
            const uint32_t so = (uint32_t)(i / (ROWS / 2)) * (uint32_t)a.N + rg0 * 4 + (i % (ROWS / 2)) * 2;
            sv = *(const uint32_t*)(a.scales + so);
            zv = *(const uint32_t*)(a.zeros + so);
        }
        // (s_even, s_odd), (z_even, z_odd) -> (s_even | z_even << 16), (s_odd | z_odd << 16)
        *(u32x2*)(szl + i * 2) = u32x2{__builtin_amdgcn_perm(zv, sv, 0x05040100u), __builtin_amdgcn_perm(zv, sv, 0x07060302u)};
    }
    {
        const h2 k16th = {(f16)0.0625f, (f16)0.0625f};
        for (int v = tid, p = 0; v < xvecs; v += kBlock, ++p) {
            u32x4 xv;
            const int e = v * 8;
            const int bm = e / a.K, k = e - bm * a.K;
            if (XG) {
                // gathered input (qlinear.py:275): x[bm][ids[k]]
                f16 t[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) t[j] = a.x[(size_t)bm * a.K + a.ids[k + j]];
#pragma unroll
                for (int j = 0; j < 4; ++j) xv[j] = as_u32(h2{t[2 * j], t[2 * j + 1]});
            } else {
                xv = xst[0];
                if (p == 1) xv = xst[1];
                if (p == 2) xv = xst[2];
                if (p == 3) xv = xst[3];
                if (p >= 4) xv = *(const u32x4*)(a.x + (size_t)e);
            }
            float bsum = 0.f;   // sum of the true x
#pragma unroll
            for (int j = 0; j < 4; ++j) bsum += (float)as_h2(xv[j])[0] + (float)as_h2(xv[j])[1];
            const bool hi = (k & 8) && (k < kq);   // k%32 in [8,16) or [24,32): meets the high nibbles
            float asum = bsum;
            if (hi) {
                asum = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const h2 t = as_h2(xv[j]) * k16th;
                    xv[j] = as_u32(t);
                    asum += (float)t[0] + (float)t[1];
                }
            }
            *(u32x4*)(xs + e) = xv;
            // the 4 vectors of a 32-k chunk sit in 4 consecutive lanes
            asum += __shfl_xor(asum, 1);
            bsum += __shfl_xor(bsum, 1);
            asum += __shfl_xor(asum, 2);
            bsum += __shfl_xor(bsum, 2);
            if ((v & 3) == 0) {
                corr[(e >> 5) * 2] = 1024.f * asum;
                corr[(e >> 5) * 2 + 1] = bsum;
            }
        }
    }
    if (OUTL) {
        // pack_oweight layout (qlinear.py:70-79): interleaved row ir = blk*4 + rr holds per 32-col chunk 64 halves
        // alternating row blk*8+rr (even) and blk*8+rr+4 (odd).  A 16-byte piece = 4 columns of both rows.
        for (int v = tid; v < slab_vecs; v += kBlock) {
            const u32x4 ov = (v == tid) ? ost : *(const u32x4*)(osrc + (size_t)v * 8);
            const int per_row = (2 * a.n_out) / 8;
            const int ir = v / per_row, piece = v % per_row;         // piece: c32 = piece/8, jj0 = (piece%8)*4
            const int col = (piece >> 3) * 32 + (piece & 7) * 4;
            const u32x2 lo = {__builtin_amdgcn_perm(ov[1], ov[0], 0x05040100u), __builtin_amdgcn_perm(ov[3], ov[2], 0x05040100u)};
            const u32x2 hi = {__builtin_amdgcn_perm(ov[1], ov[0], 0x07060302u), __builtin_amdgcn_perm(ov[3], ov[2], 0x07060302u)};
            if (RGI == 1) {
                // this block owns rows (rg0&1)*4 + rr of the 8-row interleave block
                *(u32x2*)(slab + (size_t)ir * slab_stride + col) = (rg0 & 1) ? hi : lo;
            } else {
                const int lrow = (ir >> 2) * 8 + (ir & 3);
                *(u32x2*)(slab + (size_t)lrow * slab_stride + col) = lo;
                *(u32x2*)(slab + (size_t)(lrow + 4) * slab_stride + col) = hi;
            }
        }
    }
    __syncthreads();

    // ---- 4. the step loop
    float acc[M];
#pragma unroll
    for (int bm = 0; bm < M; ++bm) acc[bm] = 0.f;

    // sum over the 16 k-pairs of the lane's chunk: pairs[d] multiplies x dword d
    auto dot16 = [&](const uint32_t (&pairs)[16], int bm, int kx) -> float {
        const u32x4* px = (const u32x4*)(xs + (size_t)bm * a.K + kx);
        float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u32x4 xv = px[j];
            p0 = dot2(as_h2(pairs[4 * j + 0]), as_h2(xv[0]), p0);
            p1 = dot2(as_h2(pairs[4 * j + 1]), as_h2(xv[1]), p1);
            p2 = dot2(as_h2(pairs[4 * j + 2]), as_h2(xv[2]), p2);
            p3 = dot2(as_h2(pairs[4 * j + 3]), as_h2(xv[3]), p3);
        }
        return (p0 + p1) + (p2 + p3);
    };
    // opaque copy of the exponent pattern so that (v & mask) | MAGIC becomes ONE v_and_or_b32 (literal + VGPR)
    uint32_t MAGIC = 0x64006400u;
    asm volatile("" : "+v"(MAGIC));
    auto nib_pairs = [&](const u32x4& wv, uint32_t (&pairs)[16]) {
        // word w gives the pairs for x dwords w, w+4, w+8, w+12 (see qeft_common.h)
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint32_t v = wv[w], t = v >> 8;
            pairs[w] = (v & 0x000f000fu) | MAGIC;        // 1024 + q      k = 2w, 2w+1
            pairs[w + 4] = (v & 0x00f000f0u) | MAGIC;    // 1024 + 16q    k = 2w+8, 2w+9      (x' = x/16)
            pairs[w + 8] = (t & 0x000f000fu) | MAGIC;    // 1024 + q      k = 2w+16, 2w+17
            pairs[w + 12] = (t & 0x00f000f0u) | MAGIC;   // 1024 + 16q    k = 2w+24, 2w+25    (x' = x/16)
        }
    };
    auto int4_chunk = [&](const u32x4& wv, int k0) {
        uint32_t pairs[16];
        nib_pairs(wv, pairs);
        const h2 szp = as_h2(szl[(k0 >> a.gshift) * ROWS + sub * 4 + r]);
        const float sf = (float)szp[0], zf = (float)szp[1];
#pragma unroll
        for (int bm = 0; bm < M; ++bm) {
            const float2 ab = *(const float2*)(corr + ((size_t)bm * (a.K / 32) + (k0 >> 5)) * 2);
            const float p = dot16(pairs, bm, k0);
            acc[bm] += sf * (p - ab.x) + zf * ab.y;
        }
    };
    auto consume = [&](const RingSlot& b, int s) {
        if (ABL & 4) {
            acc[0] += __builtin_bit_cast(float, b.w[0] ^ b.w[1] ^ b.w[2] ^ b.w[3]);
            return;
        }
        int4_chunk(b.w, s * KSTEP + koff);
    };

    // edge steps [nfull, nsteps): per lane either INT4 (loaded on demand), the fp16 outlier slice (LDS slab) or past
    // the end of K.  With n_out a multiple of KSTEP this is pure LDS work, done while the ring's loads are in flight.
    for (int s = nfull + ((wave - nfull) % kWaves + kWaves) % kWaves; s < nsteps; s += kWaves) {
        const int k0 = s * KSTEP + koff;
        if (k0 < kq) {
            const u32x4 wv = *(const u32x4*)(wbase + (uint32_t)s * (1024 / RGI) + lin * 16);
            if (!(ABL & 4)) int4_chunk(wv, k0);
        } else if (OUTL && k0 < a.K) {
            const f16* pw = slab + (size_t)(sub * 4 + r) * slab_stride + (k0 - kq);
            uint32_t pairs[16];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const u32x4 v = *(const u32x4*)(pw + j * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e) pairs[j * 4 + e] = v[e];     // natural order: dword d <-> x dword d
            }
#pragma unroll
            for (int bm = 0; bm < M; ++bm) acc[bm] += dot16(pairs, bm, k0);
        }
    }

    const int nrounds = max((nsw + D - 1) / D, 1);
    for (int rd = 0; rd + 1 < nrounds; ++rd) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int si = rd * D + d;
            consume(ring[d], wave + si * kWaves);
            issue(ring[d], wave + (si + D) * kWaves);
            __builtin_amdgcn_sched_barrier(0);   // one slot at a time: later slots' waits must not be hoisted
        }
    }
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const int si = (nrounds - 1) * D + d;
        if (si < nsw) consume(ring[d], wave + si * kWaves);
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- 5. combine: lanes sharing a row differ in lane bit 0 and the tile bits (3 .. log2(LPS)-1)
#pragma unroll
    for (int bm = 0; bm < M; ++bm) {
        float v = acc[bm];
        v += __shfl_xor(v, 1);
#pragma unroll
        for (int o = 8; o < LPS; o <<= 1) v += __shfl_xor(v, o);
        acc[bm] = v;
    }
    if ((lin & ~6) == 0) {   // lin in {0,2,4,6} -> r = lin>>1
#pragma unroll
        for (int bm = 0; bm < M; ++bm) red[wave * (ROWS * M) + bm * ROWS + sub * 4 + r] = acc[bm];
    }
    __syncthreads();
    if (tid < ROWS * M) {
        const int bm = tid / ROWS, lrow = tid % ROWS;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) v += red[w * (ROWS * M) + tid];
        const int orow = rg0 * 4 + lrow;
        if (a.bias) v += (float)a.bias[orow];
        if (a.residual) v += (float)a.residual[(size_t)bm * a.N + orow];
        a.y[(size_t)bm * a.N + orow] = (f16)v;
    }
}

template <int NW, int RGI, int M, int D, bool OUTL, bool XG, int ABL = 0, int XT = 0>
__global__ __launch_bounds__(NW * 64) void gemv_w4_kernel(GemvArgs a) {
    gemv_w4_body<NW, RGI, M, D, OUTL, XG, ABL, XT>(a, blockIdx.x);
}

// Several linears that share the same input (q/k/v, gate/up) in ONE launch: block ranges [blk_end[p-1], blk_end[p])
// belong to part p.  Same K / group / n_out / batch for all parts; N may differ.
template <int NW, int RGI, int M, int D, bool OUTL, int XT = 0>
__global__ __launch_bounds__(NW * 64) void gemv_w4_group_kernel(GemvGroupArgs g) {
    int p = 0, blk = blockIdx.x;
    if (blk >= g.blk_end[0]) { p = 1; blk -= g.blk_end[0]; if (blockIdx.x >= (unsigned)g.blk_end[1]) { p = 2; blk = blockIdx.x - g.blk_end[1]; } }
    GemvArgs a;
    a.x = g.x;
    a.qw = g.qw[p];
    a.scales = g.scales[p];
    a.zeros = g.zeros[p];
    a.ow_il = g.ow_il[p];
    a.bias = g.bias[p];
    a.ids = nullptr;
    a.residual = nullptr;
    a.y = g.y[p];
    a.N = g.N[p];
    a.K = g.K;
    a.G = g.G;
    a.n_out = g.n_out;
    a.gshift = g.gshift;
    a.xt_aux = g.xt_aux;
    a.xt_eps = g.xt_eps;
    a.sz_blk = nullptr;
    a.dbg = nullptr;
    a.dbg2 = nullptr;
    a.ow_plain = nullptr;
    a.m_rt = 1;
    gemv_w4_body<NW, RGI, M, D, OUTL, false, 0, XT>(a, blk);
}

}  // namespace qeft
