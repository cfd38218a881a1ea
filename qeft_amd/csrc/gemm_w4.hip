// Prefill / fine-tune GEMMs for the QEFT packed W4 (+ fp16 outlier slice) linear on gfx950 MFMA.
//
// Replaces gemm_w4a16_T1/T2 (qeft/kernel/quantization_new/gemm/gemm_cuda.cu:290-927) and delivers the
// fused-outlier GEMM that gemm_cuda_qeft.cu only sketched (qlinear.py:266 does it as a second F.linear).
// Not a translation of the mma.m16n8k16/ldmatrix/cp.async pipeline: the design follows the wave64 MFMA
// operand maps.
//
// Forward  y[M,N] = x[M,K] . Wdeq[N,K]^T     (v_mfma_f32_32x32x16_f16, fp32 accumulate)
//   * MFMA sums over k in any order as long as A and B agree.  The checkpoint's nibble order puts the
//     k-pairs {2s+8j, 2s+8j+1 : j=0..3} of a 32-k chunk into u32 word s, so MFMA k-step s of chunk h uses
//     word s as the B fragment of lane (n = lane&31, h = lane>>5) with NO shuffling: a lane loads the 16
//     bytes of its (row n, chunk h) once per 64-k tile straight from HBM/L2 into VGPRs, dequantises in
//     registers (bit-identical fp16 weights to the reference) and feeds 4 MFMA k-steps.  Weights never
//     touch LDS.
//   * The A operand (activations) is staged through LDS with the matching permutation applied for free
//     while the registers are written out: dwords {s, s+4, s+8, s+12} of a 32-k chunk become 16-byte
//     slot s, so every A fragment is one ds_read_b128.  Slots are XOR-swizzled with (row>>1)&7 so the
//     16-lane ds_read_b128 groups hit 16 distinct 16-byte bank groups.
//   * k-tiles inside the last n_out columns take their B fragments from the fp16 oweight slice instead
//     of the (dead) nibbles: same MFMA stream, no second kernel, no read-modify-write of y.
#include <cstdlib>
#include <type_traits>

#include "qeft_common.h"

namespace qeft {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int GEMM_THREADS = 256;

__device__ __forceinline__ int a_slot_off(int row, int slot) { return row * 128 + ((slot ^ ((row >> 1) & 7)) << 4); }

struct BRegs {
    u32x4 q[2];       // packed nibbles of (n, chunk h) for the two 32-column n-tiles
    u32x4 o[2][4];    // fp16 outlier chunk (64 B) when the tile is in the outlier slice
    f16 s[2], z[2];
};

template <bool OUTL>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_w4_kernel(const f16* __restrict__ x, const uint8_t* __restrict__ qw,
                                                               const f16* __restrict__ scales,
                                                               const f16* __restrict__ zeros,
                                                               const f16* __restrict__ ow, const f16* __restrict__ bias,
                                                               f16* __restrict__ y, int M, int N, int K, int G,
                                                               int n_out) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[2 * BM * BK * 2];  // 2 x 16 KB

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int bm0 = blockIdx.x * BM, bn0 = blockIdx.y * BN;
    const int ktiles = K / BK;
    const int kq = K - n_out;  // first outlier column

    // A staging role: thread -> (row, 32-k chunk)
    const int arow = tid >> 1, ach = tid & 1;
    const bool arow_ok = bm0 + arow < M;
    const f16* aptr = x + (size_t)(bm0 + arow) * K + ach * 32;

    // B role: lane -> column n of two n-tiles, chunk h
    int ncol[2];
    bool nok[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        ncol[nt] = bn0 + wn * 64 + nt * 32 + r;
        nok[nt] = ncol[nt] < N;
        if (!nok[nt]) ncol[nt] = N - 1;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    u32x4 areg[4];
    BRegs breg;

    auto load_a = [&](int t) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            areg[j] = arow_ok ? ((const u32x4*)(aptr + (size_t)t * BK))[j] : u32x4{0u, 0u, 0u, 0u};
    };
    auto store_a = [&](int buf) {
        uint8_t* base = lds + buf * (BM * BK * 2);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            // slot s of the chunk = dwords {s, s+4, s+8, s+12}
            const u32x4 v = {areg[0][s], areg[1][s], areg[2][s], areg[3][s]};
            *(u32x4*)(base + a_slot_off(arow, ach * 4 + s)) = v;
        }
    };
    auto load_b = [&](int t, BRegs& b) {
        const int k0 = t * BK + h * 32;
        const bool outl = OUTL && k0 >= kq;
        const int g = k0 / G;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int n = ncol[nt];
            if (outl) {
                const u32x4* p = (const u32x4*)(ow + (size_t)n * n_out + (k0 - kq));
#pragma unroll
                for (int j = 0; j < 4; ++j) b.o[nt][j] = p[j];
            } else {
                b.q[nt] = *(const u32x4*)(qw + (size_t)(n >> 2) * K * 2 + (size_t)t * 128 + (n & 3) * 32 + h * 16);
                b.s[nt] = scales[(size_t)g * N + n];
                b.z[nt] = zeros[(size_t)g * N + n];
            }
        }
    };

    load_a(0);
    load_b(0, breg);
    store_a(0);
    __syncthreads();

    for (int t = 0; t < ktiles; ++t) {
        const int buf = t & 1;
        BRegs bnext;
        const bool more = t + 1 < ktiles;
        if (more) {
            load_a(t + 1);
            load_b(t + 1, bnext);
        }
        const bool outl = OUTL && (t * BK + h * 32) >= kq;
        const uint8_t* abase = lds + buf * (BM * BK * 2);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            h8 bf[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                u32x4 bw;
                if (outl) {
                    bw = u32x4{breg.o[nt][0][s], breg.o[nt][1][s], breg.o[nt][2][s], breg.o[nt][3][s]};
                } else {
                    h2 wd[4];
                    dequant8(breg.q[nt][s], splat(breg.s[nt]), splat(breg.z[nt]), wd);
                    bw = u32x4{as_u32(wd[0]), as_u32(wd[1]), as_u32(wd[2]), as_u32(wd[3])};
                }
                bf[nt] = __builtin_bit_cast(h8, bw);
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int row = wm * 64 + mt * 32 + r;
                const h8 af = *(const h8*)(abase + a_slot_off(row, h * 4 + s));
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf[nt], acc[mt][nt], 0, 0, 0);
            }
        }
        if (more) {
            store_a(buf ^ 1);
            breg = bnext;
        }
        __syncthreads();
    }

    // D map (32x32): col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        if (!nok[nt]) continue;
        const int n = ncol[nt];
        const float bv = bias ? (float)bias[n] : 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = bm0 + wm * 64 + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (m < M) y[(size_t)m * N + n] = (f16)(acc[mt][nt][e] + bv);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Forward GEMM, pipelined version (the one gemm_w4_launch uses whenever K has >= 4 k-tiles).
//   * 128 x 128 x 64 block tile, 4 waves side by side along N (each 128 rows x 32 columns = 4 MFMA 32x32 tiles), so
//     every weight is dequantised by exactly one wave.
//   * Both operands reach LDS by LDS-DMA (global_load_lds, 16 B/lane, no VGPR round trip) into a ring of kStages
//     tiles; kStages-1 tiles stay in flight behind a COUNTED s_waitcnt vmcnt and ONE raw s_barrier per k-tile.
//     A keeps its natural k order (slot j of a 32-k chunk = 8 consecutive k = the A fragment of k-step j); B's
//     dequantised pairs are written straight into the matching fragment registers (pair j of word w -> fragment j,
//     position w).  Slots are XOR-swizzled with (row & 7) on the SOURCE address (the DMA destination is lane-linear).
//   * scales / scaled zeros of the block's 128 columns are staged once into LDS as (s | sz << 16) words.
//   * k-tiles inside the fp16 outlier slice take their B fragments straight from oweight (2 tiles for r = 128).
// ---------------------------------------------------------------------------------------------------
constexpr int kStages = 3;
constexpr int kTileBytes = BM * BK * 2 + BN * BK / 2;   // 16 KB of A + 4 KB of packed B

__host__ __device__ constexpr size_t gemm_v2_smem(int K, int G, int nwv = 4) {
    return (size_t)kStages * (BM * BK * 2 + 32 * nwv * BK / 2) + (size_t)(K / G) * 32 * nwv * 4;
}

// LDS reads of DMA-written tiles go through inline asm: hipcc (ROCm 7.2) otherwise drains the whole DMA pipeline with
// s_waitcnt vmcnt(0) before any ds_read that might alias an LDS-DMA in flight.  The loads are made visible to the
// consumer only by lds_wait() (s_waitcnt lgkmcnt(0) + a scheduling barrier), guide section 5.7 form (iii).
__device__ __forceinline__ u32x4 lds_read16(uint32_t addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
__device__ __forceinline__ uint32_t lds_read4(uint32_t addr) {
    uint32_t v;
    asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
// gfx950 transposed LDS read: per 16-lane group a block of 4 rows x 16 columns of 16-bit elements comes back
// column-major -- lane 4q+p supplies the address of row q, columns 4p..4p+3; lane i receives column i of the 4 rows.
// EXEC must be all ones; results are visible after lds_wait().
__device__ __forceinline__ u32x2 lds_read_tr8(uint32_t addr) {
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
__device__ __forceinline__ void lds_wait() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void wait_vm(int n) {   // counted wait: n outstanding vector-memory ops allowed
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
    }
}

// NWV waves side by side in N, each 128 x 32: block tile 128 x (32 NWV).  NWV = 8 (128 x 256) reads the activation tile
// once for twice the columns: 24 KB per 4.2 MFLOP instead of 20 KB per 2.1 (the GEMM is bound by the L2/MALL ->
// LDS traffic of the activation tiles, the packed weights are tiny).
template <bool OUTL, int NWV>
__global__ __launch_bounds__(64 * NWV) void gemm_w4_kernel_v2(const f16* __restrict__ x, const uint8_t* __restrict__ qw,
                                                                  const f16* __restrict__ scales,
                                                                  const f16* __restrict__ zeros,
                                                                  const f16* __restrict__ ow, const f16* __restrict__ bias,
                                                                  f16* __restrict__ y, int M, int N, int K, int G,
                                                                  int n_out, float* __restrict__ part) {
    // Split-K (gridDim.z = S > 1, mid-size M: too few 128x128 tiles to fill the chip and a K loop that is bound by
    // request latency): block z contracts its share of the INT4 k-tiles (the last one also the fp16 tiles) and writes
    // an fp32 partial tile to part[z][M][N]; gemm_splitk_reduce_kernel sums the S partials in order, adds the bias
    // and rounds once.
    constexpr int BN = 32 * NWV, GEMM_THREADS = 64 * NWV;             // (shadow the file-level 128-wide constants)
    constexpr int kTileBytes = BM * BK * 2 + BN * BK / 2;             // 16 KB of A + NWV KB of packed B
    constexpr int kAI = 16 / NWV;                                     // A-tile DMA instructions per wave (1 KB each)
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];     // [kStages][A 16 KB | B NWV KB] [sz]
    uint32_t* szl = (uint32_t*)(lds + kStages * kTileBytes);          // [K/G][BN]
    const uint32_t lds0 = (uint32_t)(uintptr_t)lds;                   // LDS byte address of the array (for the asm reads)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int bm0 = blockIdx.x * BM, bn0 = blockIdx.y * BN;
    const int ktiles = K / BK;
    const int kq = K - (OUTL ? n_out : 0);
    const int qtiles = kq / BK;            // INT4 k-tiles [0, qtiles); outlier k-tiles [qtiles, ktiles) (n_out % 64 == 0)
    const int S = gridDim.z, z = blockIdx.z;
    const int kt0 = (int)((long long)qtiles * z / S), kt1 = (int)((long long)qtiles * (z + 1) / S);   // this block's INT4 tiles
    const int ngroups = K / G;
    const int gshift = 31 - __builtin_clz(G);    // G is a power of two on this path (checked by the launcher)

    const int nloc = wave * 32 + r;
    const int ncol = min(bn0 + nloc, N - 1);
    const bool nok = bn0 + nloc < N;

    // ---- scales of the block's columns -> LDS (once)
    for (int i = tid; i < ngroups * BN; i += GEMM_THREADS) {
        const int g = i / BN, c = min(bn0 + (i % BN), N - 1);
        const uint32_t sv = ((const uint16_t*)scales)[(size_t)g * N + c], zv = ((const uint16_t*)zeros)[(size_t)g * N + c];
        szl[i] = sv | (zv << 16);
    }

    // ---- DMA sources.  A: wave w, instruction i -> rows (w*4+i)*8 + lane/8, slot' = lane%8 holds chunk slot'^((row>>1)&7)
    const f16* asrc[kAI];
#pragma unroll
    for (int i = 0; i < kAI; ++i) {
        const int row = (wave * kAI + i) * 8 + (lane >> 3);
        const int grow = min(bm0 + row, M - 1);
        asrc[i] = x + (size_t)grow * K + (((lane & 7) ^ ((row >> 1) & 7)) << 3);
    }
    const int brg = min(bn0 / 4 + (tid >> 3), N / 4 - 1);
    const uint8_t* bsrc = qw + (size_t)brg * K * 2 + (tid & 7) * 16;

    auto stage = [&](int t) {
        uint8_t* base = lds + (t % kStages) * kTileBytes;
#pragma unroll
        for (int i = 0; i < kAI; ++i)
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(asrc[i] + (size_t)t * BK),
                                             (void __attribute__((address_space(3)))*)(base + (wave * kAI + i) * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(bsrc + (size_t)min(t, qtiles > 0 ? qtiles - 1 : 0) * 128),
                                         (void __attribute__((address_space(3)))*)(base + BM * BK * 2 + wave * 1024), 16, 0, 0);
    };

    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

    // per-lane LDS offsets of the A fragments: m-tile mt, k-step j -> row*128 + ((h*4+j) ^ (row&7))*16
    uint32_t aoff[4];
#pragma unroll
    // swizzle by (row >> 1) & 7: 128-byte rows put two rows in one 64-bank period, so the 16 lanes of a ds_read_b128 group
    // (rows r .. r+15) hit 8 even and 8 odd rows -- the 8 rows of one parity need 8 different slots (SQ_LDS_BANK_CONFLICT
    // was 49 % of the LDS cycles with the (row & 7) swizzle, which repeats after 8 rows)
    for (int j = 0; j < 4; ++j) aoff[j] = (uint32_t)(r * 128 + (((h * 4 + j) ^ ((r >> 1) & 7)) << 4));   // mt*32 rows keep (row >> 1) & 7
    const uint32_t boff = (uint32_t)(BM * BK * 2 + (nloc >> 2) * 128 + (nloc & 3) * 32 + h * 16);
    const uint32_t szoff = (uint32_t)(kStages * kTileBytes + nloc * 4);

    __syncthreads();   // scale words visible to every wave; no DMA is in flight yet, so the implied vmcnt(0) is free
#pragma unroll
    for (int t = 0; t < kStages - 1; ++t)
        if (kt0 + t < kt1) stage(kt0 + t);

    auto mma_tile = [&](uint32_t tbase, const u32x4 (&bfrag)[4], u32x4 (&a0)[4]) {
        u32x4 a1[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            // fetch the next m-tile's fragments while this one's MFMAs run
            if (mt + 1 < 4) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const u32x4 v = lds_read16(tbase + (mt + 1) * 32 * 128 + aoff[j]);
                    if (mt & 1) a0[j] = v; else a1[j] = v;
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, (mt & 1) ? a1[j] : a0[j]),
                                                                __builtin_bit_cast(h8, bfrag[j]), acc[mt], 0, 0, 0);
            lds_wait();
        }
    };

    // ---- main loop over the INT4 k-tiles: kStages-1 tiles in flight, one barrier per tile
    for (int t = kt0; t < kt1; ++t) {
        const int younger = min(kt1 - 1 - t, kStages - 2);
        wait_vm(younger * (kAI + 1));
        __builtin_amdgcn_s_barrier();      // every wave's DMA for tile t landed; everyone is done reading tile t-1
        if (t + kStages - 1 < kt1) stage(t + kStages - 1);   // overwrites the buffer of tile t-1

        const uint32_t tbase = lds0 + (uint32_t)(t % kStages) * kTileBytes;
        const u32x4 q = lds_read16(tbase + boff);
        const uint32_t szw = lds_read4(lds0 + szoff + (uint32_t)(((t * BK + h * 32) >> gshift) * BN * 4));
        u32x4 a0[4];   // first m-tile's A fragments: in flight while B is dequantised
#pragma unroll
        for (int j = 0; j < 4; ++j) a0[j] = lds_read16(tbase + aoff[j]);
        lds_wait();
        // B fragments: k-step j contracts the 8 CONSECUTIVE k h*32 + 8j .. +7 (= A's natural slot j); they are pair j of
        // each of the 4 nibble words, so the dequantised pairs are written to fragment j, position w (a register naming).
        u32x4 bfrag[4];
        {
            const h2 sz = as_h2(szw);
            const h2 sc = splat(sz[0]), zc = splat(sz[1]);
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                h2 wd[4];
                dequant8(q[w], sc, zc, wd);           // wd[j] = pair (k = 2w + 8j, 2w + 8j + 1)
#pragma unroll
                for (int j = 0; j < 4; ++j) bfrag[j][w] = as_u32(wd[j]);
            }
        }
        mma_tile(tbase, bfrag, a0);
    }

    // ---- fp16 outlier k-tiles (2 for r = 128): pipeline is empty; A by DMA, B fragments straight from oweight
    if (OUTL && z == S - 1) {
        for (int t = qtiles; t < ktiles; ++t) {
            __builtin_amdgcn_s_barrier();          // all waves finished reading the buffers of earlier tiles
            uint8_t* base = lds;                   // stage buffer 0
#pragma unroll
            for (int i = 0; i < kAI; ++i)
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(asrc[i] + (size_t)t * BK),
                                                 (void __attribute__((address_space(3)))*)(base + (wave * kAI + i) * 1024), 16, 0, 0);
            u32x4 bfrag[4];
            const u32x4* p = (const u32x4*)(ow + (size_t)ncol * n_out + (t * BK + h * 32 - kq));
#pragma unroll
            for (int j = 0; j < 4; ++j) bfrag[j] = p[j];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            u32x4 a0[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) a0[j] = lds_read16(lds0 + aoff[j]);
            lds_wait();
            mma_tile(lds0, bfrag, a0);
        }
    }

    if (nok && S > 1) {
        float* pz = part + (size_t)z * M * N;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = bm0 + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (m < M) pz[(size_t)m * N + ncol] = acc[mt][e];
            }
    } else if (nok) {
        const float bv = bias ? (float)bias[ncol] : 0.f;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = bm0 + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (m < M) y[(size_t)m * N + ncol] = (f16)(acc[mt][e] + bv);
            }
    }
}

// ---------------------------------------------------------------------------------------------------
// Forward GEMM, large-M version (M >= ~2048): 256 (m) x 128 (n) x 64 (k) block tile, 4 compute waves side by side along N,
// each 256 rows x 32 columns = 8 MFMA 32x32 tiles, plus 4 loader waves (one compute + one loader wave per SIMD).
//
// Why this shape (profiles/r01_gemm_pmc.txt: the 128-row tiles issue ~0.85 VALU cycles per MFMA cycle with two waves
// per SIMD and run the matrix pipe at 57 %): the dequantisation of a wave's 32 columns x 64 k costs the same ~56 VALU
// whatever the tile height, so 256 rows halve the VALU, LDS-DMA and barrier work per MFMA (32 MFMAs per k-tile and wave
// against 16).  The grid is one block per CU at 2048 x 4096 (256 blocks), dealt so that an XCD works on one 256-row
// block of the activations (2 MB at K = 4096: resident in its 4 MB L2) while the packed weights stream past.
//   * LDS: a 4-stage ring of activation tiles (32 KB each, three in flight from L2) and a 6-slot ring of the packed weight
//     tiles + their scales (4.5 KB each, five in flight: they come from HBM / MALL, and a k-tile gated by a 2-3 us weight
//     fetch with only two tiles of lead ran the first version at a third of the MFMA rate); all by LDS-DMA behind a
//     counted s_waitcnt, ONE s_barrier per k-tile (32 MFMAs per wave between barriers).
//   * a wave DMAs and reads its OWN 1 KB of the packed B tile (its 32 columns), so B needs no cross-wave hand-off;
//     the B fragments of k-tile t + 1 are dequantised in registers while the MFMAs of k-tile t run.
//   * fp16 outlier k-tiles, bias and the fp16 epilogue as in gemm_w4_kernel_v2.
// ---------------------------------------------------------------------------------------------------
constexpr int G3_BM = 256, G3_BN = 128, G3_ST = 4, G3_BST = 6;
constexpr int G3_A = G3_BM * BK * 2, G3_B = G3_BN * BK / 2, G3_S = 512;
constexpr int G3_BOFF = G3_ST * G3_A, G3_SOFF = G3_BOFF + G3_BST * G3_B;      // [4 x A 32 KB][6 x B 4 KB][6 x scales 512 B]
constexpr size_t G3_SMEM = (size_t)G3_SOFF + G3_BST * G3_S;                   // 158720 bytes
constexpr int G3_YP = G3_BN * 2 + 16;     // pitch of the fp16 output tile staged in LDS for the epilogue (256 rows: 68 KB of the ring)

// n DMA pieces of 1 KB: lane's 16 bytes at (sbase + voff_i) -> LDS[lds_dst + 1024 i + 16 lane], i = 0..7 (one asm statement:
// M0 is written and read inside it; the pieces stay invisible to hipcc's s_waitcnt bookkeeping and are counted by hand)
__device__ __forceinline__ void g3_dma_a8(const void* sbase, const uint32_t (&voff)[8], uint32_t lds_dst) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %1\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %1\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, %1\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %6, %1\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %7, %1\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %8, %1\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %9, %1\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %10, %1\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(sbase), "s"(lds_dst), "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3]), "v"(voff[4]), "v"(voff[5]),
          "v"(voff[6]), "v"(voff[7])
        : "memory", "scc");
}
// four pieces (the 128-row tile's loader waves)
__device__ __forceinline__ void g3_dma_a4(const void* sbase, const uint32_t (&voff)[4], uint32_t lds_dst) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %1\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %1\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, %1\n\t"
        "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %6, %1\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(sbase), "s"(lds_dst), "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3])
        : "memory", "scc");
}
__device__ __forceinline__ void g3_dma_a(const void* sbase, const uint32_t (&voff)[8], uint32_t lds_dst) { g3_dma_a8(sbase, voff, lds_dst); }
__device__ __forceinline__ void g3_dma_a(const void* sbase, const uint32_t (&voff)[4], uint32_t lds_dst) { g3_dma_a4(sbase, voff, lds_dst); }
__device__ __forceinline__ void g3_dma16(const void* sbase, uint32_t voff, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(sbase), "s"(lds_dst), "v"(voff) : "memory");
}
__device__ __forceinline__ void g3_dma4(const void* sbase, uint32_t voff, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %3, %1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(sbase), "s"(lds_dst), "v"(voff) : "memory");
}

// ABL (lab only, QEFT_GEMM_ABL): 1 = the loader waves issue no DMA, 2 = the compute waves skip their k-tile bodies -- wrong
// results, but the two timings say which side of the block sets the pace (profiles/r02_gemm_v3_ablation.txt).
// EPI 1: y = silu(gate) * (x . W^T + bias) with gate [M][N] fp16 -- the MLP's SiLU(gate) * up formed in the up_proj launch's
// epilogue (same rounding as the unfused pair: the product is rounded to fp16 first, then silu_mul_kernel's formula).
// EPI 2: W holds gate and up interleaved in blocks of 64 rows (fuse.pair64_gemm_operand), so that columns 0..63 of a block's
// tile are gate and 64..127 the same columns of up: y [M][N/2] = silu(tile[:, :64]) * tile[:, 64:], one launch for both
// linears, no gate round trip through HBM.
// MT: 32-row m-tiles per block tile: 8 (256 x 128, the M >= 1024 tier) or 4 (128 x 128, round 3: the tier between the small-M
// kernels and 224 tiles of 256 rows -- M = 512 .. 1024 of a short prompt or a fine-tune batch; same roles, rings and waits, the
// activation stage is 16 KB instead of 32 and a loader wave stages 4 pieces of it instead of 8).
// BITS = 3 (round 3): `qw` is the 3-bit EXTENSION layout int32 [N/16][(K - n_out)/128][4 chunks][16 rows][3] (oracle.pack_w3) --
// the loader waves stage the two 384-byte runs (2 row sets x 2 chunks x 16 rows x 12 B) of a compute wave's 32 columns per
// k-tile, the compute lanes unpack their 12-byte record (row, chunk) in registers: no 3 -> 4-bit expansion pass.
template <bool OUTL, int ABL = 0, int EPI = 0, int MT = 8, int BITS = 4>
__global__ __launch_bounds__(512) void gemm_w4_kernel_v3(const f16* __restrict__ x, const uint8_t* __restrict__ qw,
                                                           const f16* __restrict__ scales, const f16* __restrict__ zeros,
                                                           const f16* __restrict__ ow, const f16* __restrict__ bias,
                                                           f16* __restrict__ y, int M, int N, int K, int G, int n_out, int NB,
                                                           const f16* __restrict__ gate, float* __restrict__ part, int S) {
    // Split-K (part != NULL, S > 1; round 3): the grid holds S blocks per output tile, block z of a tile contracts the k-tiles
    // [z KT, (z + 1) KT) (KT = K / 64 / S; the fp16 outlier k-tiles fall to the last block) and writes an fp32 partial tile to
    // part[z][M][N]; gemm_splitk_reduce_kernel sums the S partials in order, adds the bias and rounds once (deterministic).
    // 8 waves, two per SIMD with fixed roles: waves 0..3 compute (wave w: columns 32 w .. 32 w + 31 of the tile, all 256 rows),
    // waves 4..7 load (wave 4 + l: activation pieces 8 l .. 8 l + 7, the packed weights of compute wave l, scales / zeros).
    // An LDS-DMA instruction holds its wave for 100-200 cycles at issue; in the compute waves' own stream (first version)
    // that stalled the matrix pipe 8 times per k-tile.  Loader waves absorb it in the shadow of their SIMD partner's MFMAs.
    static_assert(MT == 8 || MT == 4, "256- or 128-row block tiles");
    static_assert(MT == 8 || EPI == 0, "the fused activation epilogues exist for the 256-row tile only");
    constexpr int BMR = 32 * MT, AP = MT;                     // rows of the tile; activation pieces (1 KB) per loader wave and stage
    constexpr int A_B = BMR * BK * 2;                         // bytes of an activation stage
    constexpr int BOFF = G3_ST * A_B, SOFF = BOFF + G3_BST * G3_B;      // [4 x A][6 x B 4 KB][6 x scales 512 B]
    extern __shared__ __attribute__((aligned(1024))) uint8_t lds[];
    const long long t_entry = ABL == 6 ? wall_clock64() : 0;
    const uint32_t lds0 = (uint32_t)(uintptr_t)lds;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-contiguous block order (blocks b, b + 8, .. share an L2): an XCD works through consecutive tiles of one row block
    const int nblk = gridDim.x, bq = nblk >> 3, br = nblk & 7, bx = blockIdx.x & 7;
    const int c = (bx < br ? bx * (bq + 1) : br * (bq + 1) + (bx - br) * bq) + (blockIdx.x >> 3);
    const int tiles = nblk / S, zsp = c / tiles, ct = c - zsp * tiles;
    const int bm0 = (ct / NB) * BMR, bn0 = (ct % NB) * G3_BN;
    const int ktiles = K / BK / S, kt0 = zsp * ktiles;      // this block's k-tiles: global kt0 .. kt0 + ktiles - 1, local index t
    const int kq = K - (OUTL ? n_out : 0);
    const int qtiles = min(max(kq / BK - kt0, 0), ktiles);   // local INT4 k-tiles [0, qtiles); fp16 outlier k-tiles [qtiles, ktiles)
    constexpr int LEAD = G3_ST - 1, LEAD_B = G3_BST - 1;

    if (wave >= 4) {
        // =========================================================== loader waves
        const int l = wave - 4;
        const int gshift = 31 - __builtin_clz(G);
        // A piece p = 8 rows x 128 B: row 8p + lane/8, LDS chunk lane%8 holds global chunk (lane%8) ^ ((row >> 1) & 7)
        uint32_t a_off[AP];
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const int row = (l * AP + i) * 8 + (lane >> 3);
            const int grow = min(bm0 + row, M - 1);
            a_off[i] = (uint32_t)grow * (uint32_t)K * 2u + (uint32_t)(((lane & 7) ^ ((row >> 1) & 7)) << 4);
        }
        // B: the 1 KB of compute wave l's 32 columns (row groups bn0/4 + 8l ..): lane -> (row group lane/8, 16-byte piece lane%8)
        const int brg = min(bn0 / 4 + l * 8 + (lane >> 3), N / 4 - 1);
        // 3 bits: lanes 0..23 the 384 bytes of row set bn0 / 16 + 2 l, lanes 24..47 the next set's (the rest repeat the last piece)
        const int l3 = min(lane, 47), set3 = min(bn0 / 16 + 2 * l + l3 / 24, N / 16 - 1);
        const uint32_t b_off = BITS == 3 ? (uint32_t)set3 * (uint32_t)(kq / 128) * 768u + (uint32_t)(l3 % 24) * 16u
                                         : (uint32_t)brg * (uint32_t)K * 2u + (uint32_t)(lane & 7) * 16u;
        // scales (l == 0) / scaled zeros (l == 1) of the tile's group: 128 columns x 2 B = 64 lanes x 4 B
        const uint32_t s_off = (uint32_t)min(bn0 + 2 * lane, N - 2) * 2u;
        const uint8_t* const sz_base = (const uint8_t*)(l == 0 ? scales : zeros);
        auto stage_a = [&](int t) {            // 8 DMA instructions
            if (ABL == 1 || ABL >= 3) return;
            g3_dma_a((const uint8_t*)x + (size_t)(kt0 + t) * (BK * 2), a_off, lds0 + (uint32_t)(t & (G3_ST - 1)) * A_B + (uint32_t)l * (AP * 1024u));
        };
        auto stage_b = [&](int t, int slot) {  // 2 (l < 2) or 1 DMA instructions
            if (ABL == 1 || ABL >= 3) return;
            const int tg = kt0 + t;
            g3_dma16(qw + (BITS == 3 ? (size_t)(tg >> 1) * 768 + (size_t)(tg & 1) * 384 : (size_t)tg * 128), b_off,
                     lds0 + BOFF + (uint32_t)slot * G3_B + (uint32_t)l * 1024u);
            if (l < 2)
                g3_dma4(sz_base + (size_t)((tg * BK) >> gshift) * N * 2, s_off, lds0 + SOFF + (uint32_t)slot * G3_S + (uint32_t)l * 256u);
        };
        // In-order completion: at the top of iteration t the activations of k-tile t + 1 (the LAST thing iteration t - 2
        // issued) must have landed; younger than them is exactly what iteration t - 1 issued: [weights / scales of k-tile
        // t + 4: 2 pieces for l < 2, else 1] then [8 activation pieces of k-tile t + 2].  The weights of k-tile t + 1 are
        // older still (iteration t - 4).  Towards the end the refills stop (weights first).
        auto wait_prev = [&](bool prev_a, bool prev_b) {
            if (!prev_a) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (!prev_b) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AP) : "memory");
            else if (l < 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AP + 2) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AP + 1) : "memory");
        };
#pragma unroll
        for (int t = 0; t < LEAD_B; ++t)
            if (t < qtiles) stage_b(t, t);
#pragma unroll
        for (int t = 0; t < LEAD; ++t)
            if (t < ktiles) stage_a(t);
        // activations of k-tile 0 landed <=> at most the 16 youngest pieces (k-tiles 1, 2) outstanding
        if (ktiles >= LEAD) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * AP) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        int slot_f = LEAD_B % G3_BST;
        for (int t = 0; t < ktiles; ++t) {         // INT4 and fp16 outlier k-tiles alike (the latter have no weight pieces)
            if (t == 0) {
                if (ktiles >= LEAD) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AP) : "memory");     // k-tile 1: all but k-tile 2's pieces
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                wait_prev((t - 1) + LEAD < ktiles, (t - 1) + LEAD_B < qtiles);
            }
            if (ABL < 3) __builtin_amdgcn_s_barrier();      // k-tile t + 1 visible to everyone; everyone is done with k-tile t - 1
            if (t + LEAD_B < qtiles) stage_b(t + LEAD_B, slot_f);      // BEFORE the activation pieces: see wait_prev
            if (t + LEAD < ktiles) stage_a(t + LEAD);
            slot_f = slot_f + 1 == G3_BST ? 0 : slot_f + 1;
        }
        return;
    }

    // =============================================================== compute waves
    const int r = lane & 31, h = lane >> 5;
    const int nloc = wave * 32 + r;
    const int ncol = min(bn0 + nloc, N - 1);

    f32x16 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

    // per-lane LDS offsets: A fragment of m-tile mt, k-step j at stage + a_rd[j] + mt * 4096
    uint32_t a_rd[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) a_rd[j] = (uint32_t)(r * 128 + (((h * 4 + j) ^ ((r >> 1) & 7)) << 4));
    const uint32_t b_rd = BITS == 3 ? (uint32_t)(BOFF + wave * 1024 + (r >> 4) * 384 + h * 192 + (r & 15) * 12)
                                    : (uint32_t)(BOFF + (nloc >> 2) * 128 + (nloc & 3) * 32 + h * 16);        // + slot * G3_B
    const uint32_t s_rd = (uint32_t)(SOFF + nloc * 2);                                             // + slot * G3_S

    // B fragments of a k-tile: k-step j contracts the 8 consecutive k h*32 + 8j .. +7 = pair j of each of the 4 nibble words
    uint32_t MAGIC3 = 0x64006400u;
    asm volatile("" : "+v"(MAGIC3));
    auto read_q = [&](int slot) {              // the lane's packed record of a k-tile: 16 bytes (4 bits) or 12 (3 bits, word 3 unused)
        if constexpr (BITS == 3) {
            const uint32_t* p = (const uint32_t*)(lds + b_rd + slot * G3_B);
            return u32x4{p[0], p[1], p[2], 0u};
        } else {
            return *(const u32x4*)(lds + b_rd + slot * G3_B);
        }
    };
    auto dequant_tile = [&](int slot, u32x4 (&bf)[4]) {
        const u32x4 q = read_q(slot);
        const h2 sc = splat(*(const f16*)(lds + s_rd + slot * G3_S)), zc = splat(*(const f16*)(lds + s_rd + slot * G3_S + 256));
        if constexpr (BITS == 3) {
#pragma unroll
            for (int e = 0; e < 16; ++e)       // fragment j, word w = pair 4 j + w of the chunk (k = h * 32 + 8 j + 2 w, + 1)
                bf[e >> 2][e & 3] = as_u32(__builtin_elementwise_fma(w3_pair_q(q[0], q[1], q[2], e, MAGIC3), sc, zc));
        } else {
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                h2 wd[4];
                dequant8(q[w], sc, zc, wd);
#pragma unroll
                for (int j = 0; j < 4; ++j) bf[j][w] = as_u32(wd[j]);
            }
        }
    };
    u32x4 bA[4], bB[4];        // B fragments of the even / odd k-tiles
    u32x4 fa[4], fb[4];        // A fragments, two m-tiles in rotation
    __builtin_amdgcn_s_barrier();              // k-tile 0 visible
    asm volatile("" ::: "memory");
    if (qtiles > 0) {
        dequant_tile(0, bA);
#pragma unroll
        for (int j = 0; j < 4; ++j) fa[j] = *(const u32x4*)(lds + a_rd[j]);
        if (ABL == 5)
            for (int j = 0; j < 4; ++j) fb[j] = fa[j];
    }

    // One k-tile = 8 phases (one per 32-row m-tile), each: the fetch of the NEXT m-tile's fragments (the last phase fetches
    // m-tile 0 of k-tile t + 1), then 4 MFMAs on the fragments fetched during the previous phase and one eighth of the
    // dequantisation of k-tile t + 1's B.  The sched_barriers pin that order: left alone, hipcc sinks the fetches behind the
    // third MFMA (one MFMA of lead instead of a whole phase; lgkmcnt waits were 22 % of the compute waves' time) or regroups
    // everything into one VALU block and one MFMA block.
    // The packed words / scale / zero of k-tile t + 1 are already in registers (qn, sn_, zn_) when the tile starts, and the
    // last phase fetches those of k-tile t + 2 (landed before barrier t: the loaders issue the weights of k-tile t + 3 ahead
    // of the activations of k-tile t + 1 they wait for), so nothing the first MFMAs need is issued after the barrier.
    // TAIL = false: k-tiles t + 1 and t + 2 are INT4 k-tiles (for t = qtiles - 2 the fetch for "t + 2" reads a stale slot and
    // is never used).  TAIL = true: k-tile t + 1, if there is one, is an fp16 outlier k-tile: its B fragments are plain
    // global loads from oweight, in flight during this tile's MFMAs -- the outlier k-tiles ride the same activation ring
    // (their first version, with an empty pipeline around each, cost ~3 us per k-tile: 6 us of a 75 us launch).
    u32x4 qn;
    f16 sn_, zn_;
    auto fetch_b = [&](int slot) {
        qn = read_q(slot);
        sn_ = *(const f16*)(lds + s_rd + slot * G3_S);
        zn_ = *(const f16*)(lds + s_rd + slot * G3_S + 256);
    };
    auto tile_body = [&](auto tail_tag, int t, int slot_n2, const u32x4 (&bc)[4], u32x4 (&bn)[4]) {
        constexpr bool TAIL = decltype(tail_tag)::value;
        if (ABL == 2) return;
        const uint8_t* st = lds + (size_t)(t & (G3_ST - 1)) * A_B;
        const uint8_t* sn = lds + (size_t)((t + 1) & (G3_ST - 1)) * A_B;
        const bool has_next = !TAIL || t + 1 < ktiles;
        const u32x4 q = qn;
        const h2 sc = splat(sn_), zc = splat(zn_);
        h2 qx[4];
        if (TAIL && OUTL && has_next) {
            const u32x4* p = (const u32x4*)(ow + (size_t)ncol * n_out + ((kt0 + t + 1) * BK + h * 32 - kq));
#pragma unroll
            for (int j = 0; j < 4; ++j) bn[j] = p[j];
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            u32x4 (&cur)[4] = (mt & 1) ? fb : fa;
            u32x4 (&nxt)[4] = (mt & 1) ? fa : fb;
            if (mt == 0) {                  // behind the first MFMA: everything it waits for (lgkmcnt(0): hipcc does not count
                                            // across the loop edge) was issued before the barrier, not just now
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, cur[0]), __builtin_bit_cast(h8, bc[0]),
                                                               acc[0], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (ABL == 5) {
            } else if (mt < MT - 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) nxt[j] = *(const u32x4*)(st + a_rd[j] + (mt + 1) * 4096);
            } else if (has_next) {
#pragma unroll
                for (int j = 0; j < 4; ++j) nxt[j] = *(const u32x4*)(sn + a_rd[j]);
                if (!TAIL) fetch_b(slot_n2);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = mt == 0 ? 1 : 0; j < 4; ++j)
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, cur[j]), __builtin_bit_cast(h8, bc[j]),
                                                                acc[mt], 0, 0, 0);
            if (!TAIL && ABL != 4 && BITS == 3) {       // 16 pairs of k-tile t + 1 over the MT phases
                constexpr int PP = 16 / MT;
#pragma unroll
                for (int i = 0; i < PP; ++i) {
                    const int e = mt * PP + i;
                    bn[e >> 2][e & 3] = as_u32(__builtin_elementwise_fma(w3_pair_q(q[0], q[1], q[2], e, MAGIC3), sc, zc));
                }
            } else if (!TAIL && ABL != 4) { // word mt / 2 of k-tile t + 1: exact q in the even phase, one rounded FMA per weight in the odd
                if constexpr (MT == 8) {
                    if ((mt & 1) == 0) {
                        nib8_to_q(q[mt >> 1], qx);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) bn[j][mt >> 1] = as_u32(__builtin_elementwise_fma(qx[j], sc, zc));
                    }
                } else {                    // four phases per k-tile: a whole word per phase
                    nib8_to_q(q[mt], qx);
#pragma unroll
                    for (int j = 0; j < 4; ++j) bn[j][mt] = as_u32(__builtin_elementwise_fma(qx[j], sc, zc));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto sync_tile = [&]() {
        if (ABL < 3) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    long long tk0 = 0, tr0 = 0;                // ABL 6: shader-clock / 100 MHz timestamps around the k loop -> the "bias" buffer
    if (ABL == 6) { tk0 = clock64(); tr0 = wall_clock64(); }
    int slot_n2 = 2 % G3_BST;                  // weight slot of k-tile t + 2
    auto bump = [&]() { slot_n2 = slot_n2 + 1 == G3_BST ? 0 : slot_n2 + 1; };
    fetch_b(1 % G3_BST);
    {   // qtiles >= 4 (launcher).  An odd count runs one leading k-tile and moves its B back to bA, so that the loop keeps
        // static register roles; the loop leaves the last INT4 k-tile (its B in bB) to the tail.
        int t = 0;
        if (qtiles & 1) {
            sync_tile();
            tile_body(std::false_type{}, 0, slot_n2, bA, bB);
            bump();
#pragma unroll
            for (int j = 0; j < 4; ++j) bA[j] = bB[j];
            t = 1;
        }
        for (; t + 2 < qtiles; t += 2) {
            sync_tile();
            tile_body(std::false_type{}, t, slot_n2, bA, bB);
            bump();
            sync_tile();
            tile_body(std::false_type{}, t + 1, slot_n2, bB, bA);
            bump();
        }
        sync_tile();
        tile_body(std::false_type{}, t, slot_n2, bA, bB);
        for (t = qtiles - 1; t < ktiles; ++t) {        // the last INT4 k-tile, then the outlier k-tiles
            sync_tile();
            tile_body(std::true_type{}, t, slot_n2, bB, bA);
#pragma unroll
            for (int j = 0; j < 4; ++j) bB[j] = bA[j];
        }
    }

    if (ABL == 6) {
        const long long tk1 = clock64(), tr1 = wall_clock64();
        if (wave == 0 && lane == 0 && bias) {
            long long* d = (long long*)bias + (size_t)blockIdx.x * 8;
            d[0] = tk1 - tk0; d[1] = tr1 - tr0; d[2] = tr0; d[3] = tr1; d[4] = t_entry;
        }
    }
    // ---- epilogue: the fp16 tile goes through LDS (the activation ring is free now) so that the stores to y are whole 256-byte
    // row segments, 16 bytes per lane: the direct form (2-byte stores, 64 bytes per row per instruction) cost ~10 us per block.
    __builtin_amdgcn_s_barrier();              // every compute wave is done with the ring (the loader waves have exited)
    asm volatile("" ::: "memory");
    if (part != nullptr) {                     // split-K: the fp32 partial tile, through LDS, as whole 512-byte row segments
        constexpr int PP = G3_BN * 4 + 16;     // pitch of the fp32 tile in LDS (128 rows: 66 KB, 256 rows: 132 KB of the free rings)
        uint8_t* const colp = lds + nloc * 4;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                *(float*)(colp + (mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * PP) = acc[mt][e];
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int ch = lane & 31, n0 = bn0 + ch * 4;
        float* const pz = part + (size_t)zsp * M * N;
#pragma unroll 4
        for (int i = 0; i < 4 * MT; ++i) {     // wave w: rows w * 8 MT + 2 i + lane / 32
            const int row = wave * (8 * MT) + i * 2 + (lane >> 5), m = bm0 + row;
            if (m >= M || n0 >= N) continue;
            *(f32x4*)(pz + (size_t)m * N + n0) = *(const f32x4*)(lds + row * PP + ch * 16);
        }
        return;
    }
    // SILU: the gate segments of the 16 row groups this lane stores below, fetched four groups at a time, the first two
    // batches before the tile goes to LDS (clamped addresses, unconditional loads)
    u32x4 ga[4], gb[4];
    auto gate_load = [&](int grp, u32x4(&dst)[4]) {
        const int n0c = min(bn0 + (lane & 15) * 8, N - 8);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = min(bm0 + wave * 64 + (grp * 4 + j) * 4 + (lane >> 4), M - 1);
            dst[j] = *(const u32x4*)(gate + (size_t)m * N + n0c);
        }
    };
    if (EPI == 1) {
        gate_load(0, ga);
        gate_load(1, gb);
    }
    {
        const float bv = bias ? (float)bias[ncol] : 0.f;
        uint8_t* const col = lds + nloc * 2;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                *(f16*)(col + (mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * G3_YP) = (f16)(acc[mt][e] + bv);
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (EPI == 0) {
        const int ch = lane & 15, n0 = bn0 + ch * 8;
#pragma unroll 4
        for (int i = 0; i < 2 * MT; ++i) {
            const int row = wave * (8 * MT) + i * 4 + (lane >> 4), m = bm0 + row;
            if (m >= M || n0 >= N) continue;
            const u32x4 v = *(const u32x4*)(lds + row * G3_YP + ch * 16);
            f16* dst = y + (size_t)m * N + n0;
            if (n0 + 8 <= N && (N & 7) == 0) *(u32x4*)dst = v;
            else {
                const h8 hv = __builtin_bit_cast(h8, v);
                for (int j = 0; j < 8 && n0 + j < N; ++j) dst[j] = hv[j];
            }
        }
    } else if (EPI == 1) {                     // N % 8 == 0 (launcher)
        const int ch = lane & 15, n0 = bn0 + ch * 8;
        auto put = [&](int grp, const u32x4(&g)[4]) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = wave * 64 + (grp * 4 + j) * 4 + (lane >> 4), m = bm0 + row;
                if (m >= M || n0 >= N) continue;
                const h8 u = __builtin_bit_cast(h8, *(const u32x4*)(lds + row * G3_YP + ch * 16)), gv = __builtin_bit_cast(h8, g[j]);
                h8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = mul_f32_to_f16(silu_f32((float)gv[e]), (float)u[e]);
                *(h8*)(y + (size_t)m * N + n0) = o;
            }
        };
        put(0, ga);
        gate_load(2, ga);
        put(1, gb);
        gate_load(3, gb);
        put(2, ga);
        put(3, gb);
    } else {                                   // EPI 2: N % 128 == 0 (launcher); 8 lanes per row of 64 outputs
        const int ch = lane & 7, nh = N >> 1, n0 = (bn0 >> 1) + ch * 8;
#pragma unroll 4
        for (int i = 0; i < 8; ++i) {
            const int row = wave * 64 + i * 8 + (lane >> 3), m = bm0 + row;
            if (m >= M) continue;
            const h8 gv = __builtin_bit_cast(h8, *(const u32x4*)(lds + row * G3_YP + ch * 16));
            const h8 u = __builtin_bit_cast(h8, *(const u32x4*)(lds + row * G3_YP + 128 + ch * 16));
            h8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = mul_f32_to_f16(silu_f32((float)gv[e]), (float)u[e]);
            *(h8*)(y + (size_t)m * nh + n0) = o;
        }
    }
    if (ABL == 6) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (wave == 0 && lane == 0 && bias) ((long long*)bias)[(size_t)blockIdx.x * 8 + 5] = wall_clock64();
    }
}

// y[m][n] = fp16(sum_z part[z][m][n] + bias[n]); 4 outputs per thread (N % 4 == 0)
__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(const float* __restrict__ part, const f16* __restrict__ bias,
                                                                 f16* __restrict__ y, int M, int N, int S) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    const size_t total = (size_t)M * N;
    if (i >= total) return;
    f32x4 acc = *(const f32x4*)(part + i);
    for (int zz = 1; zz < S; ++zz) acc += *(const f32x4*)(part + (size_t)zz * total + i);
    const int n = (int)(i % N);
    h4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (f16)(acc[j] + (bias ? (float)bias[n + j] : 0.f));
    *(h4*)(y + i) = o;
}

// Split factor for a shape: 1 unless the 128x128 tiling leaves most of the 256 CUs idle and K is long enough to cut.
int gemm_w4_split(int M, int N, int K, int n_out) {
    const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    const int qtiles = (K - n_out) / BK;
    if (tiles >= 256 || qtiles < 16) return 1;
    int s = 512 / tiles;               // aim at ~2 blocks per CU
    if (s > qtiles / 8) s = qtiles / 8; // every split keeps >= 8 k-tiles: the DMA ring needs a few to pay off
    if (s > 16) s = 16;
    return s < 2 ? 1 : s;
}

// silu_gate == kSiluPair64: the paired-halves epilogue (EPI 2) instead of an external gate tensor
extern const void* const kSiluPair64 = (const void*)(uintptr_t)1;

// shapes the paired gate|up launch takes (n2 = both linears' rows): the 256-row kernel's domain, whole 128-column tiles
bool gemm_w4_pair64_ok(int M, int n2, int K, int G, int n_out) {
    return M >= 1 && n2 % 128 == 0 && K / BK >= G3_BST && K % BK == 0 && n_out % 64 == 0 && (K - n_out) / BK >= 2 &&
           (G & (G - 1)) == 0 && G >= 64 &&
           (size_t)M * K * 2 < (1ull << 32) && (size_t)(n2 / 4) * K * 2 < (1ull << 32);
}

// 256-row or 128-row loader-wave tile?  One block per CU either way; a launch takes ceil(tiles / 256) rounds of blocks, a partly
// filled round costs a whole one, and a 256-row block takes ~1.65x a 128-row block (it is the more efficient one per flop:
// profiles/r04_gemm_m_sweep.txt -- 52-57 us against 31-35 us per round at K = 4096).  The tile with the cheaper sum of rounds wins,
// ties go to the 256-row tile.  Round 4 dropped round 2's unconditional "M >= 1024 -> 256 rows": 11008 x 4096 at M = 1408 is 516
// tiles of 256 rows = THREE rounds (167 us) but 946 of 128 rows = four cheaper ones.
static bool gemm_v3_prefers_256(int M, int N) {
    const int nb = (N + G3_BN - 1) / G3_BN, t8 = ((M + G3_BM - 1) / G3_BM) * nb, t4 = ((M + 127) / 128) * nb;
    if (M <= 256 || t8 < 129) return false;
    return 1.65 * ((t8 + 255) / 256) <= 1.0 * ((t4 + 255) / 256);
}

// Split factor of the 128-row loader-wave tier (round 3): S blocks per tile when the tiles alone fill less than 3/4 of the CUs --
// the largest S <= 256 / tiles that divides the k-tile count, leaves every block >= 8 k-tiles and the last block >= 2 INT4 ones.
int gemm_v3_split(int M, int N, int K, int n_out) {
    const int tiles = ((M + 127) / 128) * ((N + G3_BN - 1) / G3_BN), kt = K / BK;
    if (tiles >= 192 || tiles < 1 || K % BK != 0) return 1;
    for (int s = 256 / tiles > 16 ? 16 : 256 / tiles; s >= 2; --s)
        if (kt % s == 0 && kt / s >= 8 && kt / s - n_out / BK >= 2) return s;
    return 1;
}

// The tiers that read the 3-bit extension layout directly (0: none -- expand first; 8 / 4: the 256- / 128-row loader-wave tile)
int gemm_w3_native_tile(int M, int N, int K, int G, int n_out) {
    const int nb = (N + G3_BN - 1) / G3_BN;
    const bool ok = K % 128 == 0 && n_out % 128 == 0 && n_out < K && (K - n_out) / BK >= 2 && K / BK >= G3_BST && N % 16 == 0 &&
                    (G & (G - 1)) == 0 && G >= 64 && (size_t)M * K * 2 < (1ull << 32) &&
                    (size_t)(N / 16) * ((K - n_out) / 128) * 768 < (1ull << 32);
    if (!ok) return 0;
    if (gemm_v3_prefers_256(M, N)) return 8;
    if (((M + 127) / 128) * nb >= 112 && M > 128) return 4;
    return 0;
}

hipError_t gemm_w4_launch(const void* x, const void* qw, const void* scales, const void* zeros, const void* ow,
                          const void* bias, void* y, int M, int N, int K, int G, int n_out, hipStream_t st,
                          void* workspace, size_t workspace_bytes, const void* silu_gate, bool* fused_epilogue, int bits) {
    const bool outl = ow && n_out > 0;
    if (fused_epilogue) *fused_epilogue = false;      // set by the tier that forms silu(gate) * y in its own epilogue
    if (bits == 3) {
        const int tile = silu_gate ? 0 : gemm_w3_native_tile(M, N, K, G, outl ? n_out : 0);
        if (!tile) return hipErrorNotSupported;
        const int nb = (N + G3_BN - 1) / G3_BN, mbt = (M + 32 * tile - 1) / (32 * tile);
        const int smem = G3_ST * 32 * tile * BK * 2 + G3_BST * G3_B + G3_BST * G3_S;
        auto go = [&](auto kern) -> hipError_t {
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(kern, dim3(mbt * nb), dim3(512), smem, st, (const f16*)x, (const uint8_t*)qw, (const f16*)scales,
                               (const f16*)zeros, (const f16*)(outl ? ow : nullptr), (const f16*)bias, (f16*)y, M, N, K, G,
                               outl ? n_out : 0, nb, (const f16*)nullptr, (float*)nullptr, 1);
            return hipGetLastError();
        };
        g_last_variant = tile == 8 ? "gemm_v3_256x128_w3" : "gemm_v3_128x128_w3";
        if (tile == 8) return outl ? go(gemm_w4_kernel_v3<true, 0, 0, 8, 3>) : go(gemm_w4_kernel_v3<false, 0, 0, 8, 3>);
        return outl ? go(gemm_w4_kernel_v3<true, 0, 0, 4, 3>) : go(gemm_w4_kernel_v3<false, 0, 0, 4, 3>);
    }
    // 256 x 128 tiles, one wave per SIMD (gemm_w4_kernel_v3), when they give (nearly) every CU a block: the M >= 2048 tier
    // of a prefill / fine-tune step.  QEFT_GEMM_V3 = 0 / 1 forces the choice (A/B).
    {
        static const int force_v3 = getenv("QEFT_GEMM_V3") ? atoi(getenv("QEFT_GEMM_V3")) : -1;
        const int mb = (M + G3_BM - 1) / G3_BM, nb = (N + G3_BN - 1) / G3_BN;
        // (at least two INT4 k-tiles: with exactly one, the odd-count prologue of the k loop would dequantise k-tile 1 -- an fp16
        //  outlier tile whose weight slot was never staged -- as INT4; such shapes keep the 128-row kernels)
        const bool ok3 = K / BK >= G3_BST && K % BK == 0 && (!outl || n_out % 64 == 0) && (K - (outl ? n_out : 0)) / BK >= 2 &&
                         (G & (G - 1)) == 0 && G >= 64 && N % 4 == 0 && N >= 2 && (size_t)M * K * 2 < (1ull << 32) && (size_t)(N / 4) * K * 2 < (1ull << 32);
        if (silu_gate == kSiluPair64 && !(ok3 && N % 128 == 0)) return hipErrorNotSupported;     // capi checks gemm_w4_pair64_ok first
        if (ok3 && (silu_gate == kSiluPair64 || force_v3 == 1 || (force_v3 != 0 && gemm_v3_prefers_256(M, N))) &&
            (!silu_gate || N % 8 == 0)) {
            auto go3 = [&](auto kern) -> hipError_t {
                hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G3_SMEM);
                if (e != hipSuccess) return e;
                hipLaunchKernelGGL(kern, dim3(mb * nb), dim3(512), G3_SMEM, st, (const f16*)x, (const uint8_t*)qw,
                                   (const f16*)scales, (const f16*)zeros, (const f16*)(outl ? ow : nullptr), (const f16*)bias,
                                   (f16*)y, M, N, K, G, outl ? n_out : 0, nb, (const f16*)silu_gate, (float*)nullptr, 1);
                return hipGetLastError();
            };
            if (silu_gate == kSiluPair64) {
                if (fused_epilogue) *fused_epilogue = true;
                g_last_variant = "gemm_v3_256x128+silu_pair";
                return outl ? go3(gemm_w4_kernel_v3<true, 0, 2>) : go3(gemm_w4_kernel_v3<false, 0, 2>);
            }
            if (silu_gate) {
                if (fused_epilogue) *fused_epilogue = true;
                g_last_variant = "gemm_v3_256x128+silu";
                return outl ? go3(gemm_w4_kernel_v3<true, 0, 1>) : go3(gemm_w4_kernel_v3<false, 0, 1>);
            }
            g_last_variant = "gemm_v3_256x128";
#ifdef QEFT_LAB      // lab builds only (QEFT_BUILD_LAB=1 python -m qeft_amd.build): ablations / time stamps, wrong results by design
            static const int abl = getenv("QEFT_GEMM_ABL") ? atoi(getenv("QEFT_GEMM_ABL")) : 0;
            if (abl == 1) return go3(gemm_w4_kernel_v3<true, 1>);
            if (abl == 2) return go3(gemm_w4_kernel_v3<true, 2>);
            if (abl == 3) return go3(gemm_w4_kernel_v3<true, 3>);
            if (abl == 4) return go3(gemm_w4_kernel_v3<true, 4>);
            if (abl == 5) return go3(gemm_w4_kernel_v3<true, 5>);
            if (abl == 6) return go3(gemm_w4_kernel_v3<true, 6>);
#endif
            return outl ? go3(gemm_w4_kernel_v3<true>) : go3(gemm_w4_kernel_v3<false>);
        }
        // The same design with 128-row tiles (round 3) for the sizes between the small-M kernels and that tier -- M = 512 .. 1024
        // of a short prompt or a fine-tune batch, narrow N at larger M: taken from 112 tiles of 128 x 128 on (measured crossover
        // against the 128-row kernels below, profiles/r03_gemm_mid_m.txt: 4096 x 4096 at M = 512, 128 tiles: 28.4 vs 34.6 us;
        // 11008 x 4096 at M = 256, 172 tiles: 30.4 vs 43.6; 64 tiles lose).  QEFT_GEMM_V3M = 0 / 1 forces the choice.
        {
            static const int force_m = getenv("QEFT_GEMM_V3M") ? atoi(getenv("QEFT_GEMM_V3M")) : -1;
            const int mb4 = (M + 127) / 128;
            // fewer than 192 tiles and a workspace: S blocks per tile, fp32 partials, the reduce launch (QEFT_GEMM_V3S=0 disables)
            static const int force_s = getenv("QEFT_GEMM_V3S") ? atoi(getenv("QEFT_GEMM_V3S")) : -1;
            const int S3 = (force_s == 0 || force_m == 0 || !workspace || silu_gate || !ok3) ? 1 : gemm_v3_split(M, N, K, outl ? n_out : 0);
            if (S3 > 1 && workspace_bytes >= (size_t)S3 * M * N * 4) {
                constexpr int SMEM4 = G3_ST * 128 * BK * 2 + G3_BST * G3_B + G3_BST * G3_S;
                auto go4 = [&](auto kern) -> hipError_t {
                    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM4);
                    if (e != hipSuccess) return e;
                    hipLaunchKernelGGL(kern, dim3(mb4 * nb * S3), dim3(512), SMEM4, st, (const f16*)x, (const uint8_t*)qw,
                                       (const f16*)scales, (const f16*)zeros, (const f16*)(outl ? ow : nullptr), (const f16*)nullptr,
                                       (f16*)y, M, N, K, G, outl ? n_out : 0, nb, (const f16*)nullptr, (float*)workspace, S3);
                    return hipGetLastError();
                };
                g_last_variant = "gemm_v3_128x128+splitk";
                hipError_t e = outl ? go4(gemm_w4_kernel_v3<true, 0, 0, 4>) : go4(gemm_w4_kernel_v3<false, 0, 0, 4>);
                if (e != hipSuccess) return e;
                const size_t quads = (size_t)M * N / 4;
                hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((int)((quads + 255) / 256)), dim3(256), 0, st,
                                   (const float*)workspace, (const f16*)bias, (f16*)y, M, N, S3);
                return hipGetLastError();
            }
            if (ok3 && !silu_gate && (force_m == 1 || (force_m != 0 && mb4 * nb >= 112 && M > 128))) {
                constexpr int SMEM4 = G3_ST * 128 * BK * 2 + G3_BST * G3_B + G3_BST * G3_S;         // 93184 bytes
                auto go4 = [&](auto kern) -> hipError_t {
                    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM4);
                    if (e != hipSuccess) return e;
                    hipLaunchKernelGGL(kern, dim3(mb4 * nb), dim3(512), SMEM4, st, (const f16*)x, (const uint8_t*)qw,
                                       (const f16*)scales, (const f16*)zeros, (const f16*)(outl ? ow : nullptr), (const f16*)bias,
                                       (f16*)y, M, N, K, G, outl ? n_out : 0, nb, (const f16*)nullptr, (float*)nullptr, 1);
                    return hipGetLastError();
                };
                g_last_variant = "gemm_v3_128x128";
                return outl ? go4(gemm_w4_kernel_v3<true, 0, 0, 4>) : go4(gemm_w4_kernel_v3<false, 0, 0, 4>);
            }
        }
    }
    // 128 x 256 tiles (8 waves) when they still give every CU a block: always a gain where the 128 x 128 kernel's scale
    // array leaves room for one block per CU only (K = 11008: 747 -> 945 TFLOP/s at M = 2048), a few % otherwise at
    // M >= 2048; smaller problems keep the 128 x 128 tile (more blocks).  QEFT_GEMM_NWV = 4 / 8 forces one (A/B).
    static const int force_nwv = getenv("QEFT_GEMM_NWV") ? atoi(getenv("QEFT_GEMM_NWV")) : 0;
    int nwv = 4;
    const bool one_block4 = gemm_v2_smem(K, G, 4) > 80 * 1024;
    if (force_nwv == 8 || (force_nwv == 0 && N % 256 == 0 && ((M + BM - 1) / BM) * (N / 256) >= 256 && (one_block4 || M >= 2048)))
        nwv = 8;
    if (gemm_v2_smem(K, G, nwv) > 160 * 1024) nwv = 4;
    const int bn = 32 * nwv;
    dim3 grid((M + BM - 1) / BM, (N + bn - 1) / bn);
    int S = workspace ? gemm_w4_split(M, N, K, outl ? n_out : 0) : 1;
    if (nwv == 8) S = 1;
    while (S > 1 && (size_t)S * M * N * 4 > workspace_bytes) --S;
    if (N % 4 != 0) S = 1;
    const size_t smem2 = gemm_v2_smem(K, G, nwv);
    if (K / BK >= kStages && (!outl || n_out % 64 == 0) && (G & (G - 1)) == 0 && smem2 <= 160 * 1024 &&
        getenv("QEFT_GEMM_V1") == nullptr) {
        auto go = [&](auto kern) -> hipError_t {
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem2);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(kern, dim3(grid.x, grid.y, S), dim3(64 * nwv), smem2, st, (const f16*)x, (const uint8_t*)qw,
                               (const f16*)scales, (const f16*)zeros, (const f16*)(outl ? ow : nullptr), (const f16*)bias,
                               (f16*)y, M, N, K, G, outl ? n_out : 0, (float*)workspace);
            return hipGetLastError();
        };
        hipError_t e;
        if (nwv == 8) e = outl ? go(gemm_w4_kernel_v2<true, 8>) : go(gemm_w4_kernel_v2<false, 8>);
        else e = outl ? go(gemm_w4_kernel_v2<true, 4>) : go(gemm_w4_kernel_v2<false, 4>);
        if (e != hipSuccess) return e;
        g_last_variant = nwv == 8 ? "gemm_v2_128x256" : (S > 1 ? "gemm_v2_128x128+splitk" : "gemm_v2_128x128");
        if (S > 1) {
            const size_t quads = (size_t)M * N / 4;
            hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((int)((quads + 255) / 256)), dim3(256), 0, st,
                               (const float*)workspace, (const f16*)bias, (f16*)y, M, N, S);
        }
        return hipGetLastError();
    }
    grid = dim3((M + BM - 1) / BM, (N + BN - 1) / BN);
    g_last_variant = "gemm_v1";
    if (ow && n_out > 0)
        hipLaunchKernelGGL(gemm_w4_kernel<true>, grid, dim3(GEMM_THREADS), 0, st, (const f16*)x, (const uint8_t*)qw,
                           (const f16*)scales, (const f16*)zeros, (const f16*)ow, (const f16*)bias, (f16*)y, M, N, K, G,
                           n_out);
    else
        hipLaunchKernelGGL(gemm_w4_kernel<false>, grid, dim3(GEMM_THREADS), 0, st, (const f16*)x, (const uint8_t*)qw,
                           (const f16*)scales, (const f16*)zeros, (const f16*)nullptr, (const f16*)bias, (f16*)y, M, N,
                           K, G, 0);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Backward wrt input: dx[M,K] = dy[M,N] . Wdeq[N,K]  (contraction over n).
// The contraction index is the packed layout's ROW index: the B fragment of a lane is 8 consecutive n at ONE k, i.e.
// a column of the weight tile.  The tile is dequantised into LDS row-major ([n][k]: two 16-byte writes per thread) and
// read back through gfx950's transposing LDS read (ds_read_b64_tr_b16), two reads per fragment.
// Block tile: 128 (m) x 64 (k);  n advances 64 per stage; global loads of the next stage are in flight during the MFMAs.
// ---------------------------------------------------------------------------------------------------
constexpr int DX_BM = 128, DX_BK = 64, DX_BN = 64;
constexpr int WT_PITCH = 144;          // bytes per n row of the weight tile (64 k x 2 B + 16): 16-byte aligned, banks spread

__global__ __launch_bounds__(256) void gemm_w4_dx_kernel(const f16* __restrict__ dy, const uint8_t* __restrict__ qw,
                                                         const f16* __restrict__ scales, const f16* __restrict__ zeros,
                                                         const f16* __restrict__ ow, f16* __restrict__ dx, int M, int N,
                                                         int K, int G, int n_out, float* __restrict__ part) {
    // gridDim.z = S > 1 (few output tiles, long contraction): block z contracts the n-tiles [nt*z/S, nt*(z+1)/S) and
    // writes an fp32 partial tile to part[z][M][K]; gemm_splitk_reduce_kernel sums them in order.
    __shared__ __attribute__((aligned(16))) uint8_t lds_a[DX_BM * DX_BN * 2];        // dy tile [128][64], swizzled slots
    __shared__ __attribute__((aligned(16))) uint8_t lds_w[DX_BN * WT_PITCH];         // W tile [64 n][64 k (+8)] fp16

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wk = wave & 1;  // wave tile: 64 (m) x 32 (k)
    const int bm0 = blockIdx.x * DX_BM, kt = blockIdx.y;  // kt: 64-k tile index
    const int kq = K - n_out;

    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

    const int arow = tid >> 1, ach = tid & 1;
    const bool arow_ok = bm0 + arow < M;

    // W staging role: thread -> (n_local = tid>>2, chunk = (tid>>1)&1, half = tid&1): 16 of the 32 k of a chunk
    const int wn_l = tid >> 2, wch = (tid >> 1) & 1, whalf = tid & 1;

    // Two-stage software pipeline in registers: the global loads of tile n0 + 64 are issued before the MFMAs of tile
    // n0, so HBM/L2 latency overlaps the matrix work instead of sitting between two barriers.
    const int k0 = kt * 64 + wch * 32;
    const bool outl_tile = ow != nullptr && k0 >= kq;          // block-uniform per (kt, wch) pair of the thread
    struct Stage {
        u32x4 a[4];      // dy: 64 bytes of row arow
        u32x4 w[2];      // packed nibbles (w[0]) or 16 fp16 outlier weights
        uint32_t sz;     // scale | scaled zero << 16
    };
    auto gload = [&](int n0, Stage& st) {
        const f16* p = dy + (size_t)(bm0 + arow) * N + n0 + ach * 32;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = arow_ok && (n0 + ach * 32 + j * 8) < N;
            st.a[j] = ok ? ((const u32x4*)p)[j] : u32x4{0u, 0u, 0u, 0u};
        }
        const int n = min(n0 + wn_l, N - 1);
        if (outl_tile) {
            const u32x4* po = (const u32x4*)(ow + (size_t)n * n_out + (k0 - kq) + whalf * 16);
            st.w[0] = po[0];
            st.w[1] = po[1];
            st.sz = 0;
        } else {
            st.w[0] = *(const u32x4*)(qw + (size_t)(n >> 2) * K * 2 + (size_t)kt * 128 + (n & 3) * 32 + wch * 16);
            st.w[1] = st.w[0];
            const int g = k0 / G;
            st.sz = (uint32_t)((const uint16_t*)scales)[(size_t)g * N + n] | ((uint32_t)((const uint16_t*)zeros)[(size_t)g * N + n] << 16);
        }
    };
    auto lstore = [&](int n0, const Stage& st) {
#pragma unroll
        for (int j = 0; j < 4; ++j) *(u32x4*)(lds_a + a_slot_off(arow, ach * 4 + j)) = st.a[j];
        // this thread owns k_local = whalf*16 .. +15 of chunk wch of row wn_l: two runs of 8 consecutive k
        u32x4 run[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
        if (n0 + wn_l < N) {
            if (outl_tile) {
                run[0] = st.w[0];
                run[1] = st.w[1];
            } else {
                const h2 szp = as_h2(st.sz);
                const h2 sc = splat(szp[0]), zc = splat(szp[1]);
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    h2 wd[4];
                    dequant8(st.w[0][w], sc, zc, wd);            // wd[j] = pair (k = 8j + 2w, +1)
                    run[0][w] = as_u32(whalf ? wd[2] : wd[0]);
                    run[1][w] = as_u32(whalf ? wd[3] : wd[1]);
                }
            }
        }
        uint8_t* wrow = lds_w + wn_l * WT_PITCH + (wch * 32 + whalf * 16) * 2;
        *(u32x4*)wrow = run[0];
        *(u32x4*)(wrow + 16) = run[1];
    };

    // transposed-read address of this lane (step 0): group g = lane >> 4 -> columns wk*32 + (g & 1)*16 .., rows 8*(g >> 1) ..
    const uint32_t wbase_tr = (uint32_t)(uintptr_t)lds_w +
        (uint32_t)(8 * (lane >> 5) + ((lane & 15) >> 2)) * WT_PITCH + (uint32_t)(wk * 32 + ((lane >> 4) & 1) * 16 + (lane & 3) * 4) * 2;
    const int ntiles = (N + DX_BN - 1) / DX_BN, S = gridDim.z, z = blockIdx.z;
    const int n_lo = (int)((long long)ntiles * z / S) * DX_BN, n_hi = min((int)((long long)ntiles * (z + 1) / S) * DX_BN, N);
    Stage cur, nxt;
    gload(n_lo, cur);
    for (int n0 = n_lo; n0 < n_hi; n0 += DX_BN) {
        lstore(n0, cur);
        __syncthreads();
        if (n0 + DX_BN < n_hi) gload(n0 + DX_BN, nxt);
        // ---- MFMA: 4 n-steps of 16.  B fragment of lane (k col = wk*32 + r, h): n = s*16 + 8h .. +7 -- a column of
        // the [n][k] tile: two transposed reads (rows +0..3 and +4..7); lane 4q+p of its 16-lane group addresses
        // row q, columns 4p..  All 8 reads go out first, one wait.
        u32x2 bt[8];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bt[2 * s] = lds_read_tr8(wbase_tr + (uint32_t)(s * 16) * WT_PITCH);
            bt[2 * s + 1] = lds_read_tr8(wbase_tr + (uint32_t)(s * 16 + 4) * WT_PITCH);
        }
        lds_wait();
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const h8 bf = __builtin_bit_cast(h8, u32x4{bt[2 * s][0], bt[2 * s][1], bt[2 * s + 1][0], bt[2 * s + 1][1]});
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int row = wm * 64 + mt * 32 + r;
                // A fragment: n = s*16 + 8h .. +7  -> chunk (s>>1), slot ((s&1)*2 + h)
                const h8 af = *(const h8*)(lds_a + a_slot_off(row, (s >> 1) * 4 + (s & 1) * 2 + h));
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, bf, acc[mt], 0, 0, 0);
            }
        }
        __syncthreads();
        cur = nxt;
    }

    const int kcol = kt * 64 + wk * 32 + r;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = bm0 + wm * 64 + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (m < M) {
                if (S > 1) part[((size_t)z * M + m) * K + kcol] = acc[mt][e];
                else dx[(size_t)m * K + kcol] = (f16)acc[mt][e];
            }
        }
}

// Same algorithm with a 128 (m) x 128 (k) block tile (K % 128 == 0): every dy tile feeds twice the MFMAs, a thread
// stages exactly one (row, 32-k chunk) = 16 packed bytes -> four 16-byte LDS writes.  4 waves as 2 (m) x 2 (k), each 64 x 64.
constexpr int WT_PITCH2 = 272;         // bytes per n row of the [64 n][128 k] weight tile (256 + 16)

__global__ __launch_bounds__(256) void gemm_w4_dx128_kernel(const f16* __restrict__ dy, const uint8_t* __restrict__ qw,
                                                            const f16* __restrict__ scales, const f16* __restrict__ zeros,
                                                            const f16* __restrict__ ow, f16* __restrict__ dx, int M, int N,
                                                            int K, int G, int n_out) {
    __shared__ __attribute__((aligned(16))) uint8_t lds_a[DX_BM * DX_BN * 2];        // dy tile [128][64], swizzled slots
    __shared__ __attribute__((aligned(16))) uint8_t lds_w[DX_BN * WT_PITCH2];        // W tile [64 n][128 k (+8)] fp16

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wk = wave & 1;  // wave tile: 64 (m) x 64 (k)
    const int bm0 = blockIdx.x * DX_BM, kt = blockIdx.y;  // kt: 128-k tile index
    const int kq = K - n_out;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int arow = tid >> 1, ach = tid & 1;
    const bool arow_ok = bm0 + arow < M;
    const int wn_l = tid >> 2, wch = tid & 3;             // W staging role: row n_local, 32-k chunk of the 128
    const int k0 = kt * 128 + wch * 32;
    const bool outl_chunk = ow != nullptr && k0 >= kq;

    struct Stage {
        u32x4 a[4];      // dy: 64 bytes of row arow
        u32x4 w[4];      // packed nibbles (w[0]) or 32 fp16 outlier weights
        uint32_t sz;     // scale | scaled zero << 16
    };
    auto gload = [&](int n0, Stage& st) {
        const f16* p = dy + (size_t)(bm0 + arow) * N + n0 + ach * 32;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = arow_ok && (n0 + ach * 32 + j * 8) < N;
            st.a[j] = ok ? ((const u32x4*)p)[j] : u32x4{0u, 0u, 0u, 0u};
        }
        const int n = min(n0 + wn_l, N - 1);
        if (outl_chunk) {
            const u32x4* po = (const u32x4*)(ow + (size_t)n * n_out + (k0 - kq));
#pragma unroll
            for (int j = 0; j < 4; ++j) st.w[j] = po[j];
            st.sz = 0;
        } else {
            st.w[0] = *(const u32x4*)(qw + (size_t)(n >> 2) * K * 2 + (size_t)(k0 >> 6) * 128 + (n & 3) * 32 + ((k0 >> 5) & 1) * 16);
            const int g = k0 / G;
            st.sz = (uint32_t)((const uint16_t*)scales)[(size_t)g * N + n] | ((uint32_t)((const uint16_t*)zeros)[(size_t)g * N + n] << 16);
        }
    };
    auto lstore = [&](int n0, const Stage& st) {
#pragma unroll
        for (int j = 0; j < 4; ++j) *(u32x4*)(lds_a + a_slot_off(arow, ach * 4 + j)) = st.a[j];
        u32x4 run[4];    // run j = the 8 consecutive k 8j .. 8j+7 of the chunk
        if (n0 + wn_l >= N) {
#pragma unroll
            for (int j = 0; j < 4; ++j) run[j] = u32x4{0u, 0u, 0u, 0u};
        } else if (outl_chunk) {
#pragma unroll
            for (int j = 0; j < 4; ++j) run[j] = st.w[j];
        } else {
            const h2 szp = as_h2(st.sz);
            const h2 sc = splat(szp[0]), zc = splat(szp[1]);
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                h2 wd[4];
                dequant8(st.w[0][w], sc, zc, wd);            // wd[j] = pair (k = 8j + 2w, +1)
#pragma unroll
                for (int j = 0; j < 4; ++j) run[j][w] = as_u32(wd[j]);
            }
        }
        uint8_t* wrow = lds_w + wn_l * WT_PITCH2 + wch * 64;
#pragma unroll
        for (int j = 0; j < 4; ++j) *(u32x4*)(wrow + 16 * j) = run[j];
    };

    // transposed-read address of this lane for k-half 0, n-step 0 (see gemm_w4_dx_kernel)
    const uint32_t wbase_tr = (uint32_t)(uintptr_t)lds_w +
        (uint32_t)(8 * (lane >> 5) + ((lane & 15) >> 2)) * WT_PITCH2 + (uint32_t)(wk * 64 + ((lane >> 4) & 1) * 16 + (lane & 3) * 4) * 2;
    Stage cur, nxt;
    gload(0, cur);
    for (int n0 = 0; n0 < N; n0 += DX_BN) {
        lstore(n0, cur);
        __syncthreads();
        if (n0 + DX_BN < N) gload(n0 + DX_BN, nxt);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            u32x2 bt[4];
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                bt[2 * kh] = lds_read_tr8(wbase_tr + (uint32_t)(s * 16) * WT_PITCH2 + kh * 64);
                bt[2 * kh + 1] = lds_read_tr8(wbase_tr + (uint32_t)(s * 16 + 4) * WT_PITCH2 + kh * 64);
            }
            h8 af[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                af[mt] = *(const h8*)(lds_a + a_slot_off(wm * 64 + mt * 32 + r, (s >> 1) * 4 + (s & 1) * 2 + h));
            lds_wait();
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                const h8 bf = __builtin_bit_cast(h8, u32x4{bt[2 * kh][0], bt[2 * kh][1], bt[2 * kh + 1][0], bt[2 * kh + 1][1]});
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    acc[mt][kh] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[mt], bf, acc[mt][kh], 0, 0, 0);
            }
        }
        __syncthreads();
        cur = nxt;
    }

#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            const int kcol = kt * 128 + wk * 64 + kh * 32 + r;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = bm0 + wm * 64 + mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (m < M) dx[(size_t)m * K + kcol] = (f16)acc[mt][kh][e];
            }
        }
}

// ---------------------------------------------------------------------------------------------------
// dX for the M >= 2048 tier (a fine-tune step): the forward v3 block turned around.  Block tile 256 (m) x 128 (k), n advances
// 64 per tile; 4 compute waves (wave w: output columns 32 w .. 32 w + 31 of the tile, all 256 rows, 32 MFMAs per n-tile) and
// 4 loader waves, one of each per SIMD.
//   * dy tiles [256][64 n] ride a 4-stage LDS-DMA ring exactly like the forward's activations (same swizzle, same fragments).
//   * the contraction index n is the packed layout's ROW index, so the B fragment of a lane is a column of the weight tile:
//     the loader waves dequantise the tile [64 n][128 k] into LDS as fp16 (a lane owns one (n, 32-k chunk) = 16 packed
//     bytes -> four 16-byte stores, 16-byte slots XOR-swizzled by n & 3 so that both the stores and the transposing reads
//     are conflict-free) and the compute waves fetch their fragments with ds_read_b64_tr_b16 during the second half of the
//     previous tile: two barriers per n-tile, A(t) "dy tile t + 1 landed / tile t - 1 released", B(t) "W tile t + 1 is in
//     LDS".  The compute waves issue nothing but LDS reads and MFMAs.
//   * the packed weights never touch LDS: each loader lane keeps a 4-deep ring of (16 packed bytes, scale, zero) in
//     REGISTERS, loaded by inline-asm global loads that share the hand-counted vmcnt queue with the DMA pieces.
//   * the block of the fp16 outlier k-tile (k >= K - n_out; n_out = 128 = one tile) runs the same pipeline with 64 fp16
//     bytes per lane instead of 16 packed ones and no dequantisation -- it must not be the slow block of a one-wave grid.
// ---------------------------------------------------------------------------------------------------
constexpr int D3_BM = 256, D3_BK = 128, D3_BN = 64, D3_ST = 4;
constexpr int D3_A = D3_BM * D3_BN * 2, D3_W = D3_BN * D3_BK * 2;        // 32 KB dy stage, 16 KB fp16 weight tile
constexpr int D3_WOFF = D3_ST * D3_A;
constexpr size_t D3_SMEM = (size_t)D3_WOFF + 2 * D3_W;                  // 163840 bytes: the whole LDS of a CU

// The loader lanes' ring of weights in flight lives in ACCUMULATION registers named in the asm text (set S = a[8S .. 8S+5]:
// 16 packed bytes, scale, zero).  The loads return long after the statement that issued them; a value hipcc could see would
// be fair game for a register copy before the hand-placed s_waitcnt (it did exactly that at a switch: v_mov of registers
// whose loads were still in flight) -- registers it never allocates cannot be copied.
template <int S>
__device__ __forceinline__ void d3_pload(uint32_t qoff, const void* qb, uint32_t soff, const void* sb, const void* zb);
template <int S>
__device__ __forceinline__ void d3_pload3(uint32_t qoff, const void* qb, uint32_t soff, const void* sb, const void* zb);   // 3-bit records: 12 bytes
template <int S>
__device__ __forceinline__ void d3_pread(u32x4& q, uint32_t& sv, uint32_t& zv);
#define D3_RING_SET(S, A0, A1, A2, A3, A4, A5)                                                                                  \
    template <>                                                                                                                 \
    __device__ __forceinline__ void d3_pload<S>(uint32_t qoff, const void* qb, uint32_t soff, const void* sb, const void* zb) { \
        asm volatile("global_load_dwordx4 a[" #A0 ":" #A3 "], %0, %1\n\tglobal_load_ushort a" #A4 ", %2, %3\n\t"               \
                     "global_load_ushort a" #A5 ", %2, %4"                                                                      \
                     :: "v"(qoff), "s"(qb), "v"(soff), "s"(sb), "s"(zb)                                                         \
                     : "memory", "a" #A0, "a" #A1, "a" #A2, "a" #A3, "a" #A4, "a" #A5);                                         \
    }                                                                                                                           \
    template <>                                                                                                                 \
    __device__ __forceinline__ void d3_pload3<S>(uint32_t qoff, const void* qb, uint32_t soff, const void* sb, const void* zb) { \
        asm volatile("global_load_dwordx3 a[" #A0 ":" #A2 "], %0, %1\n\tglobal_load_ushort a" #A4 ", %2, %3\n\t"               \
                     "global_load_ushort a" #A5 ", %2, %4"                                                                      \
                     :: "v"(qoff), "s"(qb), "v"(soff), "s"(sb), "s"(zb)                                                         \
                     : "memory", "a" #A0, "a" #A1, "a" #A2, "a" #A3, "a" #A4, "a" #A5);                                         \
    }                                                                                                                           \
    template <>                                                                                                                 \
    __device__ __forceinline__ void d3_pread<S>(u32x4& q, uint32_t& sv, uint32_t& zv) {                                         \
        uint32_t q0, q1, q2, q3;                                                                                                \
        asm volatile("v_accvgpr_read_b32 %0, a" #A0 "\n\tv_accvgpr_read_b32 %1, a" #A1 "\n\tv_accvgpr_read_b32 %2, a" #A2 "\n\t" \
                     "v_accvgpr_read_b32 %3, a" #A3 "\n\tv_accvgpr_read_b32 %4, a" #A4 "\n\tv_accvgpr_read_b32 %5, a" #A5      \
                     : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3), "=v"(sv), "=v"(zv) :: "memory");                                 \
        q = u32x4{q0, q1, q2, q3};                                                                                              \
    }
D3_RING_SET(0, 0, 1, 2, 3, 4, 5)               // tuples start on even registers
D3_RING_SET(1, 8, 9, 10, 11, 12, 13)
D3_RING_SET(2, 16, 17, 18, 19, 20, 21)
D3_RING_SET(3, 24, 25, 26, 27, 28, 29)
#undef D3_RING_SET

// BITS = 3 (round 3): qw is the 3-bit extension layout; a loader lane owns the 12-byte record (row, 32-k chunk) of its tile
// position and unpacks it into the same four 16-byte fp16 runs -- the dX of a 3-bit layer without an expansion pass.
// MT = 4 (round 3): 128 (m) x 128 (k) tiles -- the same kernel for the sizes where 256-row tiles leave the chip short of blocks
// (M = 1024 on K = 4096: 128 tiles of 256 rows against 256 of 128); a 16 KB dy stage, 4 DMA pieces per loader wave and stage,
// barrier B after 2 of the 4 m-tile phases, two n-steps of the next tile's W fragments in each of the last two.
template <int BITS = 4, int MT = 8>
__global__ __launch_bounds__(512) void gemm_w4_dx_kernel_v3(const f16* __restrict__ dy, const uint8_t* __restrict__ qw,
                                                              const f16* __restrict__ scales, const f16* __restrict__ zeros,
                                                              const f16* __restrict__ ow, f16* __restrict__ dx, int M, int N,
                                                              int K, int G, int n_out, int NB) {
    static_assert(MT == 8 || MT == 4, "256- or 128-row tiles");
    constexpr int BMR = 32 * MT, AP = MT;                         // rows of the tile; dy pieces (1 KB) per loader wave and stage
    constexpr int A_B = BMR * D3_BN * 2, WOFF = D3_ST * A_B;      // bytes of a dy stage; the two 16 KB fp16 weight tiles behind the ring
    extern __shared__ __attribute__((aligned(1024))) uint8_t lds[];
    const uint32_t lds0 = (uint32_t)(uintptr_t)lds;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-contiguous block order (blocks b, b + 8, .. share an L2): an XCD works through the k-tiles of one row block of dy
    const int nblk = gridDim.x, bq = nblk >> 3, br = nblk & 7, bx = blockIdx.x & 7;
    const int c = (bx < br ? bx * (bq + 1) : br * (bq + 1) + (bx - br) * bq) + (blockIdx.x >> 3);
    const int bm0 = (c / NB) * BMR, kt = c % NB;
    const int ntiles = N / D3_BN;              // >= 4 (launcher)
    const int kq = K - n_out;
    const bool outl_blk = kt * D3_BK >= kq;    // n_out % 128 == 0: a k-tile is INT4 or fp16 as a whole

    if (wave >= 4) {
        // =========================================================== loader waves
        const int l = wave - 4;
        uint32_t a_off[AP];        // dy piece p = 8 rows x 128 B: row 8p + lane/8, LDS chunk lane%8 holds global chunk (lane%8) ^ ((row >> 1) & 7)
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const int row = (l * AP + i) * 8 + (lane >> 3);
            const int grow = min(bm0 + row, M - 1);
            a_off[i] = (uint32_t)grow * (uint32_t)N * 2u + (uint32_t)(((lane & 7) ^ ((row >> 1) & 7)) << 4);
        }
        auto stage_a = [&](int t) {
            g3_dma_a((const uint8_t*)dy + (size_t)t * (D3_BN * 2), a_off, lds0 + (uint32_t)(t & (D3_ST - 1)) * A_B + (uint32_t)l * (AP * 1024u));
        };
        // weight role of the lane: row group 4 l + lane/16 of the tile's 16, piece p = lane % 16 of its 256 contiguous bytes
        // = row n & 3 = (p & 7) >> 1, 32-k chunk kc = 2 (p >> 3) + (p & 1) of the 128-k tile
        // (3 bits: wave l takes row set l of the tile's four, lane = (chunk lane / 16, row lane % 16): 768 contiguous bytes per wave)
        const int rgl = l * 4 + (lane >> 4), pp = lane & 15;
        const int n_l = BITS == 3 ? l * 16 + (lane & 15) : rgl * 4 + ((pp & 7) >> 1), n3 = n_l & 3;
        const int kc = BITS == 3 ? (lane >> 4) : 2 * (pp >> 3) + (pp & 1);
        uint32_t w_dst[4];         // the lane's four 16-byte slots in the fp16 tile (+ buffer offset)
#pragma unroll
        for (int j = 0; j < 4; ++j) w_dst[j] = (uint32_t)(WOFF + n_l * 256 + (((kc * 4 + j) ^ (n3 << 2) ^ n3) << 4));
        auto sync = [&]() {
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        };

        if (!outl_blk) {
            const uint32_t set_b3 = (uint32_t)(kq / 128) * 768u;         // bytes of one 16-row set of the 3-bit stream
            const uint32_t q_off = BITS == 3 ? (uint32_t)l * set_b3 + (uint32_t)lane * 12u : (uint32_t)rgl * (uint32_t)K * 2u + (uint32_t)pp * 16u;
            const uint32_t s_off = (uint32_t)n_l * 2u;
            const size_t grp = (size_t)((kt * D3_BK) / G) * N;
            // ring set S (tile i lives in set i % 4) <- the lane's 16 packed bytes, scale and zero of tile t: 3 vector-memory operations
            auto pload = [&](auto set_tag, int t) {
                t = min(t, ntiles - 1);            // past the end: a harmless reload, the queue depth stays the same
                if constexpr (BITS == 3)
                    d3_pload3<decltype(set_tag)::value>(q_off, qw + (size_t)t * 4 * set_b3 + (size_t)kt * 768, s_off,
                                                        scales + grp + (size_t)t * D3_BN, zeros + grp + (size_t)t * D3_BN);
                else
                    d3_pload<decltype(set_tag)::value>(q_off, qw + (size_t)t * 16 * K * 2 + (size_t)kt * 256, s_off,
                                                       scales + grp + (size_t)t * D3_BN, zeros + grp + (size_t)t * D3_BN);
            };
            auto dequant_store = [&](auto set_tag, int buf) {
                u32x4 q;
                uint32_t sv, zv;
                d3_pread<decltype(set_tag)::value>(q, sv, zv);
                const h2 sc = splat(as_h2(sv)[0]), zc = splat(as_h2(zv)[0]);
                u32x4 run[4];      // run j = the 8 consecutive k 8j .. 8j + 7 of the chunk
                if constexpr (BITS == 3) {
                    uint32_t MAGIC3 = 0x64006400u;
                    asm volatile("" : "+v"(MAGIC3));
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        run[e >> 2][e & 3] = as_u32(__builtin_elementwise_fma(w3_pair_q(q[0], q[1], q[2], e, MAGIC3), sc, zc));
                } else {
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        h2 wd[4];
                        dequant8(q[w], sc, zc, wd);
#pragma unroll
                        for (int j = 0; j < 4; ++j) run[j][w] = as_u32(wd[j]);
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) *(u32x4*)(lds + w_dst[j] + buf * D3_W) = run[j];
            };
            // iteration t: [wait] A(t) [loads of tile t + 4] [dequantise tile t + 1] B(t) [dy pieces of tile t + 3].
            // In-order completion: the dy pieces of tile t + 1 were issued by iteration t - 2 (after that iteration's weight
            // loads); younger than them is exactly iteration t - 1's 3 + 8 operations; the weights of tile t + 1 (iteration
            // t - 3) are older still, so the one counted wait covers both.
            using S0 = std::integral_constant<int, 0>;
            using S1 = std::integral_constant<int, 1>;
            using S2 = std::integral_constant<int, 2>;
            using S3 = std::integral_constant<int, 3>;
            auto steady = [&](int t, auto use, auto fill) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 + AP) : "memory");
                sync();
                pload(fill, t + 4);
                dequant_store(use, (t + 1) & 1);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                sync();
                stage_a(t + 3);        // behind B(t): a DMA instruction holds the wave 100-200 cycles at issue, and the compute
                                       // waves reach B(t) after half a tile (~500 cycles)
            };
            // the last three tiles: nothing left to issue.  first: the step before it was a steady one (11 younger operations)
            auto tail = [&](bool first, int t, auto use) {
                if (first) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 + AP) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                sync();
                if (t + 1 < ntiles) dequant_store(use, (t + 1) & 1);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                sync();
            };
            pload(S0{}, 0); pload(S1{}, 1); pload(S2{}, 2); pload(S3{}, 3);
            stage_a(0); stage_a(1); stage_a(2);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(9 + 3 * AP) : "memory");            // tile 0's weights: 9 + 3 stages of dy pieces younger
            dequant_store(S0{}, 0);
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)" ::"n"(2 * AP) : "memory");      // dy tile 0
            sync();
            // t = 0: dy tile 1 landed <=> all but tile 2's pieces
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AP) : "memory");
            sync();
            pload(S0{}, 4);
            dequant_store(S1{}, 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            sync();
            stage_a(3);
            // steady steps t = 1 .. ntiles - 4 (groups of four, then 0..3 more), tail steps ntiles - 3 .. ntiles - 1
            int t = 1;
            for (; t + 7 <= ntiles; t += 4) {
                steady(t, S2{}, S1{});
                steady(t + 1, S3{}, S2{});
                steady(t + 2, S0{}, S3{});
                steady(t + 3, S1{}, S0{});
            }
            switch (ntiles - 3 - t) {
                case 0: tail(true, t, S2{}); tail(false, t + 1, S3{}); tail(false, t + 2, S0{}); break;
                case 1: steady(t, S2{}, S1{}); tail(true, t + 1, S3{}); tail(false, t + 2, S0{}); tail(false, t + 3, S1{}); break;
                case 2: steady(t, S2{}, S1{}); steady(t + 1, S3{}, S2{}); tail(true, t + 2, S0{}); tail(false, t + 3, S1{});
                        tail(false, t + 4, S2{}); break;
                default: steady(t, S2{}, S1{}); steady(t + 1, S3{}, S2{}); steady(t + 2, S0{}, S3{});
                         tail(true, t + 3, S1{}); tail(false, t + 4, S2{}); tail(false, t + 5, S3{}); break;
            }
        } else {
            // fp16 outlier k-tile: no dequantisation, so the weight tile goes straight into its LDS buffer by DMA (the
            // swizzle is applied on the SOURCE side: destination lane-linear, 4 rows of 256 B per instruction).  Buffer
            // (t + 2) & 1 is free from A(t) on (the compute waves fetched tile t's fragments during tile t - 1), which gives
            // the DMA one tile (B(t) .. B(t + 1)); a one-dword touch of every 64 bytes six tiles ahead pulls the rows into L2
            // so that one tile suffices.  iteration t: [wait] A(t) [wait] B(t) [touch t + 6, W tile t + 2, dy tile t + 3].
            const uint8_t* const obase = (const uint8_t*)ow + (size_t)(kt * D3_BK - kq) * 2;
            const size_t tile_stride = (size_t)D3_BN * n_out * 2;
            uint32_t o_off[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 16 * l + 4 * i + (lane >> 4), r3 = (lane >> 4) & 3;
                o_off[i] = (uint32_t)row * (uint32_t)n_out * 2u + (uint32_t)(((lane & 15) ^ (r3 << 2) ^ r3) << 4);
            }
            const uint32_t pf_off = (uint32_t)(16 * l + (lane >> 2)) * (uint32_t)n_out * 2u + (uint32_t)(lane & 3) * 64u;
            auto touch = [&](int t) {              // 1 operation
                t = min(t, ntiles - 1);
                asm volatile("global_load_dword a30, %0, %1" :: "v"(pf_off), "s"(obase + (size_t)t * tile_stride) : "memory", "a30");
            };
            auto stage_w = [&](int t) {            // 4 DMA operations
                const uint8_t* b = obase + (size_t)t * tile_stride;
                const uint32_t dst = lds0 + WOFF + (uint32_t)(t & 1) * D3_W + (uint32_t)l * 4096u;
#pragma unroll
                for (int i = 0; i < 4; ++i) g3_dma16(b, o_off[i], dst + i * 1024);
            };
            for (int t = 0; t < 6; ++t) touch(t);
            stage_w(0); stage_w(1);
            stage_a(0); stage_a(1); stage_a(2);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * AP) : "memory");            // dy tile 0 (and, older, both W tiles)
            sync();
            for (int t = 0; t < ntiles; ++t) {
                // dy tile t + 1: issued by iteration t - 2 (or the prologue); younger = iteration t - 1's 13 operations
                if (t == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AP) : "memory");
                else if (t + 2 < ntiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 + AP) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                sync();
                // W tile t + 1: issued by iteration t - 1 ahead of its dy pieces
                if (t + 2 < ntiles) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AP) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                sync();
                if (t + 3 < ntiles) {          // behind B(t), like the INT4 path's DMA
                    touch(t + 6);
                    stage_w(t + 2);
                    stage_a(t + 3);
                } else if (t + 2 < ntiles) {
                    stage_w(t + 2);
                }
            }
        }
        return;
    }

    // =============================================================== compute waves
    const int r = lane & 31, h = lane >> 5;
    f32x16 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

    // dy fragment of m-tile mt, n-step j: the 8 consecutive n  32 h + 8 j .. + 7  of row r (chunk 4 h + j, swizzled)
    uint32_t a_rd[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) a_rd[j] = (uint32_t)(r * 128 + (((h * 4 + j) ^ ((r >> 1) & 7)) << 4));
    // W fragment of n-step j: rows 32 h + 8 j + {0..3} and + {4..7} of column 32 wave + r, by two transposing reads: lane
    // 4 q + p of a 16-lane group addresses row q, columns 4 p .. 4 p + 3 of the group's 16 (qeft lds_read_tr8 note above)
    typedef short s4 __attribute__((ext_vector_type(4)));
    const int tq = (lane & 15) >> 2;
    const int tslot = wave * 4 + ((lane >> 4) & 1) * 2 + ((lane & 3) >> 1);
    const uint32_t w_rd = (uint32_t)(WOFF + (32 * h + tq) * 256 + ((tslot ^ (tq << 2) ^ tq) << 4) + (lane & 1) * 8);
    auto tr_read = [&](uint32_t off) -> u32x2 {
        return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(uintptr_t)(lds0 + off)));
    };
    auto fetch_w = [&](int buf, int j, u32x4& dst) {
        const u32x2 lo = tr_read(w_rd + buf * D3_W + (8 * j) * 256), hi = tr_read(w_rd + buf * D3_W + (8 * j + 4) * 256);
        dst = u32x4{lo[0], lo[1], hi[0], hi[1]};
    };

    u32x4 bA[4], bB[4];        // W fragments of the even / odd n-tiles
    u32x4 fa[4], fb[4];        // dy fragments, two m-tiles in rotation
    __builtin_amdgcn_s_barrier();              // dy tile 0 and W tile 0 are in LDS
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < 4; ++j) fetch_w(0, j, bA[j]);
#pragma unroll
    for (int j = 0; j < 4; ++j) fa[j] = *(const u32x4*)(lds + a_rd[j]);

    // One n-tile = 8 phases (one per 32-row m-tile): the fetch of the next m-tile's dy fragments a whole phase ahead (as in
    // the forward kernel), 4 MFMAs, and in phases 4..7 (behind barrier B) one n-step of the next tile's W fragments.
    auto tile_body = [&](auto next_tag, int t, const u32x4 (&bc)[4], u32x4 (&bn)[4]) {
        constexpr bool NEXT = decltype(next_tag)::value;
        const uint8_t* st = lds + (size_t)(t & (D3_ST - 1)) * A_B;
        const uint8_t* sn = lds + (size_t)((t + 1) & (D3_ST - 1)) * A_B;
        const int nbuf = (t + 1) & 1;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            u32x4 (&cur)[4] = (mt & 1) ? fb : fa;
            u32x4 (&nxt)[4] = (mt & 1) ? fa : fb;
            if (mt == 0) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, cur[0]), __builtin_bit_cast(h8, bc[0]),
                                                               acc[0], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (mt == MT / 2) {
                __builtin_amdgcn_s_barrier();              // B(t): W tile t + 1 is in LDS
                asm volatile("" ::: "memory");
            }
            if (mt < MT - 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) nxt[j] = *(const u32x4*)(st + a_rd[j] + (mt + 1) * 4096);
            } else if (NEXT) {
#pragma unroll
                for (int j = 0; j < 4; ++j) nxt[j] = *(const u32x4*)(sn + a_rd[j]);
            }
            if (NEXT && mt >= MT / 2) {                    // the next tile's W fragments, spread over the phases behind barrier B
                constexpr int PER = 8 / MT;                // n-steps per phase: 1 (8 phases) or 2 (4 phases)
#pragma unroll
                for (int jj = 0; jj < PER; ++jj) fetch_w(nbuf, (mt - MT / 2) * PER + jj, bn[(mt - MT / 2) * PER + jj]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = mt == 0 ? 1 : 0; j < 4; ++j)
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, cur[j]), __builtin_bit_cast(h8, bc[j]),
                                                                acc[mt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto sync_tile = [&]() {
        __builtin_amdgcn_s_barrier();              // A(t)
        asm volatile("" ::: "memory");
    };
    {
        int t = 0;
        if (ntiles & 1) {
            sync_tile();
            tile_body(std::true_type{}, 0, bA, bB);
#pragma unroll
            for (int j = 0; j < 4; ++j) bA[j] = bB[j];
            t = 1;
        }
        for (; t + 2 < ntiles; t += 2) {
            sync_tile();
            tile_body(std::true_type{}, t, bA, bB);
            sync_tile();
            tile_body(std::true_type{}, t + 1, bB, bA);
        }
        sync_tile();
        tile_body(std::true_type{}, t, bA, bB);
        sync_tile();
        tile_body(std::false_type{}, t + 1, bB, bA);
    }

    // ---- epilogue: as the forward kernel's -- the fp16 tile goes through LDS (the dy ring is free), 16-byte stores of whole rows
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    {
        uint8_t* const col = lds + (wave * 32 + r) * 2;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                *(f16*)(col + (mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * G3_YP) = (f16)acc[mt][e];
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    {
        const int ch = lane & 15;
#pragma unroll 4
        for (int i = 0; i < 2 * MT; ++i) {
            const int row = wave * (8 * MT) + i * 4 + (lane >> 4), m = bm0 + row;
            if (m >= M) continue;
            *(u32x4*)(dx + (size_t)m * K + kt * D3_BK + ch * 8) = *(const u32x4*)(lds + row * G3_YP + ch * 16);
        }
    }
}

// Split factor of the contraction (n) for dX: only when the 128 x 64 tiling gives too few blocks to hide the latency of
// a long n loop.
int gemm_w4_dx_split(int M, int N, int K) {
    const int blocks = ((M + DX_BM - 1) / DX_BM) * (K / DX_BK), ntiles = (N + DX_BN - 1) / DX_BN;
    if (blocks >= 1024 || ntiles < 32) return 1;
    int s = 1024 / blocks;             // measured: ~4 resident blocks per CU in total
    if (s > ntiles / 16) s = ntiles / 16;
    if (s > 8) s = 8;
    return s < 2 ? 1 : s;
}

// bits == 3: qw is the 3-bit extension layout; only the loader-wave tier reads it (hipErrorNotSupported otherwise: the caller
// expands to the 4-bit layout, qeft_expand_w3, and comes back with bits == 4)
// Tile of the loader-wave dX kernel for a shape: 8 (256 rows), 4 (128 rows) or 0 (the older kernels).  Same counting of rounds of
// blocks as the forward's choice (gemm_v3_prefers_256; dx256 takes ~1.6x a dx128v3 block: 68 against 43 us per round on 4096^2):
// the cheaper sum of rounds, ties to the 256-row tile; 128-row tiles from 112 tiles on (M > 128).
static int gemm_dx_v3_tile(int M, int K) {
    const int kb = K / D3_BK, t8 = ((M + D3_BM - 1) / D3_BM) * kb, t4 = ((M + 127) / 128) * kb;
    if (M > 256 && t8 >= 129 && 1.6 * ((t8 + 255) / 256) <= 1.0 * ((t4 + 255) / 256)) return 8;
    if (t4 >= 112 && M > 128) return 4;
    return 0;
}
static bool gemm_dx_v3_ok(int M, int N, int K, int G, int n_out) {
    return K % D3_BK == 0 && n_out % D3_BK == 0 && n_out < K && G % D3_BK == 0 && N % D3_BN == 0 && N >= 4 * D3_BN &&
           (size_t)M * N * 2 < (1ull << 32) && (size_t)(N / 4) * K * 2 < (1ull << 32) && (size_t)N * n_out * 2 < (1ull << 32);
}
bool gemm_w3_dx_native(int M, int N, int K, int G, int n_out) {
    return gemm_dx_v3_ok(M, N, K, G, n_out) && (size_t)(N / 16) * ((K - n_out) / 128) * 768 < (1ull << 32) && gemm_dx_v3_tile(M, K) != 0;
}

template <int BITS>
static hipError_t gemm_dx_v3_go(int tile, const void* dy, const void* qw, const void* scales, const void* zeros, const void* ow, void* dx,
                                int M, int N, int K, int G, int n_out, hipStream_t st) {
    const int kb = K / D3_BK, mbt = (M + 32 * tile - 1) / (32 * tile);
    const int smem = D3_ST * 32 * tile * D3_BN * 2 + 2 * D3_W;          // 163840 (256 rows) / 98304 (128 rows) bytes
    auto go = [&](auto kern) -> hipError_t {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(mbt * kb), dim3(512), smem, st, (const f16*)dy, (const uint8_t*)qw, (const f16*)scales,
                           (const f16*)zeros, (const f16*)ow, (f16*)dx, M, N, K, G, n_out, kb);
        return hipGetLastError();
    };
    return tile == 8 ? go(gemm_w4_dx_kernel_v3<BITS, 8>) : go(gemm_w4_dx_kernel_v3<BITS, 4>);
}

hipError_t gemm_w4_dx_launch(const void* dy, const void* qw, const void* scales, const void* zeros, const void* ow,
                             void* dx, int M, int N, int K, int G, int n_out, hipStream_t st, void* workspace,
                             size_t workspace_bytes, int bits) {
    const bool outl3 = ow && n_out > 0;
    if (bits == 3) {
        if (!gemm_w3_dx_native(M, N, K, G, outl3 ? n_out : 0)) return hipErrorNotSupported;
        const int tile = gemm_dx_v3_tile(M, K);
        g_last_variant = tile == 8 ? "dx256_w3" : "dx128v3_w3";
        return gemm_dx_v3_go<3>(tile, dy, qw, scales, zeros, outl3 ? ow : nullptr, dx, M, N, K, G, outl3 ? n_out : 0, st);
    }
    // Loader-wave tiles (gemm_w4_dx_kernel_v3): 256 x 128 when they give (nearly) every CU a block -- the M >= 2048 tier of a
    // fine-tune step --, 128 x 128 below that (round 3).  QEFT_DX_V3 = 0 / 1 forces the choice (1: the 256-row tile) (A/B).
    {
        static const int force_v3 = getenv("QEFT_DX_V3") ? atoi(getenv("QEFT_DX_V3")) : -1;
        const int tile = force_v3 == 1 ? 8 : force_v3 == 0 ? 0 : gemm_dx_v3_tile(M, K);
        if (tile && gemm_dx_v3_ok(M, N, K, G, outl3 ? n_out : 0)) {
            g_last_variant = tile == 8 ? "dx256" : "dx128v3";
            return gemm_dx_v3_go<4>(tile, dy, qw, scales, zeros, outl3 ? ow : nullptr, dx, M, N, K, G, outl3 ? n_out : 0, st);
        }
    }
    // the 128-wide tile when it still gives every CU two blocks; smaller problems keep the 64-wide tile (twice the
    // blocks, and a split over n below 512 of them)
    if (K % 128 == 0 && n_out % 32 == 0 && ((M + DX_BM - 1) / DX_BM) * (K / 128) >= 512) {
        dim3 grid2((M + DX_BM - 1) / DX_BM, K / 128);
        hipLaunchKernelGGL(gemm_w4_dx128_kernel, grid2, dim3(256), 0, st, (const f16*)dy, (const uint8_t*)qw,
                           (const f16*)scales, (const f16*)zeros, (n_out > 0 ? (const f16*)ow : (const f16*)nullptr),
                           (f16*)dx, M, N, K, G, n_out);
        g_last_variant = "dx128";
        return hipGetLastError();
    }
    int S = workspace ? gemm_w4_dx_split(M, N, K) : 1;
    while (S > 1 && (size_t)S * M * K * 4 > workspace_bytes) --S;
    dim3 grid((M + DX_BM - 1) / DX_BM, K / DX_BK, S);
    hipLaunchKernelGGL(gemm_w4_dx_kernel, grid, dim3(256), 0, st, (const f16*)dy, (const uint8_t*)qw,
                       (const f16*)scales, (const f16*)zeros, (n_out > 0 ? (const f16*)ow : (const f16*)nullptr),
                       (f16*)dx, M, N, K, G, n_out, (float*)workspace);
    g_last_variant = S > 1 ? "dx64+split" : "dx64";
    if (S > 1) {
        const size_t quads = (size_t)M * K / 4;
        hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((int)((quads + 255) / 256)), dim3(256), 0, st,
                           (const float*)workspace, (const f16*)nullptr, (f16*)dx, M, K, S);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// d_oweight[N, r] (fp32) = dy[M,N]^T . x[M, K-r:]   (qlinear.py:41-42).  Small (2*M*N*r flops); LDS-tiled
// fp32 FMA kernel: block = 64 (n) x 64 (j) outputs, 16x16 threads x 4x4 each, m advances 16 per stage.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void grad_oweight_kernel(const f16* __restrict__ dy, const f16* __restrict__ x,
                                                           float* __restrict__ dow, int M, int N, int K, int R) {
    __shared__ float sdy[16][64 + 1];
    __shared__ float sx[16][64 + 1];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int n0 = blockIdx.x * 64, j0 = blockIdx.y * 64;
    const int kq = K - R;
    float acc[4][4] = {};
    for (int m0 = 0; m0 < M; m0 += 16) {
        for (int idx = threadIdx.x; idx < 16 * 64; idx += 256) {
            const int mm = idx >> 6, c = idx & 63;
            const int m = m0 + mm;
            sdy[mm][c] = (m < M && n0 + c < N) ? (float)dy[(size_t)m * N + n0 + c] : 0.f;
            sx[mm][c] = (m < M && j0 + c < R) ? (float)x[(size_t)m * K + kq + j0 + c] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int mm = 0; mm < 16; ++mm) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = sdy[mm][ty * 4 + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = sx[mm][tx * 4 + j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] += a[i] * b[j];
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + ty * 4 + i, jj = j0 + tx * 4 + j;
            if (n < N && jj < R) dow[(size_t)n * R + jj] = acc[i][j];
        }
}

// MFMA version.  d_oweight = dy^T . x_o contracts over m, the ROW index of both operands, so both MFMA fragments are
// columns of row-major tiles: rows of dy and x_o go to LDS as they lie in HBM (16-byte copies) and come back through the
// transposing LDS read (as the weight tile of gemm_w4_dx_kernel).  Block = BN (n) x 64 (j) outputs over ALL m (no
// cross-block reduction: deterministic).  A wave owns 32 of every 128 rows end to end: its loads (D slabs in flight in
// registers), a PRIVATE 7-9 KB of LDS it writes and reads back transposed (LDS operations of one wave execute in order,
// so one buffer is enough and no barrier is involved), its MFMAs; the fragments of slab i+1 are fetched before the MFMAs
// of slab i.  The four waves meet once, in the final sum (wave order).
// History: fp32 FMAs on an LDS tile took longer than the dX GEMM for 3 % of its flops; block-wide 128-row stages with
// two barriers each (round 1) were not memory-bound either -- with every load an L2 hit they still took 0.4 us per stage,
// the serial chain wait / store / barrier / read / MFMA / barrier of four waves in lock step.  What bounds this form is
// the bytes a CU takes in (~55 GB/s per CU here, as in the GEMM's k loop): 2 (BN + 64) bytes per m row and block, so
// BN = 64 for wide layers, where 32-column blocks would put three blocks on a CU.
constexpr int GO_BJ = 64;
constexpr int GO_PX = GO_BJ * 2 + 16;    // bytes per m row of the x_o slab

template <int BN, int D>
__global__ __launch_bounds__(256) void grad_oweight_wave_kernel(const f16* __restrict__ dy, const f16* __restrict__ x,
                                                                float* __restrict__ dow, int M, int N, int K, int R) {
    constexpr int PDY = BN * 2 + 16;     // bytes per m row of the dy slab
    constexpr int NT = BN / 32;          // 32-column MFMA tiles of dy
    constexpr int DPR = BN / 8;          // 16-byte pieces per dy row
    constexpr int ND = 32 * DPR / 64;    // dy loads per lane and slab
    constexpr int WB = 32 * (PDY + GO_PX);
    constexpr int GO_LDS = 4 * WB > 32768 ? 4 * WB : 32768;
    typedef short s4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) uint8_t lds_all[GO_LDS];
    float(*const red)[2][16][64] = (float(*)[2][16][64])lds_all;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint8_t* const wdy = lds_all + wave * WB;
    uint8_t* const wx = wdy + 32 * PDY;
    const int n0 = blockIdx.x * BN, j0 = blockIdx.y * GO_BJ, kq = K - R;

    f32x16 acc[NT][2];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][jt][e] = 0.f;

    struct Stage {
        u32x4 d[ND];   // dy: items lane + 64 i -> (row = item / DPR, 16-byte piece = item % DPR)
        u32x4 xv[4];   // x_o: items lane + 64 i -> (row = item >> 3, piece = item & 7)
    };
    struct Frag {
        u32x2 a[2][NT][2], b[2][2][2];     // [16-row step][tile][rows 0-3 / 4-7 of the lane's 8]
    };
    // slab s of this wave = rows 128 s + 32 wave ..; loads are unconditional with clamped addresses (a load in a branch
    // makes hipcc drain vmcnt(0)); what lies outside [M) x [N) / [R) is zeroed on its way into LDS
    auto gload = [&](int s, Stage& st) {
        const int m0 = s * 128 + wave * 32;
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int item = lane + 64 * i, m = min(m0 + item / DPR, M - 1), n = min(n0 + (item % DPR) * 8, N - 8);
            st.d[i] = *(const u32x4*)(dy + (size_t)m * N + n);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int item = lane + 64 * i, m = min(m0 + (item >> 3), M - 1), j = min(j0 + (item & 7) * 8, R - 8);
            st.xv[i] = *(const u32x4*)(x + (size_t)m * K + kq + j);
        }
    };
    auto lstore = [&](const Stage& st, int s) {
        const int m0 = s * 128 + wave * 32;
        const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int item = lane + 64 * i;
            const bool ok = m0 + item / DPR < M && n0 + (item % DPR) * 8 < N;
            *(u32x4*)(wdy + (item / DPR) * PDY + (item % DPR) * 16) = ok ? st.d[i] : z;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int item = lane + 64 * i;
            const bool ok = m0 + (item >> 3) < M && j0 + (item & 7) * 8 < R;
            *(u32x4*)(wx + (item >> 3) * GO_PX + (item & 7) * 16) = ok ? st.xv[i] : z;
        }
    };
    // transposed-read addresses (16-m step 0): lane 4q+p of a 16-lane group addresses row q, columns 4p.. of its block;
    // group g = lane >> 4 -> columns (g & 1)*16 .., contraction rows 8*(g >> 1) ..
    const uint32_t rowsel = (uint32_t)(8 * (lane >> 5) + ((lane & 15) >> 2));
    const uint32_t colsel = (uint32_t)(((lane >> 4) & 1) * 16 + (lane & 3) * 4) * 2;
    auto tr_read = [&](const uint8_t* p) -> u32x2 {
        return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(uintptr_t)p));
    };
    const uint8_t* const a_tr = wdy + rowsel * PDY + colsel;
    const uint8_t* const b_tr = wx + rowsel * GO_PX + colsel;
    auto fetch = [&](Frag& f) {
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                f.a[ss][nt][0] = tr_read(a_tr + (ss * 16) * PDY + nt * 64);
                f.a[ss][nt][1] = tr_read(a_tr + (ss * 16 + 4) * PDY + nt * 64);
            }
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {
                f.b[ss][jt][0] = tr_read(b_tr + (ss * 16) * GO_PX + jt * 64);
                f.b[ss][jt][1] = tr_read(b_tr + (ss * 16 + 4) * GO_PX + jt * 64);
            }
        }
    };
    auto mma = [&](const Frag& f) {
#pragma unroll
        for (int ss = 0; ss < 2; ++ss)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int jt = 0; jt < 2; ++jt)
                    acc[nt][jt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                        __builtin_bit_cast(h8, u32x4{f.a[ss][nt][0][0], f.a[ss][nt][0][1], f.a[ss][nt][1][0], f.a[ss][nt][1][1]}),
                        __builtin_bit_cast(h8, u32x4{f.b[ss][jt][0][0], f.b[ss][jt][0][1], f.b[ss][jt][1][0], f.b[ss][jt][1][1]}),
                        acc[nt][jt], 0, 0, 0);
    };

    const int slabs = (M + 127) >> 7;
    Stage st[D];
    Frag cur, nxt;
#pragma unroll
    for (int i = 0; i < D; ++i) gload(i, st[i]);
    lstore(st[0], 0);
    gload(D, st[0]);
    fetch(cur);
    // body for slab s (in stage st[s % D]): its rows go to LDS behind the reads of slab s-1 that are already issued, its
    // stage is refilled, its fragments are fetched, and the MFMAs of slab s-1 run while they arrive.  Slabs past the
    // last are zeros.
    for (int s0 = 1; s0 <= slabs; s0 += D) {
#pragma unroll
        for (int i = 0; i < D; ++i) {
            Stage& sg = st[(i + 1) % D];
            lstore(sg, s0 + i);
            gload(s0 + i + D, sg);
            __builtin_amdgcn_sched_barrier(0);
            fetch(nxt);
            __builtin_amdgcn_sched_barrier(0);
            mma(cur);
            __builtin_amdgcn_sched_barrier(0);
            cur = nxt;
        }
    }
    __syncthreads();                                   // every wave is done with its slab buffer: `red` lies over them
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {                  // 32 columns of dy at a time through the 32 KB
        if (nt) __syncthreads();
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int e = 0; e < 16; ++e) red[wave][jt][e][lane] = acc[nt][jt][e];
        __syncthreads();
        for (int idx = tid; idx < 2 * 16 * 64; idx += 256) {
            const int jt = idx >> 10, e = (idx >> 6) & 15, l = idx & 63;
            const float v = ((red[0][jt][e][l] + red[1][jt][e][l]) + red[2][jt][e][l]) + red[3][jt][e][l];
            const int n = n0 + nt * 32 + (e & 3) + 8 * (e >> 2) + 4 * (l >> 5), j = j0 + jt * 32 + (l & 31);
            if (n < N && j < R) dow[(size_t)n * R + j] = v;
        }
    }
}

hipError_t grad_oweight_launch(const void* dy, const void* x, void* dow, int M, int N, int K, int n_out,
                               hipStream_t st) {
    if (N % 8 == 0 && n_out % 8 == 0 && K % 8 == 0) {   // 16-byte row pieces
        static const int force_bn = [] { const char* e = getenv("QEFT_DOW_BN"); return e ? atoi(e) : 0; }();
        const int jb = (n_out + GO_BJ - 1) / GO_BJ;
        const bool wide = force_bn ? force_bn == 64 : (N + 31) / 32 * jb > 512;
        const dim3 grid2((N + (wide ? 63 : 31)) / (wide ? 64 : 32), jb);
        if (wide)
            hipLaunchKernelGGL((grad_oweight_wave_kernel<64, 2>), grid2, dim3(256), 0, st, (const f16*)dy, (const f16*)x,
                               (float*)dow, M, N, K, n_out);
        else
            hipLaunchKernelGGL((grad_oweight_wave_kernel<32, 2>), grid2, dim3(256), 0, st, (const f16*)dy, (const f16*)x,
                               (float*)dow, M, N, K, n_out);
        g_last_variant = wide ? "grad_oweight_mfma_n64" : "grad_oweight_mfma";
        return hipGetLastError();
    }
    g_last_variant = "grad_oweight_fma";
    dim3 grid((N + 63) / 64, (n_out + 63) / 64);
    hipLaunchKernelGGL(grad_oweight_kernel, grid, dim3(256), 0, st, (const f16*)dy, (const f16*)x, (float*)dow, M, N, K,
                       n_out);
    return hipGetLastError();
}

}  // namespace qeft
