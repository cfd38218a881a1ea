// Lab (GPU box only): where do the 8 us between the fused norm + lm_head launch (46 us) and a pure read of the same 262 MB (38 us) go?
// Variants of the kernel's structure: V = 0 as the product; 1 without the norm prologue; 2 without the result store; 3 both;
// NT = threads per block (512 / 256), blocks chosen so that 2 / 4 blocks sit on a CU.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lm_head_lab.hip -o build/lm_head_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16;
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__device__ __forceinline__ float wave_sum(float v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
template <int V, int NT, int RIF>      // RIF: rows in flight per wave (2 = the product's cur / nxt; 3 = one more)
__global__ __launch_bounds__(NT) void lm_head(const float* __restrict__ h32, const f16* __restrict__ gamma, const f16* __restrict__ W,
                                               f16* __restrict__ logits, int vocab, float eps, int rows_per_block) {
    constexpr int LPR = 8, H = 4096, NWV = NT / 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    h2 xh[LPR][4];
    if (V & 1) {
        for (int c = 0; c < LPR; ++c) for (int j = 0; j < 4; ++j) xh[c][j] = h2{(f16)0.01f, (f16)0.02f};
    } else {
        float xv[LPR][8];
        float ss = 0.f;
#pragma unroll
        for (int c = 0; c < LPR; ++c) {
            const f32x4 a = *(const f32x4*)(h32 + c * 512 + lane * 8), b = *(const f32x4*)(h32 + c * 512 + lane * 8 + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { xv[c][j] = a[j]; xv[c][4 + j] = b[j]; ss += a[j] * a[j] + b[j] * b[j]; }
        }
        ss = wave_sum(ss);
        const float rs = rsqrtf(ss / (float)H + eps);
#pragma unroll
        for (int c = 0; c < LPR; ++c) {
            const h8 g = *(const h8*)(gamma + c * 512 + lane * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) xh[c][j] = h2{(f16)(xv[c][2 * j] * rs * (float)g[2 * j]), (f16)(xv[c][2 * j + 1] * rs * (float)g[2 * j + 1])};
        }
    }
    const int r_end = min(vocab, (int)(blockIdx.x + 1) * rows_per_block);
    int r = blockIdx.x * rows_per_block + wave;
    if constexpr (RIF == 0) {      // two row buffers with alternating roles, no register copies: the loads of one row stay in flight across the other's dots
        u32x4 A[LPR], B[LPR];
        auto ld = [&](int row, u32x4 (&dst)[LPR]) {
            const u32x4* p = (const u32x4*)(W + (size_t)min(row, vocab - 1) * H) + lane;
#pragma unroll
            for (int c = 0; c < LPR; ++c) dst[c] = __builtin_nontemporal_load(p + c * 64);
        };
        auto dots = [&](const u32x4 (&src)[LPR], int row) {
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < LPR; ++c)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, src[c][j]), xh[c][j], acc, false);
            acc = wave_sum(acc);
            if (lane == 0 && row < r_end) logits[row] = (f16)acc;
        };
        ld(r, A);
        for (; r < r_end; r += 2 * NWV) {
            ld(r + NWV, B);
            dots(A, r);
            ld(r + 2 * NWV, A);
            dots(B, r + NWV);
        }
        return;
    }
    u32x4 buf[RIF == 0 ? 1 : RIF][LPR];
    auto load_row = [&](int row, u32x4 (&dst)[LPR]) {
        const u32x4* p = (const u32x4*)(W + (size_t)min(row, vocab - 1) * H) + lane;
#pragma unroll
        for (int c = 0; c < LPR; ++c) dst[c] = __builtin_nontemporal_load(p + c * 64);
    };
#pragma unroll
    for (int i = 0; i < RIF - 1; ++i) load_row(r + i * NWV, buf[i]);
    float keep = 0.f;
    for (; r < r_end; r += NWV) {
        load_row(r + (RIF - 1) * NWV, buf[RIF - 1]);
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < LPR; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, buf[0][c][j]), xh[c][j], acc, false);
        acc = wave_sum(acc);
        if (V & 2) keep += acc; else if (lane == 0) logits[r] = (f16)acc;
#pragma unroll
        for (int i = 0; i + 1 < RIF; ++i)
#pragma unroll
            for (int c = 0; c < LPR; ++c) buf[i][c] = buf[i + 1][c];
    }
    if ((V & 2) && keep == 123.456f && lane == 0) logits[0] = (f16)keep;
}
template <typename F>
static float time_launches(int reps, int L, F f) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int l = 0; l < L; ++l) f(l);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) for (int l = 0; l < L; ++l) f(l);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / (reps * L);
}
template <int V, int NT, int RIF>
static void run(const char* name, int blocks, const float* h, const f16* g, char* W, f16* out) {
    const int vocab = 32000, rpb = (vocab + blocks - 1) / blocks;
    auto f = [&](int l) { hipLaunchKernelGGL((lm_head<V, NT, RIF>), dim3(blocks), dim3(NT), 0, 0, h, g, (const f16*)(W + (size_t)l * 262144000), out, vocab, 1e-5f, rpb); };
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) { float u = time_launches(10, 4, f); best = u < best ? u : best; }
    printf("%-44s blocks %4d x %3d: %6.2f us  %5.0f GB/s\n", name, blocks, NT, best, 262144000.0 / best / 1e3);
}
int main() {
    float* h; f16 *g, *out; char* W;
    CK(hipMalloc(&h, 4096 * 4)); CK(hipMalloc(&g, 4096 * 2)); CK(hipMalloc(&out, 32000 * 2));
    CK(hipMalloc(&W, (size_t)4 * 262144000)); CK(hipMemset(W, 0x11, (size_t)4 * 262144000)); CK(hipMemset(h, 0, 4096 * 4)); CK(hipMemset(g, 0, 4096 * 2));
    for (int pass = 0; pass < 2; ++pass) {
        run<0, 512, 2>("product form", 512, h, g, W, out);
        run<1, 512, 2>("no norm prologue", 512, h, g, W, out);
        run<2, 512, 2>("no result store", 512, h, g, W, out);
        run<3, 512, 2>("neither", 512, h, g, W, out);
        run<0, 256, 2>("product form, 4-wave blocks", 1024, h, g, W, out);
        run<0, 256, 2>("product form, 4-wave blocks", 2048, h, g, W, out);
        run<0, 512, 2>("product form", 1024, h, g, W, out);
        run<0, 256, 3>("4-wave blocks, 3 rows in flight", 1024, h, g, W, out);
        run<0, 512, 3>("3 rows in flight", 512, h, g, W, out);
        run<0, 512, 0>("two buffers, alternating roles", 512, h, g, W, out);
        run<1, 512, 0>("two buffers, alternating, no norm", 512, h, g, W, out);
        run<0, 256, 0>("two buffers, alternating, 4-wave blocks", 1024, h, g, W, out);
    }
    return 0;
}
