// Lab (GPU box only): does the cache-policy hint of a streaming read change the HBM rate?  A pure read kernel at the decode GEMV's
// launch sizes, 12 buffers cycled (nothing served from L2 / MALL), loads issued as buffer loads with the gfx950 cache-policy bits
// (aux: 1 = sc0, 2 = nt, 16 = sc1) against the plain and the __builtin_nontemporal_load forms the product uses.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/stream_policy_lab.hip -o build/stream_policy_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int AUX, int U>      // AUX: -1 plain global load, -2 nontemporal builtin, >= 0 buffer load with these aux bits
__global__ __launch_bounds__(256) void stream_read(const u32x4* __restrict__ p, size_t vec_per_block, uint32_t* out) {
    const u32x4* base = p + (size_t)blockIdx.x * vec_per_block;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(vec_per_block * 16), 0x00020000);
    u32x4 acc = {0, 0, 0, 0};
    for (size_t i = threadIdx.x; i < vec_per_block; i += 256 * U) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t idx = i + u * 256 < vec_per_block ? i + u * 256 : vec_per_block - 1;
            if constexpr (AUX == -1) v[u] = base[idx];
            else if constexpr (AUX == -2) v[u] = __builtin_nontemporal_load(base + idx);
            else v[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(idx * 16), 0, AUX);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[threadIdx.x] = 1;
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

template <int AUX>
static void run(const char* name, void* buf, uint32_t* out, int L) {
    printf("%-22s", name);
    for (double mb : {9.3, 27.8, 49.8, 250.0}) {
        size_t bytes = (size_t)(mb * 1024 * 1024) / 16384 * 16384;
        const int grid = 1024;
        size_t vpb = bytes / 16 / grid;
        // as many distinct buffers as fit the 672 MiB allocation: the footprint of a cycle exceeds the 256 MB MALL at every size
        const size_t stride = (bytes + (1u << 20) - 1) >> 20 << 20;
        const int LL = (int)(((size_t)L * 56 * 1024 * 1024) / stride);
        auto f = [&](int l) { hipLaunchKernelGGL((stream_read<AUX, 8>), dim3(grid), dim3(256), 0, 0, (const u32x4*)((char*)buf + (size_t)l * stride), vpb, out); };
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) { float u = time_launches(LL > 20 ? 6 : 20, LL, f); best = u < best ? u : best; }
        printf("  %5.1f MiB: %6.2f us %5.0f GB/s", mb, best, bytes / best / 1e3);
    }
    printf("\n");
}

int main() {
    const int L = 12;
    uint32_t* out; CK(hipMalloc(&out, 4096));
    size_t tot = (size_t)L * 56 * 1024 * 1024;
    void* buf; CK(hipMalloc(&buf, tot)); CK(hipMemset(buf, 1, tot));
    for (int pass = 0; pass < 2; ++pass) {
        run<-1>("global_load", buf, out, L);
        run<-2>("global_load nt", buf, out, L);
        run<0>("buffer_load", buf, out, L);
        run<1>("buffer_load sc0", buf, out, L);
        run<2>("buffer_load nt", buf, out, L);
        run<3>("buffer_load sc0 nt", buf, out, L);
        run<16>("buffer_load sc1", buf, out, L);
        run<17>("buffer_load sc0 sc1", buf, out, L);
        run<18>("buffer_load sc1 nt", buf, out, L);
        run<19>("buffer_load sc0 sc1 nt", buf, out, L);
    }
    return 0;
}
