// Exploration harness (GPU box only): streaming-read ceiling and ablations of the decode GEMV.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I qeft_amd/csrc tools/gemv_lab.hip -o gpurun_out/gemv_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gemv_w4_mfma.h"

using namespace qeft;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// pure streaming read: each block reads a contiguous slab of `bytes_per_block`, 16 B per lane, U loads in flight
template <int U>
__global__ __launch_bounds__(256) void stream_read(const u32x4* __restrict__ p, size_t vec_per_block, uint32_t* out) {
    const u32x4* base = p + (size_t)blockIdx.x * vec_per_block;
    u32x4 acc = {0, 0, 0, 0};
    for (size_t i = threadIdx.x; i < vec_per_block; i += 256 * U) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = (i + u * 256 < vec_per_block) ? __builtin_nontemporal_load(base + i + u * 256) : u32x4{0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[threadIdx.x] = 1;
}

struct Layer { void *qw, *sc, *sz, *ow; };

template <typename F>
float time_launches(int reps, int L, F f) {
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

template <int NW, int RGI, int D, int ABL>
void run_variant(const char* name, std::vector<Layer>& Ls, void* x, void* y, int N, int K, double bytes) {
    const int L = (int)Ls.size();
    auto kern = gemv_w4_kernel<NW, RGI, 1, D, true, false, ABL>;
    size_t smem = gemv_smem_bytes(NW, RGI, 1, K, 128, 128);
    auto f = [&](int l) {
        GemvArgs a{(const f16*)x, (const uint8_t*)Ls[l].qw, (const f16*)Ls[l].sc, (const f16*)Ls[l].sz, (const f16*)Ls[l].ow,
                   nullptr, nullptr, nullptr, (f16*)y, N, K, 128, 128, 7, nullptr, 0.f};
        hipLaunchKernelGGL(kern, dim3(N / (4 * RGI)), dim3(NW * 64), smem, 0, a);
    };
    float us = time_launches(20, L, f);
    printf("  %-20s NW=%d RGI=%d D=%d ABL=%2d : %7.2f us  %6.0f GB/s\n", name, NW, RGI, D, ABL, us, bytes / us / 1e3);
}

template <int NW, int D, int ABL>
void run_mfma(const char* name, std::vector<Layer>& Ls, void* x, void* y, int N, int K, double bytes) {
    const int L = (int)Ls.size();
    auto kern = gemv_w4_mfma_kernel<NW, 1, D, true, false, 0, ABL>;
    size_t smem = gemv_mfma_smem_bytes(NW, 1, K, 128);
    auto f = [&](int l) {
        GemvArgs a{(const f16*)x, (const uint8_t*)Ls[l].qw, (const f16*)Ls[l].sc, (const f16*)Ls[l].sz, (const f16*)Ls[l].ow,
                   nullptr, nullptr, nullptr, (f16*)y, N, K, 128, 128, 7, nullptr, 0.f};
        hipLaunchKernelGGL(kern, dim3(N / 16), dim3(NW * 64), smem, 0, a);
    };
    float us = time_launches(20, L, f);
    printf("  %-20s MFMA NW=%d D=%d ABL=%2d : %7.2f us  %6.0f GB/s\n", name, NW, D, ABL, us, bytes / us / 1e3);
}

int main(int argc, char** argv) {
    const int L = 12;
    int shapes[3][2] = {{4096, 4096}, {11008, 4096}, {4096, 11008}};
    uint32_t* out; CK(hipMalloc(&out, 4096));
    // ---- streaming ceilings
    {
        size_t tot = (size_t)L * 32 * 1024 * 1024;
        void* buf; CK(hipMalloc(&buf, tot)); CK(hipMemset(buf, 1, tot));
        for (size_t mb : {8, 22}) {
            size_t bytes = mb * 1024 * 1024;
            for (int grid : {256, 512, 1024, 2048}) {
                size_t vpb = bytes / 16 / grid;
                auto f4 = [&](int l) { hipLaunchKernelGGL(stream_read<4>, dim3(grid), dim3(256), 0, 0, (const u32x4*)((char*)buf + (size_t)l * 32 * 1024 * 1024), vpb, out); };
                auto f8 = [&](int l) { hipLaunchKernelGGL(stream_read<8>, dim3(grid), dim3(256), 0, 0, (const u32x4*)((char*)buf + (size_t)l * 32 * 1024 * 1024), vpb, out); };
                float u4 = time_launches(20, L, f4), u8 = time_launches(20, L, f8);
                printf("stream %2zu MiB grid %4d: U4 %6.2f us %6.0f GB/s | U8 %6.2f us %6.0f GB/s\n", mb, grid, u4, bytes / u4 / 1e3, u8, bytes / u8 / 1e3);
            }
        }
        CK(hipFree(buf));
    }
    for (auto& sh : shapes) {
        const int N = sh[0], K = sh[1];
        std::vector<Layer> Ls(L);
        for (auto& l : Ls) {
            CK(hipMalloc(&l.qw, (size_t)N * K / 2)); CK(hipMemset(l.qw, 0x5a, (size_t)N * K / 2));
            CK(hipMalloc(&l.sc, (size_t)K / 128 * N * 2)); CK(hipMemset(l.sc, 0x1c, (size_t)K / 128 * N * 2));
            CK(hipMalloc(&l.sz, (size_t)K / 128 * N * 2)); CK(hipMemset(l.sz, 0x9c, (size_t)K / 128 * N * 2));
            CK(hipMalloc(&l.ow, (size_t)N * 128 * 2)); CK(hipMemset(l.ow, 0x1c, (size_t)N * 128 * 2));
        }
        void *x, *y; CK(hipMalloc(&x, K * 2)); CK(hipMemset(x, 0x3c, K * 2)); CK(hipMalloc(&y, N * 2));
        double bytes = (double)N * (K - 128) / 2 + 2.0 * (K / 128) * N * 2 + (double)N * 128 * 2 + 2 * K + 2 * N;
        printf("N=%d K=%d algorithmic bytes %.0f\n", N, K, bytes);
        run_variant<8, 4, 2, 0>("valu full", Ls, x, y, N, K, bytes);
        run_variant<8, 4, 4, 0>("valu full", Ls, x, y, N, K, bytes);
        run_mfma<8, 2, 0>("mfma full", Ls, x, y, N, K, bytes);
        run_mfma<8, 4, 0>("mfma full", Ls, x, y, N, K, bytes);
        run_mfma<8, 6, 0>("mfma full", Ls, x, y, N, K, bytes);
        run_mfma<4, 4, 0>("mfma full", Ls, x, y, N, K, bytes);
        run_mfma<8, 2, 8>("mfma no xcd remap", Ls, x, y, N, K, bytes);
        run_mfma<8, 4, 8>("mfma no xcd remap", Ls, x, y, N, K, bytes);
        run_mfma<8, 4, 1>("mfma no scale ld", Ls, x, y, N, K, bytes);
        run_mfma<8, 4, 4>("mfma no math", Ls, x, y, N, K, bytes);
        run_mfma<8, 4, 5>("mfma no math/sc", Ls, x, y, N, K, bytes);
        for (auto& l : Ls) { (void)hipFree(l.qw); (void)hipFree(l.sc); (void)hipFree(l.sz); (void)hipFree(l.ow); }
        (void)hipFree(x); (void)hipFree(y);
    }
    return 0;
}
