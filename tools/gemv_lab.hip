// Exploration harness (GPU box only): streaming-read ceiling and ablations of the decode GEMV.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I qeft_amd/csrc tools/gemv_lab.hip -o gpurun_out/gemv_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
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

__global__ void fill_random(uint32_t* p, size_t n, uint32_t seed, uint32_t andmask, uint32_t ormask) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t v = (uint32_t)i * 2654435761u ^ seed;
    v ^= v >> 16; v *= 0x85ebca6bu; v ^= v >> 13; v *= 0xc2b2ae35u; v ^= v >> 16;
    p[i] = (v & andmask) | ormask;
}
static void rnd(void* p, size_t bytes, uint32_t seed, uint32_t andmask = 0xffffffffu, uint32_t ormask = 0) {
    size_t n = bytes / 4;
    hipLaunchKernelGGL(fill_random, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (uint32_t*)p, n, seed, andmask, ormask);
}

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
                   nullptr, nullptr, nullptr, (f16*)y, N, K, 128, 128, 7, nullptr, 0.f, nullptr, nullptr, nullptr, nullptr, 1};
        hipLaunchKernelGGL(kern, dim3(N / (4 * RGI)), dim3(NW * 64), smem, 0, a);
    };
    float us = time_launches(20, L, f);
    printf("  %-20s NW=%d RGI=%d D=%d ABL=%2d : %7.2f us  %6.0f GB/s\n", name, NW, RGI, D, ABL, us, bytes / us / 1e3);
}

template <int NW, int D, int ABL, int XT = 0>
void run_mfma(const char* name, std::vector<Layer>& Ls, void* x, void* y, int N, int K, double bytes, int nblk = 0) {
    const int L = (int)Ls.size();
    const int nsets = N / 16;
    if (nblk <= 0 || nblk > nsets) nblk = nsets;
    const int rs_cap = (nsets + nblk - 1) / nblk;
    auto kern = gemv_w4_mfma_kernel<NW, 1, D, true, false, XT, ABL>;
    size_t smem = gemv_mfma_smem_bytes(NW, 1, K, 128, rs_cap);
    if (smem > 64 * 1024) return;
    auto f = [&](int l) {
        GemvArgs a{(const f16*)x, (const uint8_t*)Ls[l].qw, (const f16*)Ls[l].sc, (const f16*)Ls[l].sz, (const f16*)Ls[l].ow,
                   nullptr, nullptr, nullptr, (f16*)y, N, K, 128, 128, 7, XT ? (const f16*)x : nullptr, 1e-5f, nullptr, nullptr, nullptr, nullptr, 1};
        hipLaunchKernelGGL(kern, dim3(nblk), dim3(NW * 64), smem, 0, a, rs_cap);
    };
    float us = time_launches(20, L, f);
    printf("  %-20s MFMA NW=%d D=%d blocks=%4d XT=%d ABL=%2d : %7.2f us  %6.0f GB/s\n", name, NW, D, nblk, XT, ABL, us, bytes / us / 1e3);
}

// one launch with per-wave stamps: block lifetime and phase boundaries
template <int NW, int D, int XT, int TLABL = 16>
void timeline(std::vector<Layer>& Ls, void* x, void* y, int N, int K, int nblk) {
    const int nsets = N / 16;
    if (nblk <= 0 || nblk > nsets) nblk = nsets;
    const int rs_cap = (nsets + nblk - 1) / nblk;
    auto kern = gemv_w4_mfma_kernel<NW, 1, D, true, false, XT, TLABL>;
    size_t smem = gemv_mfma_smem_bytes(NW, 1, K, 128, rs_cap);
    if (smem > 64 * 1024) return;
    unsigned long long *dbg, *dbg2; CK(hipMalloc(&dbg, (size_t)nblk * NW * 8 * 8)); CK(hipMalloc(&dbg2, (size_t)nblk * NW * 4 * 8));
    CK(hipMemset(dbg2, 0, (size_t)nblk * NW * 4 * 8));
    std::vector<unsigned long long> h((size_t)nblk * NW * 8);
    for (int rep = 0; rep < 3; ++rep) {
        GemvArgs a{(const f16*)x, (const uint8_t*)Ls[rep].qw, (const f16*)Ls[rep].sc, (const f16*)Ls[rep].sz, (const f16*)Ls[rep].ow,
                   nullptr, nullptr, nullptr, (f16*)y, N, K, 128, 128, 7, XT ? (const f16*)x : nullptr, 1e-5f, nullptr, dbg, dbg2, nullptr, 1};
        hipLaunchKernelGGL(kern, dim3(nblk), dim3(NW * 64), smem, 0, a, rs_cap);
        CK(hipDeviceSynchronize());
    }
    CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull, t1 = 0;
    for (size_t i = 0; i < h.size(); i += 8) { if (h[i] < t0) t0 = h[i]; if (h[i + 1] > t1) t1 = h[i + 1]; }
    // realtime counter: 100 MHz -> 10 ns ticks
    double span_us = (t1 - t0) * 0.01;
    std::vector<double> life, ph[5], start;
    for (size_t i = 0; i < h.size(); i += 8) {
        life.push_back((h[i + 1] - h[i]) * 0.01);
        start.push_back((h[i] - t0) * 0.01);
        for (int p = 0; p < 5; ++p) ph[p].push_back((double)(h[i + 3 + p] - h[i + 2 + p]));
    }
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    auto mx = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() - 1]; };
    printf("  timeline N=%d K=%d D=%d blocks=%d XT=%d: kernel span %.2f us; wave life median %.2f us max %.2f; start median %.2f max %.2f us\n",
           N, K, D, nblk, XT, span_us, med(life), mx(life), med(start), mx(start));
    printf("    phase cycles (median): issue-loads %.0f | transform+stage+barrier %.0f | steps %.0f | final barrier %.0f | store %.0f\n",
           med(ph[0]), med(ph[1]), med(ph[2]), med(ph[3]), med(ph[4]));
    {
        std::vector<unsigned long long> h2((size_t)nblk * NW * 4);
        CK(hipMemcpy(h2.data(), dbg2, h2.size() * 8, hipMemcpyDeviceToHost));
        std::vector<double> q[4];
        for (size_t i = 0; i < h2.size(); i += 4) for (int p = 0; p < 4; ++p) q[p].push_back((double)(long long)h2[i + p]);
        printf("    stage detail (median cycles): x arrives + sum of squares %.0f | x*gamma %.0f | staging writes %.0f | barrier %.0f\n", med(q[0]), med(q[1]), med(q[2]), med(q[3]));
    }
    CK(hipFree(dbg)); CK(hipFree(dbg2));
}

int main(int argc, char** argv) {
    const int L = 12;
    int shapes[4][2] = {{4096, 4096}, {11008, 4096}, {22016, 4096}, {4096, 11008}};
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
            CK(hipMalloc(&l.qw, (size_t)N * K / 2)); rnd(l.qw, (size_t)N * K / 2, 17);
            CK(hipMalloc(&l.sc, (size_t)K / 128 * N * 2)); rnd(l.sc, (size_t)K / 128 * N * 2, 3, 0x03ff03ffu, 0x1c001c00u);
            CK(hipMalloc(&l.sz, (size_t)K / 128 * N * 2)); rnd(l.sz, (size_t)K / 128 * N * 2, 5, 0x03ff03ffu, 0xa400a400u);
            CK(hipMalloc(&l.ow, (size_t)N * 128 * 2)); rnd(l.ow, (size_t)N * 128 * 2, 7, 0x83ff83ffu, 0x20002000u);
        }
        void *x, *y; CK(hipMalloc(&x, K * 2)); rnd(x, K * 2, 9, 0x83ff83ffu, 0x38003800u); CK(hipMalloc(&y, N * 2));
        double bytes = (double)N * (K - 128) / 2 + 2.0 * (K / 128) * N * 2 + (double)N * 128 * 2 + 2 * K + 2 * N;
        printf("N=%d K=%d algorithmic bytes %.0f\n", N, K, bytes);
        const int nb = N / 16 < 512 ? N / 16 : 256 * ((N / 16 + 384) / 768);
        run_mfma<8, 4, 0, 1>("mfma +rmsnorm D=4", Ls, x, y, N, K, bytes, nb);
        timeline<8, 4, 1>(Ls, x, y, N, K, nb);
        timeline<8, 4, 0>(Ls, x, y, N, K, nb);
        run_mfma<8, 4, 0, 0>("mfma plain D=4", Ls, x, y, N, K, bytes, nb);
        run_mfma<8, 4, 0, 2>("mfma silu*up D=4", Ls, x, y, N, K, bytes, nb);
        for (auto& l : Ls) { (void)hipFree(l.qw); (void)hipFree(l.sc); (void)hipFree(l.sz); (void)hipFree(l.ow); }
        (void)hipFree(x); (void)hipFree(y);
    }
    return 0;
}
