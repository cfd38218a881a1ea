// Lab harness for the v3 decode GEMV (GPU box only): times kernel variants directly, cycling 12 weight sets per kind so that
// nothing is served from L2 / MALL, on the four launch kinds of a Llama-2-7B decoder layer.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-kernarg-preload-count=16 -I qeft_amd/csrc tools/gemv_v3_lab.hip -o build/gemv_v3_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>
#include <functional>
#include <algorithm>
// lab hooks of gemv_v3.h: time stamps of wave 0 of every block when the launch carries a dbg buffer
#define V3_STAMP_DECL long long ts[7] = {0, 0, 0, 0, 0, 0, 0};
#define V3_STAMP(i) do { if (a.dbg) ts[i] = wall_clock64(); } while (0)
#define V3_STAMP_VALUE(v) do { if (a.dbg) { asm volatile("" ::"v"(v)); ts[6] = wall_clock64(); } } while (0)
#define V3_STAMP_FLUSH() do { if (a.dbg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); if (tid == 0) { \
    for (int i_ = 0; i_ < 4; ++i_) a.dbg[(size_t)blockIdx.x * 8 + i_] = ts[i_]; \
    a.dbg[(size_t)blockIdx.x * 8 + 5] = ts[5]; a.dbg[(size_t)blockIdx.x * 8 + 6] = ts[6]; a.dbg[(size_t)blockIdx.x * 8 + 4] = wall_clock64(); \
    a.dbg[(size_t)blockIdx.x * 8 + 7] = __builtin_amdgcn_s_getreg(6164) & 15; } } } while (0)
#include "gemv_v3.h"

namespace qeft { thread_local const char* g_last_variant = ""; }
using namespace qeft;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void fill_random(uint32_t* p, size_t n, uint32_t seed, uint32_t andmask, uint32_t ormask) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t v = (uint32_t)i * 2654435761u ^ seed;
    v ^= v >> 16; v *= 0x85ebca6bu; v ^= v >> 13; v *= 0xc2b2ae35u; v ^= v >> 16;
    p[i] = (v & andmask) | ormask;
}
static void* dalloc(size_t bytes, uint32_t seed, uint32_t andmask = 0xffffffffu, uint32_t ormask = 0) {
    void* p; CK(hipMalloc(&p, (bytes + 255) / 256 * 256));
    size_t n = (bytes + 3) / 4;
    hipLaunchKernelGGL(fill_random, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (uint32_t*)p, n, seed, andmask, ormask);
    return p;
}

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

static int qeft_lab_blocks(int nsets) {     // gemv_v3_blocks (gemv_v3.hip)
    auto cd = [](int a, int b) { return (a + b - 1) / b; };
    int nblk;
    if (nsets <= 256) nblk = nsets;
    else if (nsets < 512) nblk = cd(nsets, 2);
    else { int k = (nsets + 384) / 768; if (k < 1) k = 1; nblk = 256 * k; }
    if (cd(nsets, nblk) > V3_MAX_RS) nblk = cd(nsets, V3_MAX_RS);
    if (nsets >= 512) nblk = cd(nsets, cd(nsets, nblk));
    return nblk;
}
struct Kind { const char* name; int n, k, mode; bool ssq, res; };
struct Case { std::string label; double bytes; int L; std::function<void(int)> f; std::vector<float> us; };
static std::vector<Case> g_cases;
struct Bufs { void *qw, *szp, *ow; };

template <int NW, int D, int MODE, int RSC, bool TAB = false>
static void launch_r(const V3Args& a, int nblk, size_t smem) {
    auto kern = gemv_v3_kernel<NW, D, true, MODE, 4, 1, RSC>;      // (TAB: round 4's tabulated correction sums, measured and removed from gemv_v3.h: profiles/r04_gemv_lab.txt)
    if (smem > 64 * 1024) CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(NW * 64), smem, 0, V3_KERNEL_ARGS(a));
}
// the lab's four launch kinds have 3 (q|k|v, gate|up) or 1 (o_proj, down_proj) row sets per block under the product's block rule
template <int NW, int D, int MODE, bool TAB = false>
static void launch(const V3Args& a, int nblk, size_t smem) {
    if (a.rs_cap == 3) launch_r<NW, D, MODE, 3>(a, nblk, smem);
    else if (a.rs_cap == 1) launch_r<NW, D, MODE, 1, TAB && MODE == V3_MODE_PLAIN>(a, nblk, smem);
    else { printf("    (rs_cap %d not instantiated in the lab)\n", a.rs_cap); }
}

static V3Args make_args(const Kind& kd, void* x, void* y, void* h32, void* gam, void* ssq, void* ynorm, int nblk, int nw) {
    const int nsets = kd.n / 16;
    V3Args a{};
    a.x = (const f16*)x;
    a.g = V3Geom{kd.k, 128, kd.k / 128, (kd.k - 128) / 128, kd.k / 128, nsets};
    a.rs_cap = (nsets + nblk - 1) / nblk; a.nblk = nblk; a.sets_q = nsets / nblk; a.sets_r = nsets % nblk;
    a.ssq_in = kd.ssq ? (const float*)ssq : nullptr; a.n_ssq_in = kd.ssq ? 256 : 0; a.eps = 1e-5f;
    a.residual = kd.res ? (const float*)h32 : nullptr; a.y32 = kd.res ? (float*)h32 : nullptr;
    a.gamma_out = kd.res ? (const f16*)gam : nullptr; a.ynorm = (f16*)ynorm; a.ssq_out = (float*)ssq + 512;
    a.y = (f16*)y; a.nw = nw;
    return a;
}

template <int NW, int D, bool TAB = false>
static void run(const Kind& kd, std::vector<Bufs>& B, void* x, void* y, void* h32, void* gam, void* ssq, void* ynorm, int nblk) {
    const int nsets = kd.n / 16;
    if (nblk > nsets) nblk = nsets;
    if ((nsets + nblk - 1) / nblk > V3_MAX_RS) { printf("    (skip: %d blocks need > %d sets per block)\n", nblk, V3_MAX_RS); return; }
    if (kd.mode == V3_MODE_PAIR && NW > 8) return;
    const V3Args a = make_args(kd, x, y, h32, gam, ssq, ynorm, nblk, NW);
    const size_t smem = v3_smem_bytes(kd.k, kd.k / 128, 128, a.rs_cap, false, 1, NW);
    auto f = [a, &B, &kd, nblk, smem](int l) {
        V3Args b = a; b.qw = (const uint8_t*)B[l].qw; b.szp = (const uint8_t*)B[l].szp; b.ow = (const uint8_t*)B[l].ow;
        if (kd.mode == V3_MODE_PAIR) { if constexpr (NW <= 8) launch<NW, D, V3_MODE_PAIR>(b, nblk, smem); }
        else launch<NW, D, V3_MODE_PLAIN, TAB>(b, nblk, smem);
    };
    const double bytes = (double)kd.n * (kd.k - 128) / 2 + 2.0 * (kd.k / 128) * kd.n * 2 + (double)kd.n * 128 * 2 + 2 * kd.k + 2 * kd.n;
    char label[96];
    snprintf(label, sizeof label, "%-4s NW=%2d D=%d blocks=%4d%s", kd.name, NW, D, nblk, TAB ? " TAB" : "    ");
    g_cases.push_back(Case{label, bytes, (int)B.size(), f, {}});
}
// every registered case timed `rounds` times, the cases interleaved (box-level drift hits all of them alike); min and median
static void measure_cases(int rounds) {
    for (int r = 0; r < rounds; ++r)
        for (auto& c : g_cases) c.us.push_back(time_launches(10, c.L, c.f));
    CK(hipGetLastError());
    for (auto& c : g_cases) {
        std::sort(c.us.begin(), c.us.end());
        const float mn = c.us.front(), md = c.us[c.us.size() / 2];
        printf("  %s : min %6.2f  median %6.2f us  %6.0f GB/s (median)\n", c.label.c_str(), mn, md, c.bytes / md / 1e3);
    }
    g_cases.clear();
}

// time stamps of wave 0 of every block (100 MHz ticks) -> where a launch's time goes
template <int NW, int D>
static void timeline(const Kind& kd, std::vector<Bufs>& B, void* x, void* y, void* h32, void* gam, void* ssq, void* ynorm, int nblk) {
    const V3Args a = make_args(kd, x, y, h32, gam, ssq, ynorm, nblk, NW);
    long long* dbg; CK(hipMalloc(&dbg, (size_t)nblk * 64 * 6));
    const size_t smem = v3_smem_bytes(kd.k, kd.k / 128, 128, a.rs_cap, false, 1, NW);
    std::vector<long long> h((size_t)nblk * 8 * 6);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int warm = 0; warm < 3; ++warm)
        for (int l = 0; l < 6; ++l) {
            V3Args b = a; b.qw = (const uint8_t*)B[l].qw; b.szp = (const uint8_t*)B[l].szp; b.ow = (const uint8_t*)B[l].ow;
            b.dbg = dbg + (size_t)l * nblk * 8;
            if (l == 0 && warm == 2) CK(hipEventRecord(e0, 0));
            if (kd.mode == V3_MODE_PAIR) { if constexpr (NW <= 8) launch<NW, D, V3_MODE_PAIR>(b, nblk, smem); } else launch<NW, D, V3_MODE_PLAIN>(b, nblk, smem);
        }
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
    printf("  %-4s NW=%2d D=%d blocks=%4d: %.2f us per launch (6 back to back, stamped); per launch, us after the first block's entry (min..max over blocks):\n", kd.name, NW, D, nblk, ms * 1e3 / 6);
    long long prev_end = 0;
    for (int l = 0; l < 6; ++l) {
        long long t0 = 1LL << 62, mx[7] = {0, 0, 0, 0, 0, 0, 0}, mn[7];
        for (int i = 0; i < 7; ++i) mn[i] = 1LL << 62;
        double tail[3] = {0, 0, 0};      // mean over blocks: wave 0's steps done -> barrier, barrier -> values ready, values ready -> stores acknowledged
        for (int b = 0; b < nblk; ++b) {
            const long long* p = &h[((size_t)l * nblk + b) * 8];
            if (p[0] < t0) t0 = p[0];
            for (int i = 0; i < 7; ++i) { if (p[i] < mn[i]) mn[i] = p[i]; if (p[i] > mx[i]) mx[i] = p[i]; }
            tail[0] += (p[5] - p[3]) / 100.0 / nblk; tail[1] += (p[6] - p[5]) / 100.0 / nblk; tail[2] += (p[4] - p[6]) / 100.0 / nblk;
        }
        if (l == 3) {
            printf("     launch %d: entry 0..%.2f | ring issued %.2f..%.2f | staging landed %.2f..%.2f | steps done %.2f..%.2f | end %.2f..%.2f", l,
                   (mx[0] - t0) / 100.0, (mn[1] - t0) / 100.0, (mx[1] - t0) / 100.0, (mn[2] - t0) / 100.0, (mx[2] - t0) / 100.0,
                   (mn[3] - t0) / 100.0, (mx[3] - t0) / 100.0, (mn[4] - t0) / 100.0, (mx[4] - t0) / 100.0);
            printf(" | tail per block: wait for the block's waves %.2f, epilogue math %.2f, store + ack %.2f | gap since previous end %.2f\n", tail[0], tail[1], tail[2], (t0 - prev_end) / 100.0);
        }
        prev_end = mx[4];
    }
    CK(hipFree(dbg));
}

int main() {
    const int L = 12;
    uint32_t* out; CK(hipMalloc(&out, 4096));
    {   // streaming ceiling at the four launch sizes
        size_t tot = (size_t)L * 56 * 1024 * 1024;
        void* buf; CK(hipMalloc(&buf, tot)); CK(hipMemset(buf, 1, tot));
        for (double mb : {9.3, 23.6, 27.8, 49.8}) {
            size_t bytes = (size_t)(mb * 1024 * 1024) / 16384 * 16384;
            for (int grid : {512, 1024}) {
                size_t vpb = bytes / 16 / grid;
                auto f8 = [&](int l) { hipLaunchKernelGGL(stream_read<8>, dim3(grid), dim3(256), 0, 0, (const u32x4*)((char*)buf + (size_t)l * 56 * 1024 * 1024), vpb, out); };
                float u8 = time_launches(20, L, f8);
                printf("stream %5.1f MiB grid %4d: %6.2f us %6.0f GB/s\n", mb, grid, u8, bytes / u8 / 1e3);
            }
        }
        CK(hipFree(buf));
    }
    // "d2": down_proj's bytes as 512 blocks of one row set and half the K (what a 2-way split-K launch would stream, without its
    // combine step): LAB_SPLITK=1
    const bool splitk = getenv("LAB_SPLITK") != nullptr;
    const Kind kinds[4] = {{"qkv", 12288, 4096, V3_MODE_PLAIN, true, false}, {"o", 4096, 4096, V3_MODE_PLAIN, false, true},
                           {"gu", 22016, 4096, V3_MODE_PAIR, true, false},
                           splitk ? Kind{"d2", 8192, 5504, V3_MODE_PLAIN, false, true} : Kind{"d", 4096, 11008, V3_MODE_PLAIN, false, true}};
    void* h32 = dalloc(4096 * 4, 1, 0x807fffffu, 0x3f000000u);
    void* gam = dalloc(4096 * 2, 2, 0x03ff03ffu, 0x3c003c00u);
    void* ssq = dalloc(2048 * 4, 3, 0x007fffffu, 0x3f800000u);
    void* ynorm = dalloc(4096 * 2, 4);
    const bool full = getenv("LAB_FULL") != nullptr;
    for (const Kind& kd : kinds) {
        std::vector<Bufs> B(L);
        for (auto& b : B) {
            b.qw = dalloc((size_t)kd.n * kd.k / 2, 17);
            b.szp = dalloc((size_t)kd.n * (kd.k / 128) * 4, 3, 0x03ff03ffu, 0xa4001c00u);
            b.ow = dalloc((size_t)kd.n * 128 * 2, 7, 0x83ff83ffu, 0x20002000u);
        }
        void* x = dalloc(kd.k * 2, 9, 0x83ff83ffu, 0x38003800u);
        void* y = dalloc(kd.n * 2, 5);
        printf("%s: n=%d k=%d\n", kd.name, kd.n, kd.k);
        const int nsets = kd.n / 16;
        const int nb = (splitk && kd.n == 8192) ? 512 : qeft_lab_blocks(nsets);
        run<8, 2>(kd, B, x, y, h32, gam, ssq, ynorm, nb);
        run<8, 4>(kd, B, x, y, h32, gam, ssq, ynorm, nb);
        run<8, 6>(kd, B, x, y, h32, gam, ssq, ynorm, nb);
        if (kd.mode != V3_MODE_PAIR && !kd.ssq) {      // one row set per block: step balance (K = 11008: 85 full steps) and tabulated correction sums
            run<12, 2>(kd, B, x, y, h32, gam, ssq, ynorm, nb);
            run<12, 4>(kd, B, x, y, h32, gam, ssq, ynorm, nb);
        }
        measure_cases(7);
        if (full) {
            timeline<8, 4>(kd, B, x, y, h32, gam, ssq, ynorm, nb);
        }
        for (auto& b : B) { (void)hipFree(b.qw); (void)hipFree(b.szp); (void)hipFree(b.ow); }
        (void)hipFree(x); (void)hipFree(y);
    }
    return 0;
}
