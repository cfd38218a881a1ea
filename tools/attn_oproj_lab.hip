// Lab harness for the fused attention + o_proj launch (GPU box only): times the launch over 12 weight sets (nothing served from
// L2 / MALL), the stand-alone attention kernel beside it, and prints where wave 0 of the blocks spends the launch.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-kernarg-preload-count=16 -I qeft_amd/csrc -I tools tools/attn_oproj_lab.hip -o build/attn_oproj_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#define AO_STAMP_DECL long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define AO_STAMP(i) do { if (a.dbg) ts[i] = wall_clock64(); } while (0)
#define AO_STAMP_FLUSH() do { if (a.dbg && tid == 0) { for (int i_ = 0; i_ < 7; ++i_) a.dbg[(size_t)blockIdx.x * 8 + i_] = ts[i_]; \
    a.dbg[(size_t)blockIdx.x * 8 + 7] = wall_clock64(); } } while (0)
#define AO_LAB_ARGS unsigned long long* dbg;
#include "attn_oproj_lab_kernel.h"

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

int main(int argc, char** argv) {
    const int n_heads = argc > 1 ? atoi(argv[1]) : 32, n_kv = n_heads, hidden = n_heads * 128, K = hidden, max_seq = 2048;
    const int position = argc > 2 ? atoi(argv[2]) : 200;
    const int L = 12;
    const uint32_t H = 0x33ff33ffu;             // fp16 pairs below 0.25
    const int nsets = hidden / 16, nblk = nsets < 256 ? nsets : 256;
    if (!attn_oproj_supported(n_heads, n_kv, max_seq, hidden, K, 128, 128)) { printf("shape not supported\n"); return 1; }
    void *q = dalloc(K * 2, 1, H), *k = dalloc(K * 2, 2, H), *v = dalloc(K * 2, 3, H);
    void *cs = dalloc(64 * 4, 4, 0x007fffffu, 0x3f000000u), *sn = dalloc(64 * 4, 5, 0x007fffffu, 0x3e000000u);
    void *kc = dalloc((size_t)n_kv * max_seq * 128 * 2, 6, H), *vc = dalloc((size_t)n_kv * max_seq * 128 * 2, 7, H);
    int* pos; CK(hipMalloc(&pos, 4)); CK(hipMemcpy(pos, &position, 4, hipMemcpyHostToDevice));
    std::vector<void*> qw(L), szp(L), ow(L);
    for (int l = 0; l < L; ++l) {
        qw[l] = dalloc((size_t)nsets * K * 8, 10 + l);
        szp[l] = dalloc((size_t)nsets * v3_sz_bytes(K / 128) * 2, 30 + l, H);
        ow[l] = dalloc((size_t)nsets * 4096, 50 + l, H);
    }
    void *h32 = dalloc(hidden * 4, 70, 0x007fffffu, 0x3f000000u), *gamma = dalloc(hidden * 2, 71, H), *ynorm = dalloc(hidden * 2, 72, H);
    void *ssq = dalloc(1024 * 4, 73), *xatt = dalloc(K * 2, 74, H), *att_out = dalloc(K * 2, 75, H);
    uint32_t* state; CK(hipMalloc(&state, 16)); CK(hipMemset(state, 0, 16));
    unsigned long long* dbg; CK(hipMalloc(&dbg, 256 * 8 * 8)); CK(hipMemset(dbg, 0, 256 * 8 * 8));
    CK(hipDeviceSynchronize());

    const size_t smem = ao_lds(K, max_seq).total;
    auto kern = attn_oproj_kernel<4, 2>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    auto args = [&](int l, bool stamps) {
        AoArgs a{};
        a.pos = pos; a.out_pos = nullptr; a.q = (const f16*)q; a.k = (const f16*)k; a.v = (const f16*)v; a.cs = (const float*)cs; a.sn = (const float*)sn;
        a.heads_kv_s_tab = (uint32_t)n_heads | ((uint32_t)n_kv << 12) | (1u << 24) | (1u << 28);
        a.kc = (f16*)kc; a.vc = (f16*)vc; a.max_seq = max_seq; a.n_attn_blocks = n_heads * 2;
        a.qw = (const uint8_t*)qw[l]; a.szp = (const uint8_t*)szp[l]; a.ow = (const uint8_t*)ow[l];
        a.h32 = (float*)h32; a.gamma_out = (const f16*)gamma; a.ynorm = (f16*)ynorm; a.ssq_out = (float*)ssq;
        a.K = K; a.nsets = nsets; a.nblk = nblk; a.xatt = (f16*)xatt; a.state = state; a.timeout_ticks = 100u * 1000u * 50u;
        a.dbg = stamps ? dbg : nullptr;
        return a;
    };
    auto fused = [&](int l) { hipLaunchKernelGGL(kern, dim3(nblk), dim3(AO_NW * 64), smem, 0, args(l, false)); };
    auto attn = [&](int) {
        hipLaunchKernelGGL((rope_attn_decode_kernel<4, 2>), dim3(n_heads * 2), dim3(256), rope_attn_smem_bytes(max_seq), 0, (const int*)pos, (const int*)nullptr,
                           (const f16*)q, (const f16*)k, (const f16*)v, (const float*)cs, (uint32_t)n_heads | ((uint32_t)n_kv << 12) | (1u << 24) | (1u << 28),
                           (const float*)sn, (f16*)kc, (f16*)vc, (f16*)att_out, (float*)nullptr, max_seq, (unsigned long long*)nullptr, (const float*)nullptr);
    };
    printf("heads %d  hidden %d  position %d  lds %zu\n", n_heads, hidden, position, smem);
    for (int rep = 0; rep < 3; ++rep) {
        printf("fused attention + o_proj : %7.2f us / launch\n", time_launches(200, L, fused));
        printf("attention alone          : %7.2f us / launch\n", time_launches(200, L, attn));
    }
    uint32_t st[4]; CK(hipMemcpy(st, state, 16, hipMemcpyDeviceToHost));
    printf("state: seq %u status %#x counted %u\n", st[0], st[2], st[3]);
    // ---- one stamped launch (behind a warm one)
    for (int l = 0; l < 3; ++l) fused(l);
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(AO_NW * 64), smem, 0, args(3, true));
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> d(256 * 8);
    CK(hipMemcpy(d.data(), dbg, 256 * 8 * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull;
    for (int b = 0; b < nblk; ++b) t0 = std::min(t0, d[b * 8]);
    const char* names[8] = {"entry", "requested", "body done", "count seen", "released", "x staged", "steps done", "end"};
    for (int cls = 0; cls < 2; ++cls) {
        printf("%s blocks: phase = mean [min .. max] us after the first block's entry\n", cls == 0 ? "attention" : "other");
        for (int i = 0; i < 8; ++i) {
            double s = 0, mn = 1e9, mx = -1; int n = 0;
            for (int b = 0; b < nblk; ++b) {
                if ((b < n_heads * 2) != (cls == 0)) continue;
                const double us = (double)(d[b * 8 + i] - t0) / 100.0;
                s += us; mn = std::min(mn, us); mx = std::max(mx, us); ++n;
            }
            if (n) printf("  %-11s %6.2f [%6.2f .. %6.2f]\n", names[i], s / n, mn, mx);
        }
    }
    return 0;
}
