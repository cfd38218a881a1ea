// Lab: what does a flag hand-off between two kernels running CONCURRENTLY on two streams cost on MI355X, and is the
// data it publishes seen fresh across the 8 XCDs' L2s?  (The price of overlapping consecutive decode GEMV launches.)
//   producer (stream A): every block idles ~work_us, writes its slice of x (value = epoch), releases, counts itself in.
//   consumer (stream B, launched right behind): every block stamps its entry, spins on the counter (bounded), acquires,
//   reads ALL of x and counts values that are not this epoch's.
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/handoff_lab.hip -o /tmp/handoff_lab && /tmp/handoff_lab
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int NB = 256, XN = 4096;      // blocks, x elements (u32)

// RELAXED = true: the cheapest legal form -- one release fence, relaxed counter traffic, one acquire fence after the spin
template <bool RELAXED>
__global__ __launch_bounds__(256) void producer(uint32_t* x, unsigned* done, long long* st, uint32_t epoch, int work_ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < work_ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x < XN / NB) x[blockIdx.x * (XN / NB) + threadIdx.x] = epoch;
    __syncthreads();
    if (threadIdx.x == 0) {
        if (RELAXED) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        st[blockIdx.x * 2] = t0;
        st[blockIdx.x * 2 + 1] = wall_clock64();
    }
}

template <bool RELAXED>
__global__ __launch_bounds__(256) void consumer(const uint32_t* x, unsigned* done, long long* st, unsigned* bad, uint32_t epoch,
                                                unsigned expect) {
    const long long t0 = wall_clock64();
    if (threadIdx.x == 0) {
        int it = 0;
        if (RELAXED)
            while (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < expect && ++it < (1 << 22)) __builtin_amdgcn_s_sleep(1);
        else
            while (__hip_atomic_load(done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < expect && ++it < (1 << 22)) __builtin_amdgcn_s_sleep(1);
        if (it >= (1 << 22)) atomicAdd(bad + 1, 1u);      // gave up: never hang the box
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const long long t1 = wall_clock64();
    unsigned wrong = 0;
    for (int i = threadIdx.x; i < XN; i += 256) wrong += x[i] != epoch;
    if (wrong) atomicAdd(bad, wrong);
    if (threadIdx.x == 0) {
        st[blockIdx.x * 2] = t0;
        st[blockIdx.x * 2 + 1] = t1;
    }
}

int main() {
    uint32_t* x; unsigned *done, *bad; long long *sp, *sc;
    CK(hipMalloc(&x, XN * 4)); CK(hipMalloc(&done, 64)); CK(hipMalloc(&bad, 64));
    CK(hipMalloc(&sp, NB * 16)); CK(hipMalloc(&sc, NB * 16));
    CK(hipMemset(x, 0, XN * 4)); CK(hipMemset(done, 0, 64)); CK(hipMemset(bad, 0, 64));
    hipStream_t A, B; CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
    std::vector<long long> hp(NB * 2), hc(NB * 2);
    for (int mode : {0, 1})
    for (int work_us : {6, 12}) {
        for (int rep = 0; rep < 4; ++rep) {
            static unsigned epoch = 0;
            ++epoch;
            if (mode) {
                hipLaunchKernelGGL(producer<true>, dim3(NB), dim3(256), 0, A, x, done, sp, epoch, work_us * 100);
                hipLaunchKernelGGL(consumer<true>, dim3(NB), dim3(256), 0, B, (const uint32_t*)x, done, sc, bad, epoch, epoch * NB);
            } else {
                hipLaunchKernelGGL(producer<false>, dim3(NB), dim3(256), 0, A, x, done, sp, epoch, work_us * 100);
                hipLaunchKernelGGL(consumer<false>, dim3(NB), dim3(256), 0, B, (const uint32_t*)x, done, sc, bad, epoch, epoch * NB);
            }
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(hp.data(), sp, NB * 16, hipMemcpyDeviceToHost));
            CK(hipMemcpy(hc.data(), sc, NB * 16, hipMemcpyDeviceToHost));
            unsigned hb[2]; CK(hipMemcpy(hb, bad, 8, hipMemcpyDeviceToHost));
            long long p0 = 1LL << 62, p1 = 0, c0min = 1LL << 62, c0max = 0, c1min = 1LL << 62, c1max = 0;
            for (int b = 0; b < NB; ++b) {
                p0 = std::min(p0, hp[2 * b]); p1 = std::max(p1, hp[2 * b + 1]);
                c0min = std::min(c0min, hc[2 * b]); c0max = std::max(c0max, hc[2 * b]);
                c1min = std::min(c1min, hc[2 * b + 1]); c1max = std::max(c1max, hc[2 * b + 1]);
            }
            printf("%s work %2d us: producer %.2f us (first entry -> last signal); consumer blocks entered %.2f..%.2f us after the producer's first entry; "
                   "saw the flag %.2f..%.2f us after the last signal; stale words %u, give-ups %u\n", mode ? "relaxed" : "acq/rel", work_us, (p1 - p0) / 100.0,
                   (c0min - p0) / 100.0, (c0max - p0) / 100.0, (c1min - p1) / 100.0, (c1max - p1) / 100.0, hb[0], hb[1]);
        }
    }
    return 0;
}
