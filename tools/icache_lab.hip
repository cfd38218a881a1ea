// Lab: how fast does straight-line code execute when the instruction cache is cold?  (MI355X)
// K<N>: N dependent v_add_f32 (4 bytes each) bracketed by s_memtime; one wave per block.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int VARIANT>
__global__ __launch_bounds__(64) void straight(float* out, unsigned long long* cyc) {
    float x = out[threadIdx.x];
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
    if (VARIANT == 0) asm volatile(".rept 2000\n v_add_f32 %0, %0, %0\n .endr" : "+v"(x));
    if (VARIANT == 1) asm volatile(".rept 2000\n v_mul_f32 %0, %0, %0\n .endr" : "+v"(x));
    if (VARIANT == 2) asm volatile(".rept 14000\n v_max_f32 %0, %0, %0\n .endr" : "+v"(x));   // 56 KB evictor
    if (VARIANT == 3) asm volatile(".rept 250\n v_add_f32 %0, %0, %0\n .endr" : "+v"(x));     // 1 KB
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x + 64 * (blockIdx.x & 1)] = x;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    float* out; unsigned long long* cyc;
    const int NB = 256;
    CK(hipMalloc(&out, 1024)); CK(hipMemset(out, 0, 1024));
    CK(hipMalloc(&cyc, NB * 8));
    std::vector<unsigned long long> h(NB);
    auto report = [&](const char* what, int n) {
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h.data(), cyc, NB * 8, hipMemcpyDeviceToHost);
        double s = 0; unsigned long long mx = 0, mn = ~0ull;
        for (auto v : h) { s += v; mx = v > mx ? v : mx; mn = v < mn ? v : mn; }
        printf("%-58s mean %8.0f cyc  min %6llu max %6llu  -> %.2f cyc/instr\n", what, s / NB, mn, mx, s / NB / n);
    };
    hipLaunchKernelGGL(straight<0>, dim3(NB), dim3(64), 0, 0, out, cyc); report("K0 (2000 v_add) first launch ever", 2000);
    hipLaunchKernelGGL(straight<0>, dim3(NB), dim3(64), 0, 0, out, cyc); report("K0 again (same code, back to back)", 2000);
    hipLaunchKernelGGL(straight<0>, dim3(NB), dim3(64), 0, 0, out, cyc); report("K0 again", 2000);
    hipLaunchKernelGGL(straight<1>, dim3(NB), dim3(64), 0, 0, out, cyc); report("K1 (2000 v_mul, other code) first", 2000);
    hipLaunchKernelGGL(straight<0>, dim3(NB), dim3(64), 0, 0, out, cyc); report("K0 after K1 (both fit in 64 KB)", 2000);
    hipLaunchKernelGGL(straight<2>, dim3(NB), dim3(64), 0, 0, out, cyc); report("K2 (14000 v_max = 56 KB) first", 14000);
    hipLaunchKernelGGL(straight<2>, dim3(NB), dim3(64), 0, 0, out, cyc); report("K2 again", 14000);
    hipLaunchKernelGGL(straight<0>, dim3(NB), dim3(64), 0, 0, out, cyc); report("K0 after K2 (evicted?)", 2000);
    hipLaunchKernelGGL(straight<0>, dim3(NB), dim3(64), 0, 0, out, cyc); report("K0 again", 2000);
    hipLaunchKernelGGL(straight<3>, dim3(NB), dim3(64), 0, 0, out, cyc); report("K3 (250 v_add) first", 250);
    hipLaunchKernelGGL(straight<3>, dim3(NB), dim3(64), 0, 0, out, cyc); report("K3 again", 250);
    // stream of alternating kernels without host sync in between (like a decode layer)
    for (int r = 0; r < 3; ++r) {
        hipLaunchKernelGGL(straight<1>, dim3(NB), dim3(64), 0, 0, out, cyc);
        hipLaunchKernelGGL(straight<2>, dim3(NB), dim3(64), 0, 0, out, cyc);
        hipLaunchKernelGGL(straight<0>, dim3(NB), dim3(64), 0, 0, out, cyc);
    }
    report("K0 at the end of K1,K2,K0 x3 without host syncs", 2000);
    for (int r = 0; r < 3; ++r) {
        hipLaunchKernelGGL(straight<1>, dim3(NB), dim3(64), 0, 0, out, cyc);
        hipLaunchKernelGGL(straight<0>, dim3(NB), dim3(64), 0, 0, out, cyc);
    }
    report("K0 at the end of K1,K0 x3 without host syncs", 2000);
    return 0;
}
