// Lab: issue rate of the MFMA instructions the kernels use (cycles per instruction per wave, by waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(256) void mfma_rate(float* out, unsigned long long* cyc, int iters) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.01f); }
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    f32x4 d0 = {}, d1 = {}, d2 = {}, d3 = {};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
        } else {
            d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d1, 0, 0, 0);
            d2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d2, 0, 0, 0);
            d3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d3, 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    for (int i = 0; i < 4; ++i) s += d0[i] + d1[i] + d2[i] + d3[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 1024 * 4 * 8);
    unsigned long long h[4096];
    const int iters = 2000;
    for (int kind = 0; kind < 2; ++kind)
        for (int blocks_per_cu = 1; blocks_per_cu <= 2; ++blocks_per_cu) {
            const int nb = 256 * blocks_per_cu;     // 4 waves per block: 1 or 2 waves per SIMD
            for (int r = 0; r < 2; ++r) {
                if (kind == 0) hipLaunchKernelGGL(mfma_rate<0>, dim3(nb), dim3(256), 0, 0, out, cyc, iters);
                else hipLaunchKernelGGL(mfma_rate<1>, dim3(nb), dim3(256), 0, 0, out, cyc, iters);
                hipDeviceSynchronize();
            }
            hipMemcpy(h, cyc, nb * 4 * 8, hipMemcpyDeviceToHost);
            double s = 0; for (int i = 0; i < nb * 4; ++i) s += h[i];
            const double per = s / (nb * 4) / (iters * 4.0);
            const double flops = kind == 0 ? 32768.0 : 16384.0;
            printf("%s, %d wave(s) per SIMD: %.1f cycles per MFMA per wave -> %.0f flops/cycle/SIMD -> %.2f PFLOP/s at 2.4 GHz x 1024 SIMDs\n",
                   kind == 0 ? "v_mfma_f32_32x32x16_f16" : "v_mfma_f32_16x16x32_f16", blocks_per_cu, per,
                   flops * blocks_per_cu / per, flops * blocks_per_cu / per * 2.4e9 * 1024 / 1e15);
        }
    return 0;
}
