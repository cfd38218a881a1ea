// Small helper kernels: dense dequantisation and the outlier-slice interleave.
#include "qeft_common.h"

namespace qeft {

// Wdeq[N,K] fp16.  One thread per (row, 32-k chunk): reads the chunk's 16 bytes, writes 64 bytes.
// Role of the reference's uncompiled dequantize_weight_4bit_qeft (dequantize_cuda_qeft.cu:38-118).
__global__ __launch_bounds__(256) void dequant_w4_kernel(const uint8_t* qw, const f16* scales, const f16* zeros,
                                                         const f16* ow, f16* out, int N, int K, int G, int n_out) {
    const size_t chunks_per_rg = (size_t)K / 32 * 4;  // 16-byte pieces per row-group
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)(N / 4) * chunks_per_rg) return;
    const int rg = (int)(idx / chunks_per_rg);
    const int piece = (int)(idx % chunks_per_rg);  // tile*8 + r*2 + chunk
    const int tile = piece >> 3, r = (piece >> 1) & 3, c = piece & 1;
    const int row = rg * 4 + r, k0 = tile * 64 + c * 32;
    f16* dst = out + (size_t)row * K + k0;
    if (ow != nullptr && k0 >= K - n_out) {
        const u32x4* src = (const u32x4*)(ow + (size_t)row * n_out + (k0 - (K - n_out)));
#pragma unroll
        for (int j = 0; j < 4; ++j) ((u32x4*)dst)[j] = src[j];
        return;
    }
    const u32x4 v = *(const u32x4*)(qw + idx * 16);
    const int g = k0 / G;
    const h2 s = splat(scales[(size_t)g * N + row]), z = splat(zeros[(size_t)g * N + row]);
    uint32_t o[16];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        h2 wd[4];
        dequant8(v[w], s, z, wd);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[w + 4 * j] = as_u32(wd[j]);  // pair (2w+8j, 2w+8j+1) is dword w+4j
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) ((u32x4*)dst)[j] = u32x4{o[4 * j], o[4 * j + 1], o[4 * j + 2], o[4 * j + 3]};
}

// pack_oweight (qlinear.py:70-79): out[(n/8)*4 + n%4, (j/32)*64 + (j%32)*2 + (n%8)/4] = ow[n, j]
__global__ __launch_bounds__(256) void pack_oweight_kernel(const f16* ow, f16* il, int N, int R) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)N * R) return;
    const int n = (int)(idx / R), j = (int)(idx % R);
    il[((size_t)(n / 8) * 4 + n % 4) * (2 * R) + (j / 32) * 64 + (j % 32) * 2 + (n % 8) / 4] = ow[idx];
}

// Shadow of scales / scaled_zeros for the decode GEMV: out[(n/16)][g][n%16] = scale | scaled_zero << 16, so the 16
// rows x all groups a block needs are ONE contiguous run instead of K/G strided 32-byte pieces.
__global__ __launch_bounds__(256) void pack_scales_kernel(const uint16_t* scales, const uint16_t* zeros, uint32_t* out,
                                                          int N, int ngroups) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)N * ngroups) return;
    const int j = (int)(idx & 15);
    const int g = (int)((idx >> 4) % ngroups);
    const int b = (int)((idx >> 4) / ngroups);
    const size_t src = (size_t)g * N + b * 16 + j;
    out[idx] = (uint32_t)scales[src] | ((uint32_t)zeros[src] << 16);
}

hipError_t pack_scales_launch(const void* scales, const void* zeros, void* out, int N, int ngroups, hipStream_t st) {
    const size_t total = (size_t)N * ngroups;
    hipLaunchKernelGGL(pack_scales_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, st, (const uint16_t*)scales,
                       (const uint16_t*)zeros, (uint32_t*)out, N, ngroups);
    return hipGetLastError();
}

hipError_t dequant_w4_launch(const void* qw, const void* scales, const void* zeros, const void* ow, void* out, int N,
                             int K, int G, int n_out, hipStream_t st) {
    const size_t total = (size_t)(N / 4) * (K / 32 * 4);
    const int grid = (int)((total + 255) / 256);
    hipLaunchKernelGGL(dequant_w4_kernel, dim3(grid), dim3(256), 0, st, (const uint8_t*)qw, (const f16*)scales,
                       (const f16*)zeros, (const f16*)ow, (f16*)out, N, K, G, n_out);
    return hipGetLastError();
}

hipError_t pack_oweight_launch(const void* ow, void* il, int N, int R, hipStream_t st) {
    const size_t total = (size_t)N * R;
    hipLaunchKernelGGL(pack_oweight_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, st, (const f16*)ow,
                       (f16*)il, N, R);
    return hipGetLastError();
}

}  // namespace qeft
