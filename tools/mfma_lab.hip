// Lab: what one wave per SIMD gets out of the matrix pipe with the instruction patterns of the GEMM tile bodies.
//   mode 0: 8 accumulators, 4 back-to-back MFMAs per accumulator (the v3 GEMM's order), operands in registers
//   mode 1: 8 accumulators round-robin (no back-to-back dependency)
//   mode 2: mode 0 + 4 ds_read_b128 per 4 MFMAs feeding the A operand one phase later (the v3 GEMM's LDS traffic)
//   mode 3: mode 2 + 8 packed VALU ops per phase
//   mode 4: mode 0 with v_mfma_f32_16x16x32_f16 (16 accumulators of 4 registers)
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/mfma_lab.hip -o /tmp/mfma_lab && /tmp/mfma_lab
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void lab(float* out, int iters) {
    extern __shared__ __attribute__((aligned(1024))) uint8_t lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 32768 / 4; i += blockDim.x) ((uint32_t*)lds)[i] = 0x3c003c00u;     // fp16 1.0
    __syncthreads();
    const int r = lane & 31, h = lane >> 5;
    uint32_t a_rd[4];
    for (int j = 0; j < 4; ++j) a_rd[j] = (uint32_t)(r * 128 + (((h * 4 + j) ^ ((r >> 1) & 7)) << 4));
    u32x4 fa[4], fb[4], bc[4];
    for (int j = 0; j < 4; ++j) {
        fa[j] = *(const u32x4*)(lds + a_rd[j]);
        fb[j] = fa[j];
        bc[j] = u32x4{0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
    }
    h2 v[8];
    for (int j = 0; j < 8; ++j) v[j] = h2{(_Float16)(0.001f * lane), (_Float16)j};
    const h2 sc = {(_Float16)1.0001f, (_Float16)0.9999f}, zc = {(_Float16)0.5f, (_Float16)0.25f};
    if (MODE == 4) {
        f32x4 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int mt = 0; mt < 16; ++mt) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, fa[j]), __builtin_bit_cast(h8, bc[j]), acc[mt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        float s = 0;
        for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
        return;
    }
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mt = 0; mt < 8; ++mt)
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, fa[j]), __builtin_bit_cast(h8, bc[j]), acc[mt], 0, 0, 0);
        } else {
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) {
                u32x4(&cur)[4] = (mt & 1) ? fb : fa;
                u32x4(&nxt)[4] = (mt & 1) ? fa : fb;
                if (MODE >= 2) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) nxt[j] = *(const u32x4*)(lds + a_rd[j] + ((mt + 1) & 7) * 4096);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, cur[j]), __builtin_bit_cast(h8, bc[j]), acc[mt], 0, 0, 0);
                if (MODE == 3) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = __builtin_elementwise_fma(v[j], sc, zc);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][15];
    for (int j = 0; j < 8; ++j) s += (float)v[j][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int WAVES>
static void run(const char* name, float* out, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto k = lab<MODE, WAVES>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 32768);
    hipLaunchKernelGGL(k, dim3(256), dim3(64 * WAVES), 32768, 0, out, 16);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(64 * WAVES), 32768, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = 256.0 * WAVES * iters * 32 * 32768.0;      // 32 MFMAs of 32x32x16 (or 64 of 16x16x32) per iteration
    printf("%-44s waves/CU %d: %8.1f us  %7.1f TFLOP/s  (%.0f cycles per 32-MFMA group at 2.1 GHz)\n", name, WAVES, ms * 1e3,
           flops / ms / 1e9, ms * 1e-3 / iters * 2.1e9);
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 512 * 4);
    const int it = 4000;
    run<0, 4>("0: 4 back-to-back per accumulator", out, it);
    run<1, 4>("1: round-robin accumulators", out, it);
    run<2, 4>("2: back-to-back + 4 ds_read_b128 / phase", out, it);
    run<3, 4>("3: + 8 v_pk_fma / phase", out, it);
    run<4, 4>("4: 16x16x32, 4 back-to-back", out, it);
    run<0, 8>("0: 4 back-to-back per accumulator", out, it);
    run<1, 8>("1: round-robin accumulators", out, it);
    run<2, 8>("2: back-to-back + 4 ds_read_b128 / phase", out, it);
    run<3, 8>("3: + 8 v_pk_fma / phase", out, it);
    // short launches: does a ~60 us kernel see the same matrix rate as a 2 ms one?  (back-to-back, 3 times each)
    for (int rep = 0; rep < 3; ++rep)
        for (int n : {32, 64, 128, 256, 1024}) {
            char name[64];
            snprintf(name, sizeof name, "3: iters = %d", n);
            run<3, 4>(name, out, n);
        }
    return 0;
}
