// Micro-benchmark: fp32 MFMA issue rate on gfx950 with the accumulator/operand pattern of tap_gemm.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        a += 1.0f;
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0;
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        a += 1.0f;
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename F>
void run(const char* name, F launch, double flops_per_block_iter, int blocks, int iters) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(blocks, 10);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(e0)); launch(blocks, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("%-28s blocks=%5d  %8.3f ms  %7.1f TF/s\n", name, blocks, best, flops_per_block_iter * blocks * iters / best / 1e9);
}
int main() {
    float* out; CK(hipMalloc(&out, 4096 * 256 * 4));
    const int iters = 2000;
    for (int bpc : {1, 2, 4}) {
        int blocks = 256 * bpc;
        run("16x16x4 acc=16", [&](int g, int it) { hipLaunchKernelGGL(k16<16>, dim3(g), dim3(256), 0, 0, out, it, 1.f, 2.f); }, 4.0 * 8 * 16 * 2048, blocks, iters);
        run("16x16x4 acc=4", [&](int g, int it) { hipLaunchKernelGGL(k16<4>, dim3(g), dim3(256), 0, 0, out, it, 1.f, 2.f); }, 4.0 * 8 * 4 * 2048, blocks, iters);
        run("16x16x4 acc=2", [&](int g, int it) { hipLaunchKernelGGL(k16<2>, dim3(g), dim3(256), 0, 0, out, it, 1.f, 2.f); }, 4.0 * 8 * 2 * 2048, blocks, iters);
        run("32x32x2 acc=4", [&](int g, int it) { hipLaunchKernelGGL(k32<4>, dim3(g), dim3(256), 0, 0, out, it, 1.f, 2.f); }, 4.0 * 8 * 4 * 4096, blocks, iters);
        run("32x32x2 acc=1", [&](int g, int it) { hipLaunchKernelGGL(k32<1>, dim3(g), dim3(256), 0, 0, out, it, 1.f, 2.f); }, 4.0 * 8 * 1 * 4096, blocks, iters);
    }
    return 0;
}
