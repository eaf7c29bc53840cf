// Micro-benchmark: how fast does a VALU-only wave run next to a wave that issues fp32 MFMAs
// back-to-back on the same SIMD?  Block = 8 waves: waves 0-3 MFMA (one per SIMD), waves 4-7 VALU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// mode bit0: MFMA waves active, bit1: VALU waves active; valu_kind 0: dependent fma chain, 1: 4 independent chains, 2: exp-based ELU on 4 values
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters, int mode, int valu_kind, int prio) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float res = 0.f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) {
        if (mode & 1) {
            f32x4 acc[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
            float a = threadIdx.x, b = 2.f;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
                a += 1.f;
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) res += acc[i][0] + acc[i][3];
        }
    } else {
        if (mode & 2) {
            if (prio) __builtin_amdgcn_s_setprio(3);
            float x0 = threadIdx.x * 0.001f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
            for (int it = 0; it < iters; ++it) {
                if (valu_kind == 0) {
#pragma unroll
                    for (int u = 0; u < 128; ++u) x0 = fmaf(x0, 1.0001f, 0.5f);
                } else if (valu_kind == 1) {
#pragma unroll
                    for (int u = 0; u < 32; ++u) { x0 = fmaf(x0, 1.0001f, 0.5f); x1 = fmaf(x1, 1.0001f, 0.5f); x2 = fmaf(x2, 1.0001f, 0.5f); x3 = fmaf(x3, 1.0001f, 0.5f); }
                } else {
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        x0 = x0 > 0.f ? x0 - 1.5f : __expf(x0) - 1.f; x1 = x1 > 0.f ? x1 - 1.5f : __expf(x1) - 1.f;
                        x2 = x2 > 0.f ? x2 - 1.5f : __expf(x2) - 1.f; x3 = x3 > 0.f ? x3 - 1.5f : __expf(x3) - 1.f;
                    }
                }
            }
            res = x0 + x1 + x2 + x3;
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 512 + threadIdx.x] = res;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}
int main() {
    float* out; unsigned long long* cyc; CK(hipMalloc(&out, 1024 * 512 * 4)); CK(hipMalloc(&cyc, 64));
    const int iters = 500;
    for (int kind = 0; kind < 3; ++kind)
        for (int prio = 0; prio < 2; ++prio)
            for (int mode : {1, 2, 3}) {
                CK(hipMemset(cyc, 0, 64));
                hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, cyc, iters, mode, kind, prio);
                CK(hipDeviceSynchronize());
                unsigned long long h[8]; CK(hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost));
                printf("valu_kind=%d prio=%d mode=%d | mfma wave cycles/iter %8.1f (ideal 4096) | valu wave cycles/iter %8.1f\n", kind, prio, mode,
                       (double)h[0] / iters, (double)h[4] / iters);
            }
    return 0;
}
