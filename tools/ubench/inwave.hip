// Micro-benchmark: fillers hidden per fp32 MFMA when the VALU work is in the SAME wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int NV, int KIND>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float a = threadIdx.x, b = 2.f;
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.001f - i;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    float& y = x[(i * NV + v) & 7];
                    if (KIND == 0) y = fmaf(y, 1.0001f, 0.5f);
                    else y = y > 0.f ? y - 1.5f : __expf(y) - 1.f;
                }
                __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                if (NV) __builtin_amdgcn_sched_group_barrier(0x2, KIND == 0 ? NV : NV * 6, 0);
            }
        a += 1.f;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float res = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) res += acc[i][0] + acc[i][3];
#pragma unroll
    for (int i = 0; i < 8; ++i) res += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = res;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NV, int KIND>
void run(float* out, unsigned long long* cyc) {
    const int iters = 300;
    hipLaunchKernelGGL((k<NV, KIND>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
    CK(hipDeviceSynchronize());
    unsigned long long h; CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    printf("kind=%s fillers/MFMA=%d : %6.1f cycles per MFMA\n", KIND ? "elu(exp)" : "fma", NV, (double)h / iters / 128);
}
int main() {
    float* out; unsigned long long* cyc; CK(hipMalloc(&out, 1024 * 256 * 4)); CK(hipMalloc(&cyc, 64));
    run<0, 0>(out, cyc); run<1, 0>(out, cyc); run<2, 0>(out, cyc); run<4, 0>(out, cyc); run<6, 0>(out, cyc); run<8, 0>(out, cyc); run<12, 0>(out, cyc);
    run<1, 1>(out, cyc); run<2, 1>(out, cyc);
    return 0;
}
