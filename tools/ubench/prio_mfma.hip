// Micro-benchmark: does s_setprio steer the fp32 MFMA pipe between two waves of one SIMD?
// Block = 8 waves (2 per SIMD): waves 0-3 at priority pa, waves 4-7 at priority pb, all stream MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
template <int PA, int PB>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave < 4) __builtin_amdgcn_s_setprio(PA); else __builtin_amdgcn_s_setprio(PB);
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float a = threadIdx.x, b = 2.f;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        a += 1.f;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float res = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) res += acc[i][0] + acc[i][3];
    out[blockIdx.x * 512 + threadIdx.x] = res;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}
template <int PA, int PB>
void run(float* out, unsigned long long* cyc) {
    const int iters = 200;
    hipLaunchKernelGGL((k<PA, PB>), dim3(256), dim3(512), 0, 0, out, cyc, iters);
    CK(hipDeviceSynchronize());
    unsigned long long h[8]; CK(hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost));
    printf("prio A=%d B=%d : cycles per 128 MFMAs  waveA(0) %7.0f  waveB(4) %7.0f   (alone = 4096)\n", PA, PB, (double)h[0] / iters, (double)h[4] / iters);
}
int main() {
    float* out; unsigned long long* cyc; CK(hipMalloc(&out, 1024 * 512 * 4)); CK(hipMalloc(&cyc, 64));
    run<0, 0>(out, cyc); run<0, 1>(out, cyc); run<1, 0>(out, cyc); run<0, 3>(out, cyc); run<3, 0>(out, cyc);
    return 0;
}
