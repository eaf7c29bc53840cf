// Round 5: what does the distance between DEPENDENT v_mfma_f32_32x32x16_f16 (same accumulator) cost when other instructions sit
// between the MFMAs -- the shape of tap_gemm8's stage (units of 6 MFMAs on 2 accumulators = dependent distance 2, with LDS fragment
// reads, a few VALU and an occasional vector-memory request between the units), two waves per SIMD?
//   PATTERN 1: A A A B B B      (distance 1)      PATTERN 2: A B A B A B   (distance 2: what hipcc emits for tap_gemm8 today)
//   PATTERN 4: A B C D x 3      (distance 4)      PATTERN 8: 8 accumulators x 3 (distance 8)
//   FILL 0: MFMAs only;  1: + 3 ds_read_b128 per 6 MFMAs feeding the next operands;  2: + 8 VALU per 6 MFMAs as well
// Prints shader cycles per MFMA per SIMD (32 = the matrix pipe's rate).  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 mfma_dep_distance.hip -o mfma_dep_bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PATTERN, int FILL>
__global__ __launch_bounds__(512, 1) void k(float* out, int iters, unsigned long long* clk) {
    __shared__ __attribute__((aligned(16))) _Float16 lds[8 * 64 * 8 * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned s = (blockIdx.x * 512u + tid) * 2654435761u + 12345u;
    for (int i = tid; i < 8 * 64 * 8 * 4; i += 512) { s = s * 1664525u + 1013904223u; lds[i] = (_Float16)(((s >> 8) & 0xffff) / 32768.0f - 1.0f); }
    __syncthreads();
    f16x8 a[2], b[2];
    for (int i = 0; i < 8; ++i) { a[0][i] = lds[lane * 8 + i]; a[1][i] = lds[512 + lane * 8 + i]; b[0][i] = lds[1024 + lane * 8 + i]; b[1][i] = lds[1536 + lane * 8 + i]; }
    f32x16 acc[8];
    for (int n = 0; n < 8; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    float v0 = (float)lane, v1 = 1.f, v2 = 2.f, v3 = 3.f;
    const _Float16* base = lds + wave * 2048 + lane * 8;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        // one "stage": 48 MFMAs as 8 groups of 6
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            f16x8 na0, na1, nb0;
            if (FILL >= 1) {
                na0 = *reinterpret_cast<const f16x8*>(base + ((g & 3) * 512));
                na1 = *reinterpret_cast<const f16x8*>(base + (((g + 1) & 3) * 512));
                nb0 = *reinterpret_cast<const f16x8*>(base + (((g + 2) & 3) * 512));
            }
#pragma unroll
            for (int m = 0; m < 6; ++m) {
                int n;
                if (PATTERN == 1) n = (2 * g + m / 3) & 7;
                else if (PATTERN == 2) n = (2 * g + (m & 1)) & 7;
                else if (PATTERN == 4) n = (4 * (g >> 1) + ((g & 1) * 6 + m) % 4) & 7;
                else n = (g * 6 + m) & 7;
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m & 1], b[(m >> 1) & 1], acc[n], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (FILL >= 2 && m < 4) {   // 8 VALU per group
                    v0 = __builtin_fmaf(v0, v1, v2); v1 = __builtin_fmaf(v1, v2, v3);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (FILL >= 1) { a[0] = na0; a[1] = na1; b[0] = nb0; }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float t = v0 + v1;
    for (int n = 0; n < 8; ++n) t += acc[n][0];
    if (t == 12345.f) out[0] = t;
    if (lane == 0) clk[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int PATTERN, int FILL>
void run(int threads = 512) {
    float* out; CK(hipMalloc(&out, 4));
    unsigned long long* clk; CK(hipMalloc(&clk, 256 * 8 * 8));
    const int iters = 400;
    hipLaunchKernelGGL((k<PATTERN, FILL>), dim3(256), dim3(threads), 0, 0, out, 10, clk);
    hipLaunchKernelGGL((k<PATTERN, FILL>), dim3(256), dim3(threads), 0, 0, out, iters, clk);
    CK(hipDeviceSynchronize());
    unsigned long long hc[256 * 8]; CK(hipMemcpy(hc, clk, sizeof hc, hipMemcpyDeviceToHost));
    const int waves = threads / 64;
    double st = 0; for (int b = 0; b < 256; ++b) for (int w = 0; w < waves; ++w) st += hc[b * 8 + w];
    st /= 256 * waves;
    // per SIMD: waves / 4 waves x 48 MFMAs per iteration
    printf("%d wave(s) per SIMD, distance %d fill %d: %.1f cycles per MFMA per SIMD (wave loop %.0f cycles per 48-MFMA stage)\n", waves / 4, PATTERN, FILL,
           st / iters / (48.0 * waves / 4), st / iters);
}

int main() {
    run<1, 0>(); run<2, 0>(); run<4, 0>(); run<8, 0>();
    run<1, 1>(); run<2, 1>(); run<4, 1>(); run<8, 1>();
    run<1, 2>(); run<2, 2>(); run<4, 2>(); run<8, 2>();
    // ONE wave per SIMD (what a wave gets while its SIMD partner is loading or waiting)
    run<1, 0>(256); run<2, 0>(256); run<4, 0>(256); run<8, 0>(256);
    run<1, 1>(256); run<2, 1>(256); run<4, 1>(256); run<8, 1>(256);
    return 0;
}
