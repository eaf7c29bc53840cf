// Bare issue rate of v_mfma_f32_32x32x16_bf16 / v_mfma_f32_16x16x32_bf16 on MI355X (what is the real roofline of the
// split-operand GEMM?).  256 x OCC workgroups of 4 waves; NACC independent accumulators per wave; dependent chains.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// DATA 0: smooth operands (few significand bits set); DATA 1: random bf16 in [-1, 1) with random significands -- the
// matrix pipe's clock (power management) depends on the operand bits, so the second figure is the practical ceiling
template <int NACC, int SHAPE, int DATA = 0>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* clk) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(1.0f + i * 0.01f); }
    if (DATA == 1) {
        unsigned s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
        for (int i = 0; i < 8; ++i) {
            s = s * 1664525u + 1013904223u; a[i] = (__bf16)(((s >> 8) & 0xffff) / 32768.0f - 1.0f);
            s = s * 1664525u + 1013904223u; b[i] = (__bf16)(((s >> 8) & 0xffff) / 32768.0f - 1.0f);
        }
    }
    f32x16 acc[NACC];
    f32x4 acc4[NACC];
    for (int n = 0; n < NACC; ++n) { for (int r = 0; r < 16; ++r) acc[n][r] = 0.f; for (int r = 0; r < 4; ++r) acc4[n][r] = 0.f; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep)
#pragma unroll
            for (int n = 0; n < NACC; ++n) {
                if (SHAPE == 32) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[n], 0, 0, 0);
                else acc4[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc4[n], 0, 0, 0);
            }
    }
    float s = 0.f;
    for (int n = 0; n < NACC; ++n) s += acc[n][0] + acc4[n][0];
    if (s == 12345.f) out[0] = s;
    if (clk && threadIdx.x == 0) {   // shader-clock ticks and 100 MHz real-time ticks this workgroup lived
        clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
        clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

template <int NACC, int SHAPE, int DATA = 0>
void run(int occ) {
    float* out; CK(hipMalloc(&out, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000;
    unsigned long long* clk; CK(hipMalloc(&clk, 256 * occ * 16)); CK(hipMemset(clk, 0, 256 * occ * 16));
    hipLaunchKernelGGL((k<NACC, SHAPE, DATA>), dim3(256 * occ), dim3(256), 0, 0, out, 10, nullptr);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<NACC, SHAPE, DATA>), dim3(256 * occ), dim3(256), 0, 0, out, iters, clk);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long hc[512 * 2]; CK(hipMemcpy(hc, clk, 256 * occ * 16, hipMemcpyDeviceToHost));
    double st = 0, rt = 0; for (int i = 0; i < 256 * occ; ++i) { st += hc[2 * i]; rt += hc[2 * i + 1]; }
    printf("[shader clock %.0f MHz] ", 100.0 * st / rt);
    const double flops = (double)256 * occ * 4 * iters * 8 * NACC * (SHAPE == 32 ? 32768.0 : 16384.0);
    printf("%s, %s operands, %d accumulators, %d workgroup(s)/CU: %.1f TFLOP/s bf16\n", SHAPE == 32 ? "32x32x16" : "16x16x32", DATA ? "random" : "smooth", NACC, occ, flops / (ms * 1e-3) / 1e12);
}

int main() {
    run<1, 32>(1); run<2, 32>(1); run<4, 32>(1); run<4, 32>(2); run<1, 32>(2);
    run<1, 16>(1); run<4, 16>(1); run<4, 16>(2);
    run<4, 32, 1>(1); run<4, 32, 1>(2); run<4, 16, 1>(1); run<4, 16, 1>(2);
    return 0;
}
