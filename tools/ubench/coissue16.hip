// coissue.hip for the fp16 matrix pipe: how fast does a VALU wave run next to a wave that streams v_mfma_f32_32x32x16_f16 on the same
// SIMD, and what does the MFMA wave lose?  Block = 8 waves: waves 0-3 MFMA (one per SIMD), waves 4-7 VALU.
//   valu_kind 0: one dependent fma chain   1: four independent fma chains   2: epilogue-like (fma + class/cndmask/max + buffer store)
//   hipcc --offload-arch=gfx950 -O3 -o coissue16_bin coissue16.hip && ./coissue16_bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
template <int valu_kind>
__global__ __launch_bounds__(512) void k(float* out, float* sink, unsigned long long* cyc, int iters, int mode) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float res = 0.f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) {
        if (mode & 1) {
            f32x16 acc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
            f16x8 a, b;
#pragma unroll
            for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.01f + e); b[e] = (_Float16)(0.5f + e); }
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 32; ++u)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);   // 128 MFMAs per iteration
                a[0] += (_Float16)1.f;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) res += acc[i][0] + acc[i][15];
        }
    } else {
        if (mode & 2) {
            float x0 = threadIdx.x * 0.001f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
            unsigned am = 0;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(sink + (size_t)blockIdx.x * 65536), 0, 65536 * 4, 0x00020000);
            for (int it = 0; it < iters; ++it) {
                if (valu_kind == 0) {
#pragma unroll
                    for (int u = 0; u < 128; ++u) x0 = fmaf(x0, 1.0001f, 0.5f);
                } else if (valu_kind == 1) {
#pragma unroll
                    for (int u = 0; u < 32; ++u) { x0 = fmaf(x0, 1.0001f, 0.5f); x1 = fmaf(x1, 1.0001f, 0.5f); x2 = fmaf(x2, 1.0001f, 0.5f); x3 = fmaf(x3, 1.0001f, 0.5f); }
                } else if (valu_kind >= 3) {   // 4 independent chains, exact instruction stream: 3: bare, 4: s_nop 0, 5: s_nop 1, 6: s_nop 3 after every fma, 7: s_nop 7
#define FMA4(NOP) asm volatile("v_fma_f32 %0, %0, %4, %5\n" NOP "v_fma_f32 %1, %1, %4, %5\n" NOP "v_fma_f32 %2, %2, %4, %5\n" NOP "v_fma_f32 %3, %3, %4, %5\n" NOP \
                               : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(1.0001f), "v"(0.5f))
#pragma unroll
                    for (int u = 0; u < 32; ++u) {
                        if (valu_kind == 3) FMA4("");
                        if (valu_kind == 4) FMA4("s_nop 0\n");
                        if (valu_kind == 5) FMA4("s_nop 1\n");
                        if (valu_kind == 6) FMA4("s_nop 3\n");
                        if (valu_kind == 7) FMA4("s_nop 7\n");
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < 32; ++u) {     // 4 VALU + 1 store per value, 32 values: 128 VALU + 32 stores per iteration
                        x0 = fmaf(x0, 1.0001f, 0.5f);
                        const unsigned bits = __builtin_amdgcn_classf(x0, 0x1F8) ? (__float_as_uint(x0) & 0x7fffffffu) : 0u;
                        am = bits > am ? bits : am;
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(x0), rs, (int)(threadIdx.x & 255) * 4, (u * 256 + (it & 7) * 8192) * 4, 0);
                    }
                }
            }
            res = x0 + x1 + x2 + x3 + __uint_as_float(am);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 512 + threadIdx.x] = res;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}
int main() {
    float *out, *sink; unsigned long long* cyc;
    CK(hipMalloc(&out, 1024 * 512 * 4)); CK(hipMalloc(&sink, (size_t)256 * 65536 * 4)); CK(hipMalloc(&cyc, 64));
    const int iters = 300;
    for (int kind = 0; kind < 8; ++kind)
        for (int mode : {1, 2, 3}) {
            CK(hipMemset(cyc, 0, 64));
#define L(K) if (kind == K) hipLaunchKernelGGL(k<K>, dim3(256), dim3(512), 0, 0, out, sink, cyc, iters, mode)
            L(0); L(1); L(2); L(3); L(4); L(5); L(6); L(7);
            CK(hipDeviceSynchronize());
            unsigned long long h[8]; CK(hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost));
            printf("valu_kind=%d mode=%d | mfma wave cycles per MFMA %6.1f | valu wave cycles per VALU instruction %6.1f\n", kind, mode,
                   (double)h[0] / iters / 128, (double)h[4] / iters / 128);
        }
    return 0;
}
