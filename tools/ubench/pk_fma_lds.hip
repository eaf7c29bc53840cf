// Micro-repro ATTEMPT for profiles/r3_pk_fma_hazard.md: enc_front_kernel's stem returned wrong, run-to-run different values in lanes
// 48..63 when hipcc's SLP vectoriser turned its scalar FMAs into v_pk_fma_f32 whose sample operand is ONE dword of a register pair
// freshly loaded by ds_read2_b32, selected with op_sel / op_sel_hi, issued right behind the s_waitcnt that covers the load -- only
// with a second wave on the SIMD.  This kernel isolates that instruction pattern:
//   loop:  ds_read2_b32 x[0:1] <- LDS (per-lane sample pair, rewritten every iteration by the wave itself)
//          ds_read_b128 w[0:3] <- LDS (weights)
//          s_waitcnt lgkmcnt(0)            (the loads and the wait are one asm statement, the two packed FMAs the next: hipcc places nothing between)
//          v_pk_fma_f32 acc[0:1], w[0:1], x[0:1], acc[0:1] op_sel_hi:[1,0,1]     (both halves use x[0])
//          v_pk_fma_f32 acc[2:3], w[2:3], x[0:1], acc[2:3] op_sel:[0,1,0] op_sel_hi:[1,1,1]   (both halves use x[1])
//          the same four products as v_fma_f32 into a reference accumulator
//   every `burst` iterations a few v_mfma_f32_16x16x32_f16 run (mode bit 1), as in the fused chain.
// The packed and the scalar accumulators must agree bit for bit; the host counts lanes where they do not, by 16-lane quarter.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o pk_fma_lds_bin pk_fma_lds.hip && ./pk_fma_lds_bin
// Launch variants: 4 waves per CU (one per SIMD) and 8 / 16 waves per CU (two / four per SIMD), with and without MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int V>
__global__ __launch_bounds__(256) void k(unsigned* bad, float* sink, int iters, int mode) {
    __shared__ __attribute__((aligned(16))) float lds[4][64 * 2 + 64 * 4];     // per wave: 64 sample pairs, 64 weight quads
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* sx = &lds[wave][0];
    float* sw = &lds[wave][128];
    for (int i = 0; i < 4; ++i) sw[lane * 4 + i] = 0.25f + 0.001f * (float)((lane * 4 + i) % 97);
    f32x2 pa = {0.f, 0.f}, pb = {0.f, 0.f};              // packed accumulators
    float r0 = 0.f, r1 = 0.f, r2 = 0.f, r3 = 0.f;        // scalar reference
    f32x4 macc = {0.f, 0.f, 0.f, 0.f};
    f16x8 ma, mb;
    for (int e = 0; e < 8; ++e) { ma[e] = (_Float16)(0.01f * lane + e); mb[e] = (_Float16)(0.5f + e); }
    unsigned seed = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    unsigned mism = 0, mism_b = 0;
    for (int it = 0; it < iters; ++it) {
        seed = seed * 1664525u + 1013904223u;
        const float a = (float)(seed >> 8) * (1.0f / 16777216.0f) - 0.5f;
        seed = seed * 1664525u + 1013904223u;
        const float b = (float)(seed >> 8) * (1.0f / 16777216.0f) - 0.5f;
        sx[lane * 2] = a;
        sx[lane * 2 + 1] = b;
        const unsigned xaddr = (unsigned)(size_t)(sx + lane * 2), waddr = (unsigned)(size_t)(sw + lane * 4);
        f32x2 x;
        f32x4 w;
        if (V == 2) {        // the sample pair comes from the VALU, not from a fresh LDS read
            asm volatile("s_waitcnt lgkmcnt(0)\nds_read_b128 %[w], %[wa]\ns_waitcnt lgkmcnt(0)\n" : [w] "=&v"(w) : [wa] "v"(waddr) : "memory");
            x = f32x2{a, b};
            asm volatile("" : "+v"(x));
        } else {
            asm volatile(
                "s_waitcnt lgkmcnt(0)\n"
                "ds_read2_b32 %[x], %[xa] offset1:1\n"
                "ds_read_b128 %[w], %[wa]\n"
                "s_waitcnt lgkmcnt(0)\n"
                : [x] "=&v"(x), [w] "=&v"(w) : [xa] "v"(xaddr), [wa] "v"(waddr) : "memory");
        }
        const f32x2 wl = {w.x, w.y}, wh = {w.z, w.w};
        if (V == 0 || V == 2)
            asm volatile(
                "v_pk_fma_f32 %[pa], %[wl], %[x], %[pa] op_sel_hi:[1,0,1]\n"
                "v_pk_fma_f32 %[pb], %[wh], %[x], %[pb] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n"
                : [pa] "+v"(pa), [pb] "+v"(pb) : [x] "v"(x), [wl] "v"(wl), [wh] "v"(wh));
        if (V == 4)          // the same behind four idle states
            asm volatile(
                "s_nop 4\n"
                "v_pk_fma_f32 %[pa], %[wl], %[x], %[pa] op_sel_hi:[1,0,1]\n"
                "v_pk_fma_f32 %[pb], %[wh], %[x], %[pb] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n"
                : [pa] "+v"(pa), [pb] "+v"(pb) : [x] "v"(x), [wl] "v"(wl), [wh] "v"(wh));
        if (V == 1)          // no operand selection: lanes' halves pair up (x.lo with w.lo / w.z, x.hi with w.y / w.w)
            asm volatile(
                "v_pk_fma_f32 %[pa], %[wl], %[x], %[pa]\n"
                "v_pk_fma_f32 %[pb], %[wh], %[x], %[pb]\n"
                : [pa] "+v"(pa), [pb] "+v"(pb) : [x] "v"(x), [wl] "v"(wl), [wh] "v"(wh));
        if (V == 3) {        // packed multiply + packed add with the same operand selection
            f32x2 ta, tb;
            asm volatile(
                "v_pk_mul_f32 %[ta], %[wl], %[x] op_sel_hi:[1,0]\n"
                "v_pk_mul_f32 %[tb], %[wh], %[x] op_sel:[0,1] op_sel_hi:[1,1]\n"
                "v_pk_add_f32 %[pa], %[ta], %[pa]\n"
                "v_pk_add_f32 %[pb], %[tb], %[pb]\n"
                : [pa] "+v"(pa), [pb] "+v"(pb), [ta] "=&v"(ta), [tb] "=&v"(tb) : [x] "v"(x), [wl] "v"(wl), [wh] "v"(wh));
        }
        // reference: the same products, plain fp32 FMAs on values re-read by the compiler's own loads
        const float xa = sx[lane * 2], xb = sx[lane * 2 + 1];
        const f32x4 wr = *reinterpret_cast<const f32x4*>(sw + lane * 4);
        if (V == 3) {        // unfused reference for the unfused variant
            float t0, t1, t2, t3;
            asm volatile("v_mul_f32 %0, %4, %8\nv_mul_f32 %1, %5, %8\nv_mul_f32 %2, %6, %9\nv_mul_f32 %3, %7, %9\n"
                         : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3) : "v"(wr.x), "v"(wr.y), "v"(wr.z), "v"(wr.w), "v"(xa), "v"(xb));
            asm volatile("v_add_f32 %0, %4, %0\nv_add_f32 %1, %5, %1\nv_add_f32 %2, %6, %2\nv_add_f32 %3, %7, %3\n"
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(t0), "v"(t1), "v"(t2), "v"(t3));
        } else if (V == 1) {
            asm volatile("v_fma_f32 %0, %4, %8, %0\nv_fma_f32 %1, %5, %9, %1\nv_fma_f32 %2, %6, %8, %2\nv_fma_f32 %3, %7, %9, %3\n"
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(wr.x), "v"(wr.y), "v"(wr.z), "v"(wr.w), "v"(xa), "v"(xb));
        } else {
            asm volatile("v_fma_f32 %0, %4, %8, %0\nv_fma_f32 %1, %5, %8, %1\nv_fma_f32 %2, %6, %9, %2\nv_fma_f32 %3, %7, %9, %3\n"     // (asm: hipcc would pack these too)
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(wr.x), "v"(wr.y), "v"(wr.z), "v"(wr.w), "v"(xa), "v"(xb));
        }
        if ((mode & 1) && (it & 7) == 0) {
#pragma unroll
            for (int u = 0; u < 6; ++u) macc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ma, mb, macc, 0, 0, 0);
        }
        // first instruction (src1's LOW half broadcast, op_sel_hi:[1,0,1]) and second (src1's HIGH half broadcast, op_sel:[0,1,0]) apart
        if (__float_as_uint(pa.x) != __float_as_uint(r0) || __float_as_uint(pa.y) != __float_as_uint(r1)) {
            ++mism;
            pa = f32x2{r0, r1};      // resynchronise: count events, not their echo
        }
        if (__float_as_uint(pb.x) != __float_as_uint(r2) || __float_as_uint(pb.y) != __float_as_uint(r3)) {
            ++mism_b;
            pb = f32x2{r2, r3};
        }
    }
    if (mism) atomicAdd(&bad[lane >> 4], mism);
    if (mism_b) atomicAdd(&bad[4 + (lane >> 4)], mism_b);
    sink[blockIdx.x * 256 + threadIdx.x] = pa.x + pa.y + pb.x + pb.y + macc.x;
}

int main() {
    unsigned* bad;
    float* sink;
    CK(hipMalloc(&bad, 32));
    CK(hipMalloc(&sink, 4096 * 256 * 4));
    const int iters = 20000;
    const char* vn[5] = {"LDS pair + op_sel broadcast (the stem's pattern)", "LDS pair, no operand selection", "VALU-made pair + op_sel broadcast",
                         "v_pk_mul + v_pk_add with op_sel (LDS pair)", "the stem's pattern behind s_nop 4"};
    for (int v = 0; v < 5; ++v)
        for (int mode = 0; mode < 2; ++mode)
            for (int per_cu : {1, 2, 4}) {       // 256-thread blocks per CU: 1, 2, 4 waves per SIMD
                if (mode == 0 && per_cu != 2) continue;
                CK(hipMemset(bad, 0, 32));
                auto fn = v == 0 ? k<0> : v == 1 ? k<1> : v == 2 ? k<2> : v == 3 ? k<3> : k<4>;
                hipLaunchKernelGGL(fn, dim3(256 * per_cu), dim3(256), 0, 0, bad, sink, iters, mode);
                CK(hipDeviceSynchronize());
                unsigned h[8];
                CK(hipMemcpy(h, bad, 32, hipMemcpyDeviceToHost));
                printf("variant %d [%s], %s, %d waves per SIMD: mismatching lane-iterations by 16-lane quarter: low-half broadcast %u %u %u %u | high-half broadcast %u %u %u %u of %lld\n", v, vn[v],
                       mode ? "MFMAs in the loop" : "no MFMAs", per_cu, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], (long long)iters * 256 * per_cu * 16);
            }
    return 0;
}
