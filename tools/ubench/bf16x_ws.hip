// Probe: split-operand GEMM (see bf16x_gemm.hip: fp32 operands as three exact bf16 terms, 6 partial products on
// v_mfma_f32_32x32x16_bf16) with WAVE SPECIALISATION.  bf16x_gemm.hip's best lock-step kernel (v3/v4: every wave
// loads, splits, stages, computes; one barrier per 48 MFMAs) stops at ~0.40 of the bf16 peak -- the ceiling the
// programming guide reports for the 128 x 128 / barrier-per-stage structure.  Here a workgroup has 8 waves:
//   waves 4-7 (loaders): global fp32 A rows -> registers (two stages ahead) -> split -> LDS ring of 3 stage buffers;
//   waves 0-3 (matrix) : A fragments from LDS one k-step ahead, B fragments (pre-split, fragment order) straight
//                        from L2 two k-steps ahead, 48 MFMAs per stage, nothing else.
// One s_barrier per stage orders the ring: in iteration i the loaders fill buffer (i+2) % 3 while the matrix waves
// read buffer i % 3 (and may already fetch the first fragments of buffer (i+1) % 3, complete since barrier i-1).
// C[M][N] = A[M][K] * B[N][K]^T.   MT = 32-row tiles per matrix wave (2: 128 x 128 tile, 4: 256 x 128 tile).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BN = 128, KC = 32, P2 = 40, NBUF = 3;

template <int MT>
__global__ __launch_bounds__(512) void gemm_ws(const float* __restrict__ A, const __bf16* __restrict__ Bf, float* __restrict__ C, int M, int N, int K, int mode) {
    constexpr int BM = 64 * MT;                 // 2 matrix-wave rows x MT tiles of 32
    constexpr int PL = BM * P2;                 // bf16 per plane
    constexpr int LS = BM * 8 / 256;            // float4 slots per loader thread and stage
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __bf16* As = reinterpret_cast<__bf16*>(sm); // [NBUF][3][BM * P2]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int S = K / KC;
    if (wave >= 4) {
        // ------------------------------------------------------------------ loaders
        const int lt = tid - 256;
        f32x4 r0[LS], r1[LS];
        auto gload = [&](int s, f32x4 (&r)[LS]) {
#pragma unroll
            for (int i = 0; i < LS; ++i) {
                const int e = lt + 256 * i, row = e >> 3, q = e & 7;
                r[i] = *reinterpret_cast<const f32x4*>(A + (long long)(m0 + row) * K + s * KC + 4 * q);
            }
        };
        auto lstore = [&](int buf, const f32x4 (&r)[LS]) {
#pragma unroll
            for (int i = 0; i < LS; ++i) {
                const int e = lt + 256 * i, row = e >> 3, q = e & 7;
                unsigned h[4], m[4], l[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v = r[i][j];
                    const unsigned bh = __float_as_uint(v) & 0xffff0000u;
                    const float r1_ = v - __uint_as_float(bh);
                    const unsigned bm = __float_as_uint(r1_) & 0xffff0000u;
                    h[j] = bh; m[j] = bm; l[j] = __float_as_uint(r1_ - __uint_as_float(bm));
                }
                __bf16* base = As + buf * 3 * PL + row * P2 + 4 * q;
                unsigned* dh = reinterpret_cast<unsigned*>(base);
                unsigned* dm = reinterpret_cast<unsigned*>(base + PL);
                unsigned* dl = reinterpret_cast<unsigned*>(base + 2 * PL);
                dh[0] = (h[0] >> 16) | h[1]; dh[1] = (h[2] >> 16) | h[3];
                dm[0] = (m[0] >> 16) | m[1]; dm[1] = (m[2] >> 16) | m[3];
                dl[0] = (l[0] >> 16) | (l[1] & 0xffff0000u); dl[1] = (l[2] >> 16) | (l[3] & 0xffff0000u);
            }
        };
        gload(0, r0);
        if (S > 1) gload(1, r1);
        lstore(0, r0);
        if (S > 1) lstore(1, r1);
        if (S > 2) gload(2, r0);
        if (S > 3) gload(3, r1);
        __syncthreads();
        for (int i = 0; i < S; i += 2) {
            if (i + 2 < S) lstore((i + 2) % NBUF, r0);
            if (i + 4 < S) gload(i + 4, r0);
            if (!(mode & 4)) __syncthreads();
            if (i + 1 >= S) break;
            if (i + 3 < S) lstore((i + 3) % NBUF, r1);
            if (i + 5 < S) gload(i + 5, r1);
            if (!(mode & 4)) __syncthreads();
        }
        return;
    }
    // ---------------------------------------------------------------------- matrix waves
    const int wm = wave >> 1, wn = wave & 1;
    const int i32 = lane & 31, kh = lane >> 5;
    const int ksteps = K / 16;
    f32x16 acc[MT][2];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const __bf16* bbase = Bf + ((long long)((n0 + wn * 64) / 32) * ksteps) * (3 * 64 * 8) + lane * 8;
    auto bload = [&](int s_, bf16x8 (&bf)[3][2]) {
        const int sc = s_ < ksteps ? s_ : ksteps - 1;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                bf[pl][b] = *reinterpret_cast<const bf16x8*>(bbase + (((long long)b * ksteps + sc) * 3 + pl) * (64 * 8));
    };
    const int arow = (wm * (32 * MT) + i32) * P2 + 8 * kh;
    auto aread = [&](int buf, int ks, bf16x8 (&af)[3][MT]) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int a = 0; a < MT; ++a) af[pl][a] = *reinterpret_cast<const bf16x8*>(As + (buf * 3 + pl) * PL + arow + a * 32 * P2 + ks * 16);
    };
    auto mma = [&](const bf16x8 (&af)[3][MT], const bf16x8 (&bf)[3][2]) {
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                f32x16 c = acc[a][b];
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][a], bf[2][b], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2][a], bf[0][b], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][a], bf[1][b], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][a], bf[1][b], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][a], bf[0][b], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][a], bf[0][b], c, 0, 0, 0);
                acc[a][b] = c;
            }
    };
    bf16x8 bx[3][2], by[3][2], bz[3][2];
    bf16x8 af0[3][MT], af1[3][MT];
    bload(0, bx);
    bload(1, by);
    bload(2, bz);
    __syncthreads();                            // ring buffers 0 and 1 are filled
    aread(0, 0, af0);
    aread(0, 1, af1);
    int i = 0;
    // one stage = two k-steps; B sets rotate (x, y, z) -> (z, x, y) -> (y, z, x)
    auto stage = [&](bf16x8 (&u0)[3][2], bf16x8 (&u1)[3][2], bf16x8 (&sp)[3][2]) {
        const int buf = i % NBUF;
        if (!(mode & 2)) aread(buf, 1, af1);
        if (!(mode & 1)) bload(2 * i + 2, sp);
        mma(af0, u0);
        if (i + 1 < S && !(mode & 2)) aread((i + 1) % NBUF, 0, af0);   // complete since the previous barrier
        if (!(mode & 1)) bload(2 * i + 3, u0);
        mma(af1, u1);
        if (!(mode & 4)) __syncthreads();
        ++i;
    };
    for (;;) {
        stage(bx, by, bz); if (i >= S) break;
        stage(bz, bx, by); if (i >= S) break;
        stage(by, bz, bx); if (i >= S) break;
    }
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * (32 * MT) + a * 32 + 8 * (r / 4) + kh * 4 + (r % 4);
                const int col = n0 + wn * 64 + b * 32 + i32;
                C[(long long)row * N + col] = acc[a][b][r];
            }
}

template <int MT>
static void run(int mode, const float* A, const __bf16* Bf, float* C, int M, int N, int K, std::vector<float>& hC, const std::vector<float>& hA,
                const std::vector<float>& hB) {
    constexpr int BM = 64 * MT;
    const int lds = NBUF * 3 * BM * P2 * 2;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_ws<MT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const dim3 grid(N / BN, M / BM);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(gemm_ws<MT>, grid, dim3(512), lds, 0, A, Bf, C, M, N, K, mode);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    CK(hipGetLastError());
    printf("mode %d: wave-specialised, tile %d x %d, K = %d: %.3f ms per GEMM -> %.1f fp32-equivalent TFLOP/s (%.0f TFLOP/s of bf16 MFMA work)", mode, BM, BN, K, ms / 10,
           2.0 * M * N * K / (ms / 10 * 1e-3) / 1e12, 12.0 * M * N * K / (ms / 10 * 1e-3) / 1e12);
    CK(hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost));
    double sumsq = 0, refsq = 0, maxrel = 0;
    for (int t = 0; t < 4000; ++t) {
        const int i = (t * 7919) % M, j = (t * 104729) % N;
        double ref = 0;
        for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)i * K + k] * hB[(size_t)j * K + k];
        const double d = hC[(size_t)i * N + j] - ref;
        sumsq += d * d; refsq += ref * ref;
        maxrel = fmax(maxrel, fabs(d) / (fabs(ref) + 1e-3));
    }
    printf(";  rms error / rms value %.3e, max rel %.2e\n", sqrt(sumsq / refsq), maxrel);
}

int main() {
    const int M = 8192, N = 4096;
    for (int K : {1024, 4096}) {
        std::vector<float> hA((size_t)M * K), hB((size_t)N * K), hC((size_t)M * N);
        unsigned s = 12345u;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
        for (auto& v : hA) v = rnd() * 1.3f;
        for (auto& v : hB) v = rnd() * 0.7f;
        float *A, *C;
        CK(hipMalloc(&A, hA.size() * 4)); CK(hipMalloc(&C, hC.size() * 4));
        CK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
        std::vector<unsigned short> hBf((size_t)3 * N * K);
        for (int nt = 0; nt < N / 32; ++nt)
            for (int s_ = 0; s_ < K / 16; ++s_)
                for (int l = 0; l < 64; ++l)
                    for (int e = 0; e < 8; ++e) {
                        const float v = hB[(size_t)(nt * 32 + (l & 31)) * K + s_ * 16 + 8 * (l >> 5) + e];
                        unsigned b; memcpy(&b, &v, 4);
                        const unsigned bh = b & 0xffff0000u; float fh; memcpy(&fh, &bh, 4);
                        const float r1 = v - fh; unsigned b1; memcpy(&b1, &r1, 4);
                        const unsigned bm = b1 & 0xffff0000u; float fm; memcpy(&fm, &bm, 4);
                        const float r2 = r1 - fm; unsigned b2; memcpy(&b2, &r2, 4);
                        const size_t base = (((size_t)nt * (K / 16) + s_) * 3) * 512 + (size_t)l * 8 + e;
                        hBf[base] = bh >> 16; hBf[base + 512] = bm >> 16; hBf[base + 1024] = b2 >> 16;
                    }
        __bf16* Bf;
        CK(hipMalloc(&Bf, hBf.size() * 2)); CK(hipMemcpy(Bf, hBf.data(), hBf.size() * 2, hipMemcpyHostToDevice));
        for (int mode : {0, 1, 2, 4, 3, 7}) run<2>(mode, A, Bf, C, M, N, K, hC, hA, hB);   // timing modes: 1 no B loads, 2 no A reads, 4 no barriers (wrong results)
        // run<4>: 256 x 128 tile needs a 2-buffer ring (LDS) and a leaner fragment schedule (VGPRs); not built yet
        CK(hipFree(A)); CK(hipFree(C)); CK(hipFree(Bf));
    }
    return 0;
}
