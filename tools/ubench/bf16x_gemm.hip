// Feasibility probe: fp32-fidelity GEMM on the bf16 matrix pipe ("bf16 x N" split-operand emulation) on MI355X.
// Every fp32 operand is split EXACTLY into three bf16 terms (a = hi + mid + lo, 8 significand bits each, by
// truncation), the product is sum_{i,j} a_i * b_j accumulated in fp32 by v_mfma_f32_32x32x16_bf16:
//   TERMS = 9: all nine partial products (only fp32-accumulation rounding remains);
//   TERMS = 6: drop mid*lo, lo*mid, lo*lo (relative error of a product <= ~2^-22).
// The bf16 pipe runs 16x the fp32 MFMA rate, so 9 (6) terms cost 0.56x (0.375x) of the fp32-MFMA time at equal
// efficiency.  This program measures what a straightforward LDS-tiled kernel reaches, operands split on the fly
// after the LDS read, and the error against an fp64 reference.  C[M][N] = A[M][K] * B[N][K]^T, all fp32 in HBM.
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
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, KC = 32, KCP = KC + 4;

// 8 fp32 -> three packs of 8 bf16 (truncation split: exact, a = hi + mid + lo)
__device__ __forceinline__ void split8(const f32x4 x0, const f32x4 x1, bf16x8& hi, bf16x8& mid, bf16x8& lo) {
    unsigned h[8], m[8], l[8];
    const float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const unsigned b = __float_as_uint(v[i]);
        const unsigned bh = b & 0xffff0000u;
        const float r1 = v[i] - __uint_as_float(bh);
        const unsigned bm = __float_as_uint(r1) & 0xffff0000u;
        const float r2 = r1 - __uint_as_float(bm);
        h[i] = bh; m[i] = bm; l[i] = __float_as_uint(r2);
    }
    u32x4 ph, pm, pl;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ph[i] = (h[2 * i] >> 16) | (h[2 * i + 1] & 0xffff0000u);
        pm[i] = (m[2 * i] >> 16) | (m[2 * i + 1] & 0xffff0000u);
        pl[i] = (l[2 * i] >> 16) | (l[2 * i + 1] & 0xffff0000u);
    }
    hi = __builtin_bit_cast(bf16x8, ph); mid = __builtin_bit_cast(bf16x8, pm); lo = __builtin_bit_cast(bf16x8, pl);
}

template <int TERMS>
__global__ __launch_bounds__(256) void gemm(const float* A, const float* B, float* C, int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) float As[2][BM * KCP], Bs[2][BN * KCP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;                 // 2 x 2 waves, each 64 x 64 = 2 x 2 MFMA tiles of 32 x 32
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int i32 = lane & 31, kh = lane >> 5;               // MFMA operand: row i32, k = 8*kh .. 8*kh+7
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    auto stage = [&](int buf, int k0) {
        for (int e = tid; e < BM * (KC / 4); e += 256) {
            const int row = e / (KC / 4), q = e % (KC / 4);
            *reinterpret_cast<f32x4*>(&As[buf][row * KCP + 4 * q]) = *reinterpret_cast<const f32x4*>(A + (long long)(m0 + row) * K + k0 + 4 * q);
            *reinterpret_cast<f32x4*>(&Bs[buf][row * KCP + 4 * q]) = *reinterpret_cast<const f32x4*>(B + (long long)(n0 + row) * K + k0 + 4 * q);
        }
    };
    stage(0, 0);
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < K; k0 += KC) {
        if (k0 + KC < K) stage(buf ^ 1, k0 + KC);
#pragma unroll
        for (int ks = 0; ks < KC / 16; ++ks) {
            bf16x8 ah[2], am[2], al[2], bh[2], bm[2], bl[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const float* p = &As[buf][(wm * 64 + a * 32 + i32) * KCP + ks * 16 + 8 * kh];
                split8(*reinterpret_cast<const f32x4*>(p), *reinterpret_cast<const f32x4*>(p + 4), ah[a], am[a], al[a]);
            }
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const float* p = &Bs[buf][(wn * 64 + b * 32 + i32) * KCP + ks * 16 + 8 * kh];
                split8(*reinterpret_cast<const f32x4*>(p), *reinterpret_cast<const f32x4*>(p + 4), bh[b], bm[b], bl[b]);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    f32x16 c = acc[a][b];
                    // small terms first
                    if (TERMS == 9) {
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[a], bl[b], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[a], bl[b], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[a], bm[b], c, 0, 0, 0);
                    }
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bl[b], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[a], bh[b], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[a], bm[b], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bm[b], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[a], bh[b], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bh[b], c, 0, 0, 0);
                    acc[a][b] = c;
                }
        }
        __syncthreads();
        buf ^= 1;
    }
    // C layout of 32x32: column = lane & 31, rows 8*(r/4) + (lane>>5)*4 + r%4
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + a * 32 + 8 * (r / 4) + kh * 4 + (r % 4);
                const int col = n0 + wn * 64 + b * 32 + i32;
                C[(long long)row * N + col] = acc[a][b][r];
            }
}

// v2: operands split ONCE -- B offline into three bf16 planes Bp[3][N][K], A while it is staged into LDS -- so
// the k-loop carries only LDS fragment reads and MFMAs; global loads for the next stage are in flight during the
// MFMA phase (register prefetch), single LDS buffer, two barriers per stage.
constexpr int P2 = 40;      // LDS row pitch in bf16 (32 + 8)
template <int TERMS>
__global__ __launch_bounds__(256) void gemm2(const float* A, const __bf16* Bp, float* C, int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) __bf16 As[3][BM * P2], Bs[3][BN * P2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int i32 = lane & 31, kh = lane >> 5;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    // staging slots: A: 128 rows x 8 float4 = 1024 -> 4 per thread; B: 3 planes x 128 rows x 4 (8 bf16) = 1536 -> 6 per thread
    f32x4 ra[4];
    u32x4 rb[6];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i, row = e >> 3, q = e & 7;
            ra[i] = *reinterpret_cast<const f32x4*>(A + (long long)(m0 + row) * K + k0 + 4 * q);
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int e = tid + 256 * i, pl = e >> 9, row = (e >> 2) & 127, q = e & 3;
            rb[i] = *reinterpret_cast<const u32x4*>(Bp + ((long long)pl * N + n0 + row) * K + k0 + 8 * q);
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i, row = e >> 3, q = e & 7;
            unsigned h[4], m[4], l[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = ra[i][j];
                const unsigned bh = __float_as_uint(v) & 0xffff0000u;
                const float r1 = v - __uint_as_float(bh);
                const unsigned bm = __float_as_uint(r1) & 0xffff0000u;
                h[j] = bh; m[j] = bm; l[j] = __float_as_uint(r1 - __uint_as_float(bm));
            }
            unsigned* dh = reinterpret_cast<unsigned*>(&As[0][row * P2 + 4 * q]);
            unsigned* dm = reinterpret_cast<unsigned*>(&As[1][row * P2 + 4 * q]);
            unsigned* dl = reinterpret_cast<unsigned*>(&As[2][row * P2 + 4 * q]);
            dh[0] = (h[0] >> 16) | h[1]; dh[1] = (h[2] >> 16) | h[3];
            dm[0] = (m[0] >> 16) | m[1]; dm[1] = (m[2] >> 16) | m[3];
            dl[0] = (l[0] >> 16) | (l[1] & 0xffff0000u); dl[1] = (l[2] >> 16) | (l[3] & 0xffff0000u);
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int e = tid + 256 * i, pl = e >> 9, row = (e >> 2) & 127, q = e & 3;
            *reinterpret_cast<u32x4*>(&Bs[pl][row * P2 + 8 * q]) = rb[i];
        }
    };
    gload(0);
    lstore();
    __syncthreads();
    for (int k0 = 0; k0 < K; k0 += KC) {
        const bool more = k0 + KC < K;
        if (more) gload(k0 + KC);
#pragma unroll
        for (int ks = 0; ks < KC / 16; ++ks) {
            bf16x8 af[3][2], bf[3][2];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                for (int a = 0; a < 2; ++a) af[pl][a] = *reinterpret_cast<const bf16x8*>(&As[pl][(wm * 64 + a * 32 + i32) * P2 + ks * 16 + 8 * kh]);
#pragma unroll
                for (int b = 0; b < 2; ++b) bf[pl][b] = *reinterpret_cast<const bf16x8*>(&Bs[pl][(wn * 64 + b * 32 + i32) * P2 + ks * 16 + 8 * kh]);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    f32x16 c = acc[a][b];
                    if (TERMS == 9) {
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2][a], bf[2][b], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][a], bf[2][b], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2][a], bf[1][b], c, 0, 0, 0);
                    }
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][a], bf[2][b], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2][a], bf[0][b], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][a], bf[1][b], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][a], bf[1][b], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][a], bf[0][b], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][a], bf[0][b], c, 0, 0, 0);
                    acc[a][b] = c;
                }
        }
        __syncthreads();
        if (more) {
            lstore();
            __syncthreads();
        }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + a * 32 + 8 * (r / 4) + kh * 4 + (r % 4);
                const int col = n0 + wn * 64 + b * 32 + i32;
                C[(long long)row * N + col] = acc[a][b][r];
            }
}

// v3: the weight operand never touches LDS: B is pre-split AND pre-packed in MFMA B-fragment order
//   Bf[n-tile of 32][k-step of 16][plane 3][lane 64][8 bf16]   (a wave-load = 1 KB contiguous per plane)
// and loaded straight from L2 into registers one k-step ahead; A is split while staged into a double-buffered LDS
// slab (3 planes).  LDS now serves only the A fragments (the probe v2 was LDS-bandwidth-bound: 3 planes of A and B
// fragments + staging ~ 96 of the 128 B/clk).
__device__ unsigned long long* g_trace;   // [wave 4][kstep 64][4 stamps]
#define TRC(s_, k_) do { if (trace_on && lane == 0 && (s_) < 64) g_trace[((wave * 64) + (s_)) * 4 + (k_)] = __builtin_amdgcn_s_memtime(); } while (0)
template <int TERMS>
__global__ __launch_bounds__(256) void gemm3(const float* A, const __bf16* Bf, float* C, int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) __bf16 As[2][3][BM * P2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int i32 = lane & 31, kh = lane >> 5;
    const int ksteps = K / 16;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    f32x4 ra[4];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i, row = e >> 3, q = e & 7;
            ra[i] = *reinterpret_cast<const f32x4*>(A + (long long)(m0 + row) * K + k0 + 4 * q);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i, row = e >> 3, q = e & 7;
            unsigned h[4], m[4], l[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = ra[i][j];
                const unsigned bh = __float_as_uint(v) & 0xffff0000u;
                const float r1 = v - __uint_as_float(bh);
                const unsigned bm = __float_as_uint(r1) & 0xffff0000u;
                h[j] = bh; m[j] = bm; l[j] = __float_as_uint(r1 - __uint_as_float(bm));
            }
            unsigned* dh = reinterpret_cast<unsigned*>(&As[buf][0][row * P2 + 4 * q]);
            unsigned* dm = reinterpret_cast<unsigned*>(&As[buf][1][row * P2 + 4 * q]);
            unsigned* dl = reinterpret_cast<unsigned*>(&As[buf][2][row * P2 + 4 * q]);
            dh[0] = (h[0] >> 16) | h[1]; dh[1] = (h[2] >> 16) | h[3];
            dm[0] = (m[0] >> 16) | m[1]; dm[1] = (m[2] >> 16) | m[3];
            dl[0] = (l[0] >> 16) | (l[1] & 0xffff0000u); dl[1] = (l[2] >> 16) | (l[3] & 0xffff0000u);
        }
    };
    // B fragments of this wave's two n-tiles for k-step s: [b][plane]
    const __bf16* bbase = Bf + ((long long)((n0 + wn * 64) / 32) * ksteps) * (3 * 64 * 8) + lane * 8;
    auto bload = [&](int s_, bf16x8 (&bf)[3][2]) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                bf[pl][b] = *reinterpret_cast<const bf16x8*>(bbase + (((long long)b * ksteps + s_) * 3 + pl) * (64 * 8));
    };
    const bool trace_on = g_trace && blockIdx.x == 3 && blockIdx.y == 17;
    gload(0);
    lstore(0);
    bf16x8 bcur[3][2], bnxt[3][2];
    bload(0, bcur);
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < K; k0 += KC) {
        const bool more = k0 + KC < K;
        if (more) gload(k0 + KC);
#pragma unroll
        for (int ks = 0; ks < KC / 16; ++ks) {
            const int s_ = k0 / 16 + ks;
            TRC(s_, 0);
            if (s_ + 1 < ksteps) bload(s_ + 1, bnxt);
            bf16x8 af[3][2];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int a = 0; a < 2; ++a) af[pl][a] = *reinterpret_cast<const bf16x8*>(&As[buf][pl][(wm * 64 + a * 32 + i32) * P2 + ks * 16 + 8 * kh]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            TRC(s_, 1);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    f32x16 c = acc[a][b];
                    if (TERMS == 9) {
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2][a], bcur[2][b], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][a], bcur[2][b], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2][a], bcur[1][b], c, 0, 0, 0);
                    }
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][a], bcur[2][b], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2][a], bcur[0][b], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][a], bcur[1][b], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][a], bcur[1][b], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][a], bcur[0][b], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][a], bcur[0][b], c, 0, 0, 0);
                    acc[a][b] = c;
                }
            TRC(s_, 2);
            if (ks == 0 && more) lstore(buf ^ 1);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int b = 0; b < 2; ++b) bcur[pl][b] = bnxt[pl][b];
            TRC(s_, 3);
        }
        __syncthreads();
        buf ^= 1;
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + a * 32 + 8 * (r / 4) + kh * 4 + (r % 4);
                const int col = n0 + wn * 64 + b * 32 + i32;
                C[(long long)row * N + col] = acc[a][b][r];
            }
}

// v6: v3 with 8 waves and a 128 x 256 tile: the A slab is split ONCE per 256 output columns (half the split / staging
// VALU work per MFMA: VALU and MFMA issue of co-resident waves serialise on this chip), one workgroup per CU.
template <int TERMS>
__global__ __launch_bounds__(512) void gemm6(const float* A, const __bf16* Bf, float* C, int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) __bf16 As[2][3][BM * P2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;               // 2 x 4 waves of 64 x 64: 128 x 256 tile
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * 256;
    const int i32 = lane & 31, kh = lane >> 5;
    const int ksteps = K / 16;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    f32x4 ra[2];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = tid + 512 * i, row = e >> 3, q = e & 7;
            ra[i] = *reinterpret_cast<const f32x4*>(A + (long long)(m0 + row) * K + k0 + 4 * q);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = tid + 512 * i, row = e >> 3, q = e & 7;
            unsigned h[4], m[4], l[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = ra[i][j];
                const unsigned bh = __float_as_uint(v) & 0xffff0000u;
                const float r1 = v - __uint_as_float(bh);
                const unsigned bm = __float_as_uint(r1) & 0xffff0000u;
                h[j] = bh; m[j] = bm; l[j] = __float_as_uint(r1 - __uint_as_float(bm));
            }
            unsigned* dh = reinterpret_cast<unsigned*>(&As[buf][0][row * P2 + 4 * q]);
            unsigned* dm = reinterpret_cast<unsigned*>(&As[buf][1][row * P2 + 4 * q]);
            unsigned* dl = reinterpret_cast<unsigned*>(&As[buf][2][row * P2 + 4 * q]);
            dh[0] = (h[0] >> 16) | h[1]; dh[1] = (h[2] >> 16) | h[3];
            dm[0] = (m[0] >> 16) | m[1]; dm[1] = (m[2] >> 16) | m[3];
            dl[0] = (l[0] >> 16) | (l[1] & 0xffff0000u); dl[1] = (l[2] >> 16) | (l[3] & 0xffff0000u);
        }
    };
    // B fragments of this wave's two n-tiles for k-step s: [b][plane]
    const __bf16* bbase = Bf + ((long long)((n0 + wn * 64) / 32) * ksteps) * (3 * 64 * 8) + lane * 8;
    auto bload = [&](int s_, bf16x8 (&bf)[3][2]) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                bf[pl][b] = *reinterpret_cast<const bf16x8*>(bbase + (((long long)b * ksteps + s_) * 3 + pl) * (64 * 8));
    };
    const bool trace_on = false;
    gload(0);
    lstore(0);
    bf16x8 bcur[3][2], bnxt[3][2];
    bload(0, bcur);
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < K; k0 += KC) {
        const bool more = k0 + KC < K;
        if (more) gload(k0 + KC);
#pragma unroll
        for (int ks = 0; ks < KC / 16; ++ks) {
            const int s_ = k0 / 16 + ks;
            TRC(s_, 0);
            if (s_ + 1 < ksteps) bload(s_ + 1, bnxt);
            bf16x8 af[3][2];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int a = 0; a < 2; ++a) af[pl][a] = *reinterpret_cast<const bf16x8*>(&As[buf][pl][(wm * 64 + a * 32 + i32) * P2 + ks * 16 + 8 * kh]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            TRC(s_, 1);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    f32x16 c = acc[a][b];
                    if (TERMS == 9) {
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2][a], bcur[2][b], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][a], bcur[2][b], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2][a], bcur[1][b], c, 0, 0, 0);
                    }
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][a], bcur[2][b], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2][a], bcur[0][b], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][a], bcur[1][b], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][a], bcur[1][b], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][a], bcur[0][b], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][a], bcur[0][b], c, 0, 0, 0);
                    acc[a][b] = c;
                }
            TRC(s_, 2);
            if (ks == 0 && more) lstore(buf ^ 1);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int b = 0; b < 2; ++b) bcur[pl][b] = bnxt[pl][b];
            TRC(s_, 3);
        }
        __syncthreads();
        buf ^= 1;
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + a * 32 + 8 * (r / 4) + kh * 4 + (r % 4);
                const int col = n0 + wn * 64 + b * 32 + i32;
                C[(long long)row * N + col] = acc[a][b][r];
            }
}

// v4: like v3 (weights pre-split, fragment order, straight from L2) but the A slab stays fp32 in LDS (4 B/element
// instead of 6: KC = 64 fits double-buffered, one barrier per 96 MFMAs) and is split after the fragment read --
// the split VALU work then sits between the MFMAs of the same basic block instead of in front of a barrier.
constexpr int KC4 = 64, KCP4 = KC4 + 4;
template <int TERMS>
__global__ __launch_bounds__(256, 2) void gemm4(const float* A, const __bf16* Bf, float* C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) float sm4[];
    float* As = sm4;                                          // [2][BM][KCP4]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int i32 = lane & 31, kh = lane >> 5;
    const int ksteps = K / 16;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    f32x4 ra[8];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = tid + 256 * i, row = e >> 4, q = e & 15;
            ra[i] = *reinterpret_cast<const f32x4*>(A + (long long)(m0 + row) * K + k0 + 4 * q);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = tid + 256 * i, row = e >> 4, q = e & 15;
            *reinterpret_cast<f32x4*>(&As[(buf * BM + row) * KCP4 + 4 * q]) = ra[i];
        }
    };
    const __bf16* bbase = Bf + ((long long)((n0 + wn * 64) / 32) * ksteps) * (3 * 64 * 8) + lane * 8;
    auto bload = [&](int s_, bf16x8 (&bf)[3][2]) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                bf[pl][b] = *reinterpret_cast<const bf16x8*>(bbase + (((long long)b * ksteps + s_) * 3 + pl) * (64 * 8));
    };
    gload(0);
    lstore(0);
    bf16x8 bcur[3][2], bnxt[3][2];
    bload(0, bcur);
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < K; k0 += KC4) {
        const bool more = k0 + KC4 < K;
        if (more) gload(k0 + KC4);
#pragma unroll
        for (int ks = 0; ks < KC4 / 16; ++ks) {
            const int s_ = k0 / 16 + ks;
            if (s_ + 1 < ksteps) bload(s_ + 1, bnxt);
            bf16x8 ah[2], am[2], al[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const float* p = &As[(buf * BM + wm * 64 + a * 32 + i32) * KCP4 + ks * 16 + 8 * kh];
                split8(*reinterpret_cast<const f32x4*>(p), *reinterpret_cast<const f32x4*>(p + 4), ah[a], am[a], al[a]);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    f32x16 c = acc[a][b];
                    if (TERMS == 9) {
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[a], bcur[2][b], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[a], bcur[2][b], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[a], bcur[1][b], c, 0, 0, 0);
                    }
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bcur[2][b], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[a], bcur[0][b], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[a], bcur[1][b], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bcur[1][b], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[a], bcur[0][b], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bcur[0][b], c, 0, 0, 0);
                    acc[a][b] = c;
                }
            if (ks == 1 && more) lstore(buf ^ 1);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int b = 0; b < 2; ++b) bcur[pl][b] = bnxt[pl][b];
        }
        __syncthreads();
        buf ^= 1;
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + a * 32 + 8 * (r / 4) + kh * 4 + (r % 4);
                const int col = n0 + wn * 64 + b * 32 + i32;
                C[(long long)row * N + col] = acc[a][b][r];
            }
}

// v5: one workgroup per CU with a 256 x 128 tile: 2 x 2 waves of 128 x 64 (4 x 2 MFMA tiles, 128 accumulator
// registers); A split while staged into a double-buffered slab of 3 planes (123 KB of LDS), B straight from L2 two
// k-steps ahead; per k-step and wave 48 MFMAs for 12 fragment reads and 6 B loads, one barrier per 96 MFMAs.
constexpr int BM5 = 256;
template <int TERMS>
__global__ __launch_bounds__(256) void gemm5(const float* A, const __bf16* Bf, float* C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) float sm5[];
    __bf16* As = reinterpret_cast<__bf16*>(sm5);             // [2][3][BM5 * P2]
    constexpr int PL = BM5 * P2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM5, n0 = blockIdx.x * BN;
    const int i32 = lane & 31, kh = lane >> 5;
    const int ksteps = K / 16;
    f32x16 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    f32x4 ra[8];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = tid + 256 * i, row = e >> 3, q = e & 7;
            ra[i] = *reinterpret_cast<const f32x4*>(A + (long long)(m0 + row) * K + k0 + 4 * q);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int e = tid + 256 * i, row = e >> 3, q = e & 7;
            unsigned h[4], m[4], l[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = ra[i][j];
                const unsigned bh = __float_as_uint(v) & 0xffff0000u;
                const float r1 = v - __uint_as_float(bh);
                const unsigned bm = __float_as_uint(r1) & 0xffff0000u;
                h[j] = bh; m[j] = bm; l[j] = __float_as_uint(r1 - __uint_as_float(bm));
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
    const __bf16* bbase = Bf + ((long long)((n0 + wn * 64) / 32) * ksteps) * (3 * 64 * 8) + lane * 8;
    auto bload = [&](int s_, bf16x8 (&bf)[3][2]) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                bf[pl][b] = *reinterpret_cast<const bf16x8*>(bbase + (((long long)b * ksteps + (s_ < ksteps ? s_ : 0)) * 3 + pl) * (64 * 8));
    };
    auto kstep = [&](int buf, int ks, const bf16x8 (&bf)[3][2]) {
        bf16x8 af[3][4];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int a = 0; a < 4; ++a) af[pl][a] = *reinterpret_cast<const bf16x8*>(As + (buf * 3 + pl) * PL + (wm * 128 + a * 32 + i32) * P2 + ks * 16 + 8 * kh);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                f32x16 c = acc[a][b];
                if (TERMS == 9) {
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2][a], bf[2][b], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][a], bf[2][b], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2][a], bf[1][b], c, 0, 0, 0);
                }
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][a], bf[2][b], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2][a], bf[0][b], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][a], bf[1][b], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][a], bf[1][b], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][a], bf[0][b], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][a], bf[0][b], c, 0, 0, 0);
                acc[a][b] = c;
            }
    };
    gload(0);
    lstore(0);
    bf16x8 bx[3][2], by[3][2], bz[3][2];
    bload(0, bx);
    bload(1, by);
    __syncthreads();
    int buf = 0;
    // stages of 2 k-steps; B sets rotate (x,y,z) -> (z,x,y) -> (y,z,x)
    auto stage = [&](int k0, bf16x8 (&u0)[3][2], bf16x8 (&u1)[3][2], bf16x8 (&sp)[3][2]) {
        const bool more = k0 + KC < K;
        const int s_ = k0 / 16;
        if (more) gload(k0 + KC);
        bload(s_ + 2, sp);
        kstep(buf, 0, u0);
        bload(s_ + 3, u0);
        kstep(buf, 1, u1);
        if (more) lstore(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    };
    for (int k0 = 0; k0 < K;) {
        stage(k0, bx, by, bz); k0 += KC; if (k0 >= K) break;
        stage(k0, bz, bx, by); k0 += KC; if (k0 >= K) break;
        stage(k0, by, bz, bx); k0 += KC;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 128 + a * 32 + 8 * (r / 4) + kh * 4 + (r % 4);
                const int col = n0 + wn * 64 + b * 32 + i32;
                C[(long long)row * N + col] = acc[a][b][r];
            }
}

int main() {
    const int M = 8192, N = 4096, K = 1024;
    std::vector<float> hA((size_t)M * K), hB((size_t)N * K), hC((size_t)M * N);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto& v : hA) v = rnd() * 1.3f;
    for (auto& v : hB) v = rnd() * 0.7f;
    float *A, *B, *C;
    CK(hipMalloc(&A, hA.size() * 4)); CK(hipMalloc(&B, hB.size() * 4)); CK(hipMalloc(&C, hC.size() * 4));
    CK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    // offline split of B into three bf16 planes
    std::vector<unsigned short> hBp((size_t)3 * N * K);
    for (size_t i = 0; i < hB.size(); ++i) {
        unsigned b; memcpy(&b, &hB[i], 4);
        const unsigned bh = b & 0xffff0000u; float fh; memcpy(&fh, &bh, 4);
        const float r1 = hB[i] - fh; unsigned b1; memcpy(&b1, &r1, 4);
        const unsigned bm = b1 & 0xffff0000u; float fm; memcpy(&fm, &bm, 4);
        const float r2 = r1 - fm; unsigned b2; memcpy(&b2, &r2, 4);
        hBp[i] = bh >> 16; hBp[hB.size() + i] = bm >> 16; hBp[2 * hB.size() + i] = b2 >> 16;
    }
    __bf16* Bp; CK(hipMalloc(&Bp, hBp.size() * 2)); CK(hipMemcpy(Bp, hBp.data(), hBp.size() * 2, hipMemcpyHostToDevice));
    // fragment-order packing of the planes for v3
    std::vector<unsigned short> hBf((size_t)3 * N * K);
    for (int nt = 0; nt < N / 32; ++nt)
        for (int s_ = 0; s_ < K / 16; ++s_)
            for (int pl = 0; pl < 3; ++pl)
                for (int l = 0; l < 64; ++l)
                    for (int e = 0; e < 8; ++e)
                        hBf[((((size_t)nt * (K / 16) + s_) * 3 + pl) * 64 + l) * 8 + e] =
                            hBp[(size_t)pl * N * K + (size_t)(nt * 32 + (l & 31)) * K + s_ * 16 + 8 * (l >> 5) + e];
    __bf16* Bf; CK(hipMalloc(&Bf, hBf.size() * 2)); CK(hipMemcpy(Bf, hBf.data(), hBf.size() * 2, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    unsigned long long* dtr; CK(hipMalloc(&dtr, 4 * 64 * 4 * 8)); CK(hipMemset(dtr, 0, 4 * 64 * 4 * 8));
    if (getenv("TRACE")) CK(hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &dtr, sizeof dtr));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm4<6>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BM * KCP4 * 4));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm4<9>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BM * KCP4 * 4));
    const int lds5 = 2 * 3 * BM5 * P2 * 2;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm5<6>), hipFuncAttributeMaxDynamicSharedMemorySize, lds5));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm5<9>), hipFuncAttributeMaxDynamicSharedMemorySize, lds5));
    for (int ver = 3; ver <= 6; ++ver)
    for (int terms : {6, 9}) {
        const dim3 grid(N / (ver == 6 ? 256 : BN), M / (ver == 5 ? BM5 : BM));
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            for (int i = 0; i < 10; ++i) {
                if (ver == 1) {
                    if (terms == 6) hipLaunchKernelGGL(gemm<6>, grid, dim3(256), 0, 0, A, B, C, M, N, K);
                    else hipLaunchKernelGGL(gemm<9>, grid, dim3(256), 0, 0, A, B, C, M, N, K);
                } else if (ver == 2) {
                    if (terms == 6) hipLaunchKernelGGL(gemm2<6>, grid, dim3(256), 0, 0, A, Bp, C, M, N, K);
                    else hipLaunchKernelGGL(gemm2<9>, grid, dim3(256), 0, 0, A, Bp, C, M, N, K);
                } else if (ver == 3) {
                    if (terms == 6) hipLaunchKernelGGL(gemm3<6>, grid, dim3(256), 0, 0, A, Bf, C, M, N, K);
                    else hipLaunchKernelGGL(gemm3<9>, grid, dim3(256), 0, 0, A, Bf, C, M, N, K);
                } else if (ver == 4) {
                    if (terms == 6) hipLaunchKernelGGL(gemm4<6>, grid, dim3(256), 2 * BM * KCP4 * 4, 0, A, Bf, C, M, N, K);
                    else hipLaunchKernelGGL(gemm4<9>, grid, dim3(256), 2 * BM * KCP4 * 4, 0, A, Bf, C, M, N, K);
                } else if (ver == 6) {
                    if (terms == 6) hipLaunchKernelGGL(gemm6<6>, grid, dim3(512), 0, 0, A, Bf, C, M, N, K);
                    else hipLaunchKernelGGL(gemm6<9>, grid, dim3(512), 0, 0, A, Bf, C, M, N, K);
                } else {
                    if (terms == 6) hipLaunchKernelGGL(gemm5<6>, grid, dim3(256), lds5, 0, A, Bf, C, M, N, K);
                    else hipLaunchKernelGGL(gemm5<9>, grid, dim3(256), lds5, 0, A, Bf, C, M, N, K);
                }
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 2) printf("v%d bf16 x %d: %.3f ms per GEMM -> %.1f fp32-equivalent TFLOP/s", ver, terms, ms / 10, 2.0 * M * N * K / (ms / 10 * 1e-3) / 1e12);
        }
        CK(hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost));
        double maxrel = 0, sumsq = 0, refsq = 0, fp32err = 0;
        for (int t = 0; t < 4000; ++t) {
            const int i = (t * 7919) % M, j = (t * 104729) % N;
            double ref = 0;
            float f32 = 0.f;
            for (int k = 0; k < K; ++k) { ref += (double)hA[(size_t)i * K + k] * hB[(size_t)j * K + k]; f32 = fmaf(hA[(size_t)i * K + k], hB[(size_t)j * K + k], f32); }
            const double d = hC[(size_t)i * N + j] - ref;
            sumsq += d * d; refsq += ref * ref;
            fp32err += (f32 - ref) * (f32 - ref);
            maxrel = fmax(maxrel, fabs(d) / (fabs(ref) + 1e-3));
        }
        printf(";  rms error / rms value %.3e (sequential fp32 fma chain: %.3e), max rel %.2e\n", sqrt(sumsq / refsq), sqrt(fp32err / refsq), maxrel);
    }
    if (getenv("TRACE")) {
        std::vector<unsigned long long> tr(4 * 64 * 4);
        CK(hipMemcpy(tr.data(), dtr, tr.size() * 8, hipMemcpyDeviceToHost));
        for (int w = 0; w < 4; w += 3) {
            printf("wave %d: k-step: [reads+wait] [mfma issue] [store/copy] [to next k-step incl. barrier]\n", w);
            for (int s_ = 8; s_ < 20; ++s_) {
                const unsigned long long* t = &tr[(w * 64 + s_) * 4];
                const unsigned long long* n = &tr[(w * 64 + s_ + 1) * 4];
                printf("  %2d: %5llu %5llu %5llu %5llu\n", s_, t[1] - t[0], t[2] - t[1], t[3] - t[2], n[0] - t[3]);
            }
        }
    }
    return 0;
}
