// Probe for round 2: split-operand GEMM (tap_gemm6.h arithmetic: 3 bf16 terms per fp32 operand, 6 partial products) when
// the ACTIVATION operand arrives PRE-SPLIT from HBM as three bf16 planes (written by its producer's epilogue) instead of
// being split inside the GEMM's staging loop:
//   * A planes [3][M][K] bf16 go HBM -> LDS by LDS-DMA (buffer_load ... lds, 16 B per lane), no VGPR staging, no VALU,
//     no ds_write; the LDS image is XOR-swizzled through the per-lane SOURCE address (rows of BK bf16, 16-byte chunks
//     permuted by a row-dependent mask) so the fragment reads (ds_read_b128) are bank-conflict-free;
//   * B (weights) as in tap_gemm6: pre-split, fragment-packed, L2 -> registers one stage ahead.
// C[M][N] = A[M][K] * B[N][K]^T.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 bf16x_pre.hip -o bf16x_pre_bin
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

__device__ unsigned long long* g_clk = nullptr;   // [blocks][2]: shader-clock ticks, 100 MHz real-time ticks per workgroup

// WMT x WNT 32x32 tiles per wave; waves arranged WGM x WGN; BK = k per stage
template <int WGM, int WGN, int WMT, int WNT, int BK, int OCC>
__global__ __launch_bounds__(WGM* WGN * 64, OCC) void gemm7(const __bf16* __restrict__ Ap, const __bf16* __restrict__ Bf, float* __restrict__ C,
                                                               int M, int N, int K) {
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    constexpr int NW = WGM * WGN, NT = NW * 64, BM = WGM * WMT * 32, BN = WGN * WNT * 32;
    constexpr int CH = BK / 8;                      // 16-byte chunks per row
    constexpr int ROWB = BK * 2;                    // bytes per LDS row
    constexpr int PLANE = BM * ROWB;                // bytes per plane
    constexpr int RPI = 1024 / ROWB;                // rows per DMA instruction (64 lanes x 16 B)
    constexpr int INSTR = 3 * BM / RPI;             // DMA instructions per stage
    constexpr int IPW = INSTR / NW;                 // per wave
    static_assert(INSTR % NW == 0, "");
    constexpr int KS = BK / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];     // [2][3][BM][ROWB]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int i32 = lane & 31, kh = lane >> 5;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int ksteps = K / 16;

    f32x16 acc[WMT][WNT];
#pragma unroll
    for (int a = 0; a < WMT; ++a)
#pragma unroll
        for (int b = 0; b < WNT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // ---- DMA plan: instruction j of this wave covers plane pj, rows rj .. rj + RPI - 1; lane -> (row, slot)
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc((void*)Ap, 0, (int)((long long)3 * M * K * 2), 0x00020000);
    int a_voff[IPW];
    int a_ldsoff[IPW];
#pragma unroll
    for (int t = 0; t < IPW; ++t) {
        const int j = wave * IPW + t;
        const int pl = j / (BM / RPI), rb = (j % (BM / RPI)) * RPI;
        const int row = rb + lane / CH, slot = lane % CH;
        const int sw = CH == 4 ? ((row >> 2) & 3) : ((row >> 1) & 7);
        const int src = slot ^ sw;
        a_voff[t] = (int)(((long long)pl * M + m0 + row) * K * 2) + src * 16;
        a_ldsoff[t] = pl * PLANE + rb * ROWB;       // wave-uniform
    }
    auto dma = [&](int buf, int k0) {
#pragma unroll
        for (int t = 0; t < IPW; ++t)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ars, (__attribute__((address_space(3))) void*)(smem + buf * 3 * PLANE + a_ldsoff[t]), 16,
                                                     a_voff[t], k0 * 2, 0, 0);
    };
    // ---- fragment read offsets (bytes inside a plane) for k-step ks: row (wm*WMT + a)*32 + i32, chunk (2 ks + kh) ^ sw
    const int frow = (wm * WMT * 32 + i32) * ROWB;
    const int fsw = CH == 4 ? ((i32 >> 2) & 3) : ((i32 >> 1) & 7);
    // ---- B fragments
    const __bf16* bbase = Bf + ((long long)((n0 + wn * WNT * 32) / 32) * ksteps) * (3 * 64 * 8) + lane * 8;
    auto bload = [&](int s_, bf16x8 (&bf)[KS][3][WNT]) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int b = 0; b < WNT; ++b)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    bf[ks][pl][b] = *reinterpret_cast<const bf16x8*>(bbase + (((long long)b * ksteps + s_ + ks) * 3 + pl) * (64 * 8));
    };
    auto compute = [&](int buf, const bf16x8 (&bf)[KS][3][WNT]) {
        const char* base = smem + buf * 3 * PLANE + frow;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8 af[3][WMT];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int a = 0; a < WMT; ++a)
                    af[pl][a] = *reinterpret_cast<const bf16x8*>(base + pl * PLANE + a * 32 * ROWB + (((2 * ks + kh) ^ fsw) * 16));
#pragma unroll
            for (int a = 0; a < WMT; ++a)
#pragma unroll
                for (int b = 0; b < WNT; ++b) {
                    f32x16 c = acc[a][b];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][a], bf[ks][2][b], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2][a], bf[ks][0][b], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][a], bf[ks][1][b], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][a], bf[ks][1][b], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][a], bf[ks][0][b], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][a], bf[ks][0][b], c, 0, 0, 0);
                    acc[a][b] = c;
                }
        }
    };
    bf16x8 b0[KS][3][WNT], b1[KS][3][WNT];
    dma(0, 0);
    bload(0, b0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int stages = K / BK;
    for (int s = 0; s < stages; s += 2) {
        if (s + 1 < stages) { dma(1, (s + 1) * BK); bload((s + 1) * KS, b1); }
        compute(0, b0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (s + 1 >= stages) break;
        if (s + 2 < stages) { dma(0, (s + 2) * BK); bload((s + 2) * KS, b0); }
        compute(1, b1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < WMT; ++a)
#pragma unroll
        for (int b = 0; b < WNT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + (wm * WMT + a) * 32 + 8 * (r / 4) + kh * 4 + (r % 4);
                const int col = n0 + (wn * WNT + b) * 32 + i32;
                C[(long long)row * N + col] = acc[a][b][r];
            }
    if (g_clk && threadIdx.x == 0) {
        const unsigned long long bid = (unsigned long long)blockIdx.y * gridDim.x + blockIdx.x;
        g_clk[2 * bid] = __builtin_amdgcn_s_memtime() - t0;
        g_clk[2 * bid + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
#endif
}

template <int WGM, int WGN, int WMT, int WNT, int BK, int OCC>
void run(const char* name, const __bf16* Ap, const __bf16* Bf, float* C, int M, int N, int K, const std::vector<float>& hA, const std::vector<float>& hB,
         std::vector<float>& hC) {
    constexpr int BM = WGM * WMT * 32, BN = WGN * WNT * 32;
    const size_t lds = (size_t)2 * 3 * BM * BK * 2;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm7<WGM, WGN, WMT, WNT, BK, OCC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const dim3 grid(N / BN, M / BM);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((gemm7<WGM, WGN, WMT, WNT, BK, OCC>), grid, dim3(WGM * WGN * 64), lds, 0, Ap, Bf, C, M, N, K);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    CK(hipGetLastError());
    {   // average shader clock while this kernel runs: s_memtime ticks per 100 MHz real-time tick
        const size_t nb = (size_t)grid.x * grid.y;
        unsigned long long* dclk; CK(hipMalloc(&dclk, nb * 16)); CK(hipMemset(dclk, 0, nb * 16));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_clk), &dclk, sizeof dclk));
        hipLaunchKernelGGL((gemm7<WGM, WGN, WMT, WNT, BK, OCC>), grid, dim3(WGM * WGN * 64), lds, 0, Ap, Bf, C, M, N, K);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> hc(nb * 2); CK(hipMemcpy(hc.data(), dclk, nb * 16, hipMemcpyDeviceToHost));
        double st = 0, rt = 0; for (size_t i = 0; i < nb; ++i) { st += hc[2 * i]; rt += hc[2 * i + 1]; }
        unsigned long long* z = nullptr; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_clk), &z, sizeof z)); CK(hipFree(dclk));
        printf("  [s_memtime / s_memrealtime = %.3f -> counter runs at %.0f MHz if real-time is 100 MHz] ", st / rt, 100.0 * st / rt);
    }
    CK(hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost));
    double sumsq = 0, refsq = 0;
    for (int t = 0; t < 3000; ++t) {
        const int i = (t * 7919) % M, j = (t * 104729) % N;
        double ref = 0;
        for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)i * K + k] * hB[(size_t)j * K + k];
        const double d = hC[(size_t)i * N + j] - ref;
        sumsq += d * d; refsq += ref * ref;
    }
    printf("%-34s M%d N%d K%d: %.3f ms -> %.1f fp32-equivalent TFLOP/s; rms err / rms value %.3e\n", name, M, N, K, ms / 10,
           2.0 * M * N * K / (ms / 10 * 1e-3) / 1e12, sqrt(sumsq / refsq));
}

static void split3(float v, unsigned short (&o)[3]) {
    unsigned b; memcpy(&b, &v, 4);
    const unsigned bh = b & 0xffff0000u; float fh; memcpy(&fh, &bh, 4);
    const float r1 = v - fh; unsigned b1; memcpy(&b1, &r1, 4);
    const unsigned bm = b1 & 0xffff0000u; float fm; memcpy(&fm, &bm, 4);
    const float r2 = r1 - fm; unsigned b2; memcpy(&b2, &r2, 4);
    o[0] = bh >> 16; o[1] = bm >> 16; o[2] = b2 >> 16;
}

int main() {
    for (int K : {1024, 4096}) {
        const int M = 8192, N = 4096;
        std::vector<float> hA((size_t)M * K), hB((size_t)N * K), hC((size_t)M * N);
        unsigned s = 12345u;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
        for (auto& v : hA) v = rnd() * 1.3f;
        for (auto& v : hB) v = rnd() * 0.7f;
        std::vector<unsigned short> hAp((size_t)3 * M * K), hBf((size_t)3 * N * K);
        for (size_t i = 0; i < hA.size(); ++i) {
            unsigned short t[3]; split3(hA[i], t);
            for (int pl = 0; pl < 3; ++pl) hAp[(size_t)pl * M * K + i] = t[pl];
        }
        for (int nt = 0; nt < N / 32; ++nt)
            for (int s_ = 0; s_ < K / 16; ++s_)
                for (int l = 0; l < 64; ++l)
                    for (int e = 0; e < 8; ++e) {
                        unsigned short t[3]; split3(hB[(size_t)(nt * 32 + (l & 31)) * K + s_ * 16 + 8 * (l >> 5) + e], t);
                        for (int pl = 0; pl < 3; ++pl) hBf[((((size_t)nt * (K / 16) + s_) * 3 + pl) * 64 + l) * 8 + e] = t[pl];
                    }
        __bf16 *Ap, *Bf; float* C;
        CK(hipMalloc(&Ap, hAp.size() * 2)); CK(hipMalloc(&Bf, hBf.size() * 2)); CK(hipMalloc(&C, hC.size() * 4));
        CK(hipMemcpy(Ap, hAp.data(), hAp.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(Bf, hBf.data(), hBf.size() * 2, hipMemcpyHostToDevice));
        run<2, 2, 2, 2, 32, 2>("128x128, 4 waves, BK 32, occ 2", Ap, Bf, C, M, N, K, hA, hB, hC);
        run<2, 2, 2, 2, 64, 1>("128x128, 4 waves, BK 64, occ 1", Ap, Bf, C, M, N, K, hA, hB, hC);
        run<2, 4, 2, 2, 32, 1>("128x256, 8 waves, BK 32, occ 1", Ap, Bf, C, M, N, K, hA, hB, hC);
        run<4, 2, 2, 2, 32, 1>("256x128, 8 waves, BK 32, occ 1", Ap, Bf, C, M, N, K, hA, hB, hC);
        run<2, 2, 4, 2, 32, 1>("256x128, 4 waves (128x64), BK 32", Ap, Bf, C, M, N, K, hA, hB, hC);
        CK(hipFree(Ap)); CK(hipFree(Bf)); CK(hipFree(C));
    }
    return 0;
}
