// tap_gemm v2: same contraction as tap_gemm.h (see there for the conv -> GEMM mapping), restructured
// so that HBM/L2 latency is paid once per stage instead of once per 16-byte load:
//   * a stage = (K-chunk of 32 columns, tap j); all of a thread's loads for the NEXT stage are issued
//     back-to-back into registers before the MFMA phase of the current stage (software pipeline),
//     then ELU'd and written to the other LDS buffer after it; ONE barrier per stage;
//   * the activation slab of a chunk (BM + J - 1 rows) is staged once and reused by its J taps;
//   * the epilogue transposes the accumulators through LDS so that each output row leaves as
//     contiguous 16-byte stores (a 16x16x4 accumulator holds 4 rows x 16 columns per wave otherwise).
#pragma once
#include "tap_gemm.h"

namespace ac {

template <int WGM, int WGN, int WM, int WN>
struct TapCfg {
    static constexpr int BM = WGM * WM * 16, BN = WGN * WN * 16, NT = WGM * WGN * 64;
    static constexpr int MAXJ = 8;
    static constexpr int A_ROWS = BM + MAXJ - 1;
    static constexpr int A_SLOTS = (A_ROWS * (KC / 4) + NT - 1) / NT;   // float4 loads per thread per chunk
    static constexpr int W_SLOTS = (BN * (KC / 4) + NT - 1) / NT;
    static constexpr int A_FLOATS = A_ROWS * KCP, W_FLOATS = BN * KCP;
    static constexpr int CP = BN + 4;                                    // epilogue pitch
    static constexpr size_t main_bytes = (size_t)(2 * A_FLOATS + 2 * W_FLOATS) * 4;
    static constexpr size_t epi_bytes = (size_t)BM * CP * 4;
    static constexpr size_t lds_bytes = main_bytes > epi_bytes ? main_bytes : epi_bytes;
};


template <int WGM, int WGN, int WM, int WN>
__global__ __launch_bounds__(WGM* WGN * 64) void tap_gemm2_kernel(const TapGemmParams p) {
    using Cfg = TapCfg<WGM, WGN, WM, WN>;
    constexpr int BM = Cfg::BM, BN = Cfg::BN, NT = Cfg::NT;
    constexpr int A_SLOTS = Cfg::A_SLOTS, W_SLOTS = Cfg::W_SLOTS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As0 = smem;
    float* Ws0 = smem + 2 * Cfg::A_FLOATS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int li = lane & 15, kq = lane >> 4;

    int id = blockIdx.x;
    const int nt = id % p.ntiles; id /= p.ntiles;
    const int mt = id % p.mtiles;
    const int b = id / p.mtiles;
    const int m0 = mt * BM, n0 = nt * BN;

    f32x4 acc[WM][WN];
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int c = 0; c < WN; ++c) acc[a][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- stage iterator state: (segment si, chunk c0, tap j)
    int si = 0, c0 = 0, j = 0;
    f32x4 ra[A_SLOTS], rw[W_SLOTS];

    auto seg_cw = [&](int s_) { return p.seg[s_].s * p.seg[s_].cin; };

    // issue the activation-slab loads of chunk (s_, c_) into ra[]
    auto load_a = [&](int s_, int c_) {
        const TapSeg& sg = p.seg[s_];
        const int Cw = sg.s * sg.cin;
        const int R = BM + sg.J - 1;
        const float* xb = sg.x + (long long)b * sg.bs;
        float alen = 3.0e38f;
        if (sg.rel_len) alen = (float)sg.L * sg.rel_len[b];
#pragma unroll
        for (int i = 0; i < A_SLOTS; ++i) {
            const int e = tid + i * NT;
            const int row = e / (KC / 4), q = e % (KC / 4);
            const int c = c_ + 4 * q;
            ra[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (row < R && c < Cw) {
                const int r = m0 - (sg.J - 1) + row;
                const int tp = sg.cin_shift >= 0 ? (c >> sg.cin_shift) : (c / sg.cin);
                const int ci = c - tp * sg.cin;
                const long long jj = src_index(sg, r * sg.s + tp);
                if (jj >= 0 && (float)jj < alen) ra[i] = *reinterpret_cast<const f32x4*>(xb + jj * sg.ts + ci);
            }
        }
    };
    auto store_a = [&](int s_, float* dst) {
        const bool elu = p.seg[s_].elu != 0;
#pragma unroll
        for (int i = 0; i < A_SLOTS; ++i) {
            const int e = tid + i * NT;
            const int row = e / (KC / 4), q = e % (KC / 4);
            if (row < Cfg::A_ROWS) *reinterpret_cast<f32x4*>(&dst[row * KCP + 4 * q]) = elu ? elu4(ra[i]) : ra[i];
        }
    };
    auto load_w = [&](int s_, int c_, int j_) {
        const TapSeg& sg = p.seg[s_];
        const int Cw = sg.s * sg.cin;
        const long long kbase = (long long)sg.kofs + (long long)j_ * Cw + c_;
#pragma unroll
        for (int i = 0; i < W_SLOTS; ++i) {
            const int e = tid + i * NT;
            const int n = e / (KC / 4), q = e % (KC / 4);
            rw[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (n < BN && n0 + n < p.N && c_ + 4 * q < Cw)
                rw[i] = *reinterpret_cast<const f32x4*>(p.w + (long long)(n0 + n) * p.Ktot + kbase + 4 * q);
        }
    };
    auto store_w = [&](float* dst) {
#pragma unroll
        for (int i = 0; i < W_SLOTS; ++i) {
            const int e = tid + i * NT;
            const int n = e / (KC / 4), q = e % (KC / 4);
            if (n < BN) *reinterpret_cast<f32x4*>(&dst[n * KCP + 4 * q]) = rw[i];
        }
    };

    // ---- prologue: stage 0 into buffers 0
    load_a(0, 0);
    load_w(0, 0, 0);
    store_a(0, As0);
    store_w(Ws0);
    __syncthreads();
    int abuf = 0, wbuf = 0;

    for (;;) {
        // next stage coordinates
        int nsi = si, nc0 = c0, nj = j + 1;
        bool new_chunk = false;
        if (nj == p.seg[si].J) {
            nj = 0;
            nc0 = c0 + KC;
            new_chunk = true;
            if (nc0 >= seg_cw(si)) { nc0 = 0; nsi = si + 1; }
        }
        const bool has_next = nsi < p.nseg;
        if (has_next) {
            if (new_chunk) load_a(nsi, nc0);
            load_w(nsi, nc0, nj);
        }
        // ---- MFMA over the current stage
        const float* Ac = As0 + abuf * Cfg::A_FLOATS;
        const float* Wc = Ws0 + wbuf * Cfg::W_FLOATS;
#pragma unroll
        for (int ks = 0; ks < KC / 16; ++ks) {
            f32x4 af[WM], bf[WN];
#pragma unroll
            for (int a = 0; a < WM; ++a)
                af[a] = *reinterpret_cast<const f32x4*>(&Ac[((wm * WM + a) * 16 + li + j) * KCP + ks * 16 + 4 * kq]);
#pragma unroll
            for (int c = 0; c < WN; ++c)
                bf[c] = *reinterpret_cast<const f32x4*>(&Wc[((wn * WN + c) * 16 + li) * KCP + ks * 16 + 4 * kq]);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int a = 0; a < WM; ++a)
#pragma unroll
                    for (int c = 0; c < WN; ++c)
                        acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[a][u], bf[c][u], acc[a][c], 0, 0, 0);
        }
        if (!has_next) break;
        if (new_chunk) {
            abuf ^= 1;
            store_a(nsi, As0 + abuf * Cfg::A_FLOATS);
        }
        wbuf ^= 1;
        store_w(Ws0 + wbuf * Cfg::W_FLOATS);
        __syncthreads();
        si = nsi; c0 = nc0; j = nj;
    }

    // ---- epilogue through LDS: Cs[m][n] (pitch CP), then row-contiguous 16-byte stores
    __syncthreads();
    float* Cs = smem;
    constexpr int CP = Cfg::CP;
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int c = 0; c < WN; ++c) {
            const int n = (wn * WN + c) * 16 + li;
            const float bv = (p.bias && n0 + n < p.N) ? p.bias[n0 + n] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) Cs[((wm * WM + a) * 16 + kq * 4 + r) * CP + n] = acc[a][c][r] + bv;
        }
    __syncthreads();
    float* yb = p.y + (long long)b * p.y_bs;
    if ((p.N & 3) == 0 && (p.y_rs & 3) == 0 && (p.y_bs & 3) == 0) {
        for (int e = tid; e < BM * (BN / 4); e += NT) {
            const int row = e / (BN / 4), q = e % (BN / 4);
            const int m = m0 + row, n = n0 + 4 * q;
            if (m < p.M && n < p.N)
                *reinterpret_cast<f32x4*>(yb + (long long)m * p.y_rs + n) = *reinterpret_cast<const f32x4*>(&Cs[row * CP + 4 * q]);
        }
    } else {
        for (int e = tid; e < BM * BN; e += NT) {
            const int row = e / BN, cn = e % BN;
            const int m = m0 + row, n = n0 + cn;
            if (m < p.M && n < p.N) yb[(long long)m * p.y_rs + n] = Cs[row * CP + cn];
        }
    }
}

}  // namespace ac
