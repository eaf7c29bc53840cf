// tap_gemm v3: producer/consumer wave specialisation of the tap-GEMM (tap_gemm.h has the conv -> GEMM
// mapping).  PMC counters on v1/v2 (profiles/r1_tapgemm_pmc.md) showed the fp32 MFMA pipe ~50 % busy
// with the waves of co-resident blocks marching in convoy: every wave alternates a staging phase
// (address math, ELU, LDS writes, load waits) and an MFMA phase, and two waves on one SIMD end up in
// the same phase.  Here a workgroup has CW consumer waves that ONLY read fragments from LDS and
// issue v_mfma_f32_16x16x4_f32, and as many producer waves that ONLY move data: the SIMD's matrix
// pipe and its VALU/LDS/VMEM pipes run concurrently (separate pipes, MI355X_MICROARCH.md "Wave
// scheduling"), one raw s_barrier per stage, LDS double-buffered, producer loads for stage s+2 in
// flight across the barrier (no vmcnt drain: raw s_barrier + lgkmcnt(0) only).
#pragma once
#include "tap_gemm2.h"

namespace ac {

template <int WGM, int WGN, int WM, int WN>
struct Tap3Cfg {
    static constexpr int CW = WGM * WGN;           // consumer waves (= producer waves)
    static constexpr int BM = WGM * WM * 16, BN = WGN * WN * 16;
    static constexpr int NT = 2 * CW * 64, NP = CW * 64;
    static constexpr int MAXJ = 8;
    static constexpr int A_ROWS = BM + MAXJ - 1;
    static constexpr int A_SLOTS = (A_ROWS * (KC / 4) + NP - 1) / NP;
    static constexpr int W_SLOTS = (BN * (KC / 4) + NP - 1) / NP;
    static constexpr int A_FLOATS = A_ROWS * KCP, W_FLOATS = BN * KCP;
    static constexpr int CP = BN + 4;
    static constexpr size_t main_bytes = (size_t)(2 * A_FLOATS + 2 * W_FLOATS) * 4;
    static constexpr size_t epi_bytes = (size_t)BM * CP * 4;
    static constexpr size_t lds_bytes = main_bytes > epi_bytes ? main_bytes : epi_bytes;
};

__device__ __forceinline__ void lds_barrier() {
    // make this wave's LDS traffic visible, then meet the workgroup; outstanding global loads stay in flight
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

#ifdef TAP3_TRACE
__device__ unsigned long long* g_tap3_trace;   // [block-slot][role][stage][4]
#define TRC(role, st, k) do { if (trace_on && lane == 0 && (st) < 64) g_tap3_trace[((trace_slot * 2 + (role)) * 64 + (st)) * 4 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TRC(role, st, k) do {} while (0)
#endif

template <int WGM, int WGN, int WM, int WN>
__global__ __launch_bounds__(2 * WGM * WGN * 64) void tap_gemm3_kernel(const TapGemmParams p) {
    using Cfg = Tap3Cfg<WGM, WGN, WM, WN>;
    constexpr int BM = Cfg::BM, BN = Cfg::BN, NT = Cfg::NT, NP = Cfg::NP, CW = Cfg::CW;
    constexpr int A_SLOTS = Cfg::A_SLOTS, W_SLOTS = Cfg::W_SLOTS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As0 = smem;
    float* Ws0 = smem + 2 * Cfg::A_FLOATS;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave >= CW;
#ifdef TAP3_TRACE
    const bool trace_on = g_tap3_trace && (blockIdx.x == 0 || blockIdx.x == 700) && (wave == 0 || wave == CW);
    const int trace_slot = blockIdx.x == 0 ? 0 : 1;
#endif

    int id = blockIdx.x;
    const int nt = id % p.ntiles; id /= p.ntiles;
    const int mt = id % p.mtiles;
    const int b = id / p.mtiles;
    const int m0 = mt * BM, n0 = nt * BN;

    // total number of stages = sum over segments of chunks * taps
    int nstages = 0;
    for (int s_ = 0; s_ < p.nseg; ++s_) nstages += ((p.seg[s_].s * p.seg[s_].cin + KC - 1) / KC) * p.seg[s_].J;

    f32x4 acc[WM][WN];
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int c = 0; c < WN; ++c) acc[a][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (producer) {
#ifndef TAP3_NO_PRIO
        // producers carry few instructions but sit on the critical path of every stage: let them win
        // issue arbitration against the MFMA stream of the (older) consumer waves
        __builtin_amdgcn_s_setprio(3);
#endif
        const int pt = tid - NP;   // 0 .. NP-1
        f32x4 ra[A_SLOTS], rw[W_SLOTS];
        // Producers are starved of issue slots while the consumers' MFMA stream runs (a few fillers per
        // 32-cycle MFMA), so their per-stage instruction count is what bounds the kernel: everything
        // that does not change from stage to stage is hoisted into per-slot constants.
        int arow[A_SLOTS], aq[A_SLOTS], alds[A_SLOTS];     // slab row, 16-B column, LDS float offset
        int wn_[W_SLOTS], wq[W_SLOTS], wlds[W_SLOTS];
        long long woff[W_SLOTS];                            // (n0+n)*Ktot + 4q  (or -1: outside N)
#pragma unroll
        for (int i = 0; i < A_SLOTS; ++i) {
            const int e = pt + i * NP;
            arow[i] = e / (KC / 4);
            aq[i] = e % (KC / 4);
            alds[i] = arow[i] < Cfg::A_ROWS ? arow[i] * KCP + 4 * aq[i] : -1;
        }
#pragma unroll
        for (int i = 0; i < W_SLOTS; ++i) {
            const int e = pt + i * NP;
            wn_[i] = e / (KC / 4);
            wq[i] = e % (KC / 4);
            wlds[i] = wn_[i] < BN ? wn_[i] * KCP + 4 * wq[i] : -1;
            woff[i] = (wn_[i] < BN && n0 + wn_[i] < p.N) ? (long long)(n0 + wn_[i]) * p.Ktot + 4 * wq[i] : -1;
        }
        // cursor of the stage whose data is being LOADED, plus per-segment constants
        int si = 0, c0 = 0, j = 0;
        int seg_J = 0, seg_Cw = 0;
        bool seg_interior = false, seg_elu = false;
        const float* seg_xb = nullptr;
        int aoff[A_SLOTS];                                  // interior blocks: element offset of (row, 4q) at c0 = 0, or -1
        auto enter_segment = [&]() {
            const TapSeg& sg = p.seg[si];
            seg_J = sg.J;
            seg_Cw = sg.s * sg.cin;
            seg_elu = sg.elu != 0;
            seg_xb = sg.x + (long long)b * sg.bs;
            const int R = BM + sg.J - 1;
            // interior: every time index the slab touches is a real sample, rows are contiguous
            const long long lo = (long long)(m0 - (sg.J - 1)) * sg.s;
            const long long hi = (long long)(m0 + BM - 1) * sg.s + (sg.s - 1);
            seg_interior = lo >= 0 && hi < sg.L && sg.rel_len == nullptr && sg.ts == sg.cin;
#pragma unroll
            for (int i = 0; i < A_SLOTS; ++i)
                aoff[i] = (arow[i] < R) ? (m0 - (sg.J - 1) + arow[i]) * seg_Cw + 4 * aq[i] : -1;
        };
        auto advance = [&]() {
            if (++j == seg_J) {
                j = 0;
                c0 += KC;
                if (c0 >= seg_Cw) {
                    c0 = 0;
                    ++si;
                    if (si < p.nseg) enter_segment();
                }
            }
        };
        auto load_stage = [&](bool with_a) {
            if (with_a) {
                if (seg_interior) {
#pragma unroll
                    for (int i = 0; i < A_SLOTS; ++i) {
                        ra[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                        if (aoff[i] >= 0 && c0 + 4 * aq[i] < seg_Cw) ra[i] = *reinterpret_cast<const f32x4*>(seg_xb + (aoff[i] + c0));
                    }
                } else {
                    const TapSeg& sg = p.seg[si];
                    const int R = BM + sg.J - 1;
                    float alen = 3.0e38f;
                    if (sg.rel_len) alen = (float)sg.L * sg.rel_len[b];
#pragma unroll
                    for (int i = 0; i < A_SLOTS; ++i) {
                        const int c = c0 + 4 * aq[i];
                        const int r = m0 - (sg.J - 1) + arow[i];
                        const int tp = sg.cin_shift >= 0 ? (c >> sg.cin_shift) : (c / sg.cin);
                        const int ci = c - tp * sg.cin;
                        const long long jj = src_index(sg, r * sg.s + tp);
                        const bool ok = arow[i] < R && c < seg_Cw && jj >= 0 && (float)jj < alen;
                        ra[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                        if (ok) ra[i] = *reinterpret_cast<const f32x4*>(seg_xb + jj * sg.ts + ci);
                    }
                }
            }
            const long long kbase = (long long)p.seg[si].kofs + (long long)j * seg_Cw + c0;
#pragma unroll
            for (int i = 0; i < W_SLOTS; ++i) {
                rw[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (woff[i] >= 0 && c0 + 4 * wq[i] < seg_Cw) rw[i] = *reinterpret_cast<const f32x4*>(p.w + (woff[i] + kbase));
            }
        };
        auto store_stage = [&](bool with_a, bool elu, float* adst, float* wdst) {
            if (with_a) {
#pragma unroll
                for (int i = 0; i < A_SLOTS; ++i)
                    if (alds[i] >= 0) *reinterpret_cast<f32x4*>(&adst[alds[i]]) = elu ? elu4(ra[i]) : ra[i];
            }
#pragma unroll
            for (int i = 0; i < W_SLOTS; ++i)
                if (wlds[i] >= 0) *reinterpret_cast<f32x4*>(&wdst[wlds[i]]) = rw[i];
        };
        // stage 0 -> buffers 0
        int abuf = 0, wbuf = 0;
        enter_segment();
        load_stage(true);
        store_stage(true, seg_elu, As0, Ws0);
        // loads for stage 1 stay in flight across the barrier
        bool pend_a = false, pend_elu = false;
        if (nstages > 1) {
            advance();
            pend_a = j == 0;
            pend_elu = seg_elu;
            load_stage(pend_a);
        }
        lds_barrier();
        for (int s = 0; s < nstages; ++s) {
            TRC(1, s, 0);
            if (s + 1 < nstages) {
                if (pend_a) abuf ^= 1;
                wbuf ^= 1;
#ifdef TAP3_TRACE
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                TRC(1, s, 1);
#endif
                store_stage(pend_a, pend_elu, As0 + abuf * Cfg::A_FLOATS, Ws0 + wbuf * Cfg::W_FLOATS);
                TRC(1, s, 2);
                if (s + 2 < nstages) {
                    advance();
                    pend_a = j == 0;
                    pend_elu = seg_elu;
                    load_stage(pend_a);
                }
            }
            TRC(1, s, 3);
            lds_barrier();
        }
    } else {
        const int wm = wave / WGN, wn = wave % WGN;
        const int li = lane & 15, kq = lane >> 4;
        int si = 0, c0 = 0, j = 0, abuf = 0, wbuf = 0;
        lds_barrier();
        for (int s = 0; s < nstages; ++s) {
            TRC(0, s, 0);
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
            // advance the consumer's view of (segment, chunk, tap)
            if (++j == p.seg[si].J) {
                j = 0;
                c0 += KC;
                abuf ^= 1;
                if (c0 >= p.seg[si].s * p.seg[si].cin) { c0 = 0; ++si; }
            }
            wbuf ^= 1;
            TRC(0, s, 1);
            lds_barrier();
            TRC(0, s, 2);
        }
    }

    // ---- epilogue through LDS (all waves): Cs[m][n], then row-contiguous 16-byte stores
    float* Cs = smem;
    constexpr int CP = Cfg::CP;
    if (!producer) {
        const int wm = wave / WGN, wn = wave % WGN;
        const int li = lane & 15, kq = lane >> 4;
#pragma unroll
        for (int a = 0; a < WM; ++a)
#pragma unroll
            for (int c = 0; c < WN; ++c) {
                const int n = (wn * WN + c) * 16 + li;
                const float bv = (p.bias && n0 + n < p.N) ? p.bias[n0 + n] : 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) Cs[((wm * WM + a) * 16 + kq * 4 + r) * CP + n] = acc[a][c][r] + bv;
            }
    }
    __syncthreads();
    float* yb = p.y + (long long)b * p.y_bs;
    if ((p.N & 3) == 0 && (p.y_rs & 3) == 0 && (p.y_bs & 3) == 0 && (reinterpret_cast<uintptr_t>(p.y) & 15) == 0) {
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
