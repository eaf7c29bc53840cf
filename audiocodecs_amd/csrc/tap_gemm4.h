// tap_gemm v4: the fast tap-GEMM (tap_gemm.h documents the conv -> GEMM mapping; the generic kernel
// there remains the fallback for odd shapes).  Built on this round's measurements
// (profiles/r1_tapgemm_investigation.md): on gfx950 the fp32-input MFMA shares the fp32 vector
// lanes, so every VALU instruction -- from any wave of the SIMD -- costs ~4 cycles of matrix time.
// The main loop therefore carries almost no VALU work:
//   * inputs arrive already activated (ELU is applied once, in the producing layer's epilogue);
//   * each thread owns a fixed set of 16-byte staging slots; their global byte offsets and LDS
//     addresses are computed once per segment; per stage the loads are buffer_load_dwordx4 with
//     that constant VGPR offset and a scalar (SGPR) chunk offset;
//   * next stage's loads are issued before the MFMA phase and written to the other LDS buffer
//     after it: one barrier per stage;
//   * tiles that touch a clip edge (reflect / zero padding, ragged tail) take a slow, exact path
//     for their loads only -- two tiles per clip.
// Requirements (host-checked): every segment has s*cin % 32 == 0, 16-byte aligned bases, no length
// mask, and contiguous time steps (ts == cin) unless s == 1, where ts is simply the row pitch.
#pragma once
#include "tap_gemm.h"

namespace ac {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int WGM, int WGN, int WM, int WN>
struct Tap4Cfg {
    static constexpr int BM = WGM * WM * 16, BN = WGN * WN * 16, NT = WGM * WGN * 64;
    static constexpr int MAXJ = 8;
    static constexpr int A_ROWS = BM + MAXJ - 1;
    static constexpr int A_SLOTS = (A_ROWS * (KC / 4) + NT - 1) / NT;
    static constexpr int W_SLOTS = (BN * (KC / 4) + NT - 1) / NT;
    static constexpr int A_FLOATS = A_ROWS * KCP, W_FLOATS = BN * KCP;
    static constexpr int CP = BN + 4;
    static constexpr size_t main_bytes = (size_t)(2 * A_FLOATS + 2 * W_FLOATS) * 4;
    static constexpr size_t epi_bytes = (size_t)BM * CP * 4;
    static constexpr size_t lds_bytes = main_bytes > epi_bytes ? main_bytes : epi_bytes;
};

__device__ __forceinline__ f32x4 bufload16(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
}

#ifdef TAP4_TRACE
__device__ unsigned long long* g_tap4_trace;   // [slot][wave][stage][5]
#define TRC4(st, k) do { if (trace_on && lane == 0 && (st) < 64) g_tap4_trace[((trace_slot * 4 + wave) * 64 + (st)) * 5 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TRC4(st, k) do {} while (0)
#endif

template <int WGM, int WGN, int WM, int WN>
__global__ __launch_bounds__(WGM* WGN * 64) void tap_gemm4_kernel(const TapGemmParams p) {
    using Cfg = Tap4Cfg<WGM, WGN, WM, WN>;
    constexpr int BM = Cfg::BM, BN = Cfg::BN, NT = Cfg::NT;
    constexpr int A_SLOTS = Cfg::A_SLOTS, W_SLOTS = Cfg::W_SLOTS;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As0 = smem;
    float* Ws0 = smem + 2 * Cfg::A_FLOATS;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int li = lane & 15, kq = lane >> 4;
#ifdef TAP4_TRACE
    const bool trace_on = g_tap4_trace && (blockIdx.x == 700 || blockIdx.x == 701);
    const int trace_slot = blockIdx.x - 700;
    int trc_stage = 0;
#endif

    // XCD-aware tile order: workgroup b runs on XCD b % 8 (observed dispatch, used for speed only), each
    // XCD has its own L2.  Give every XCD a contiguous range of the (clip, m-tile, n-tile) list with the
    // n-tile fastest, so the n-tiles that re-read one activation slab hit the same L2 instead of
    // fetching it up to 8 times (PMC: 1.39x the algorithmic bytes before this remap).
    int id;
    {
        const int total = gridDim.x, q = total >> 3, r = total & 7, xcd = blockIdx.x & 7, pos = blockIdx.x >> 3;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;
    }
    const int nt = id % p.ntiles; id /= p.ntiles;
    const int mt = id % p.mtiles;
    const int b = id / p.mtiles;
    const int m0 = mt * BM, n0 = nt * BN;

    f32x4 acc[WM][WN];
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int c = 0; c < WN; ++c) acc[a][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- per-slot constants
    int a_lds[A_SLOTS], a_boff[A_SLOTS];
    int w_lds[W_SLOTS], w_boff[W_SLOTS];
#pragma unroll
    for (int i = 0; i < A_SLOTS; ++i) {
        const int e = tid + i * NT;
        a_lds[i] = (e / (KC / 4)) < Cfg::A_ROWS ? (e / (KC / 4)) * KCP + 4 * (e % (KC / 4)) : -1;
    }
#pragma unroll
    for (int i = 0; i < W_SLOTS; ++i) {
        const int e = tid + i * NT;
        const int n = e / (KC / 4), q = e % (KC / 4);
        w_lds[i] = n < BN ? n * KCP + 4 * q : -1;
        const int ng = min(n0 + (n < BN ? n : 0), p.N - 1);   // columns past N: read a valid row, never stored
        w_boff[i] = (ng * p.Ktot + 4 * q) * 4;
    }
    const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.N * p.Ktot * 4, 0x00020000);

    // ---- segment state (wave-uniform)
    int si = 0, c0 = 0, j = 0;
    int seg_J, seg_Cw, seg_kofs;
    int seg_tapoff = 0;     // dilated segments (DAC) reload the A slab per tap: byte offset of one tap step
    bool seg_reload = false;
    bool seg_interior;      // fast loads: per-slot constant row offsets (+ a zero mask on clip-edge tiles)
    unsigned a_zero = 0;    // bit i: slot i is padding that reads as zero (edge tiles of stride-1 segments)
    __amdgpu_buffer_rsrc_t a_rs;
    auto enter_segment = [&](int s_) {
        const TapSeg& sg = p.seg[s_];
        seg_J = sg.J;
        seg_Cw = sg.s * sg.cin;
        seg_kofs = sg.kofs;
        seg_reload = sg.dil != 1;
        // time steps touched by the tile: t = (m + j*dil)*s + tp - pad  (tap_gemm.h TapSeg)
        const long long lo = (long long)m0 * sg.s - sg.pad;
        const long long hi = (long long)(m0 + BM - 1 + (sg.J - 1) * sg.dil) * sg.s + (sg.s - 1) - sg.pad;
        const bool inside = lo >= 0 && hi < sg.L;
        // stride-1 undilated segments: a slot's source row does not depend on the chunk, so even clip-edge
        // tiles (reflect / zero rows, ragged tail; 2 of 6 tiles at 750 frames) keep the constant-offset loads
        seg_interior = inside || (sg.s == 1 && !seg_reload);
        a_zero = 0;
        const int tsf = sg.s == 1 ? (int)sg.ts : sg.cin;     // floats per time step
        seg_tapoff = sg.dil * sg.s * tsf * 4;
        a_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(sg.x + (long long)b * sg.bs), 0,
                                                 (int)(((long long)(sg.L - 1) * sg.ts + sg.cin) * 4), 0x00020000);
        const int R = seg_reload ? BM : BM + sg.J - 1;
#pragma unroll
        for (int i = 0; i < A_SLOTS; ++i) {
            const int e = tid + i * NT;
            const int row = e / (KC / 4), q = e % (KC / 4);
            long long t = (long long)(m0 + (row < R ? row : 0)) * sg.s - sg.pad;          // rows past R: any valid row
            if (!inside) {
                if (sg.s == 1 && !seg_reload) {
                    const long long jj = row < R ? src_index(sg, (int)t) : 0;             // [HF]:139-162 reflect / zero rule
                    if (jj < 0) a_zero |= 1u << i;
                    t = jj < 0 ? 0 : jj;
                } else {
                    t = 0;                                                                // unused: slow path
                }
            }
            a_boff[i] = (int)((t * tsf + 4 * q) * 4);
        }
    };
    f32x4 ra[A_SLOTS], rw[W_SLOTS];
    auto load_a = [&](int s_, int c_, int j_) {
        if (seg_interior) {
            const int soff = c_ * 4 + (seg_reload ? j_ * seg_tapoff : 0);
#pragma unroll
            for (int i = 0; i < A_SLOTS; ++i) ra[i] = bufload16(a_rs, a_boff[i], soff);
            // padding slots (a_zero) are cleared in store_a, after the MFMA phase: masking here would wait for
            // the loads right after issuing them (every chunk of a clip-edge tile; 2 of 6 tiles at 750 frames)
        } else {
            // exact edge handling ([HF]:139-162 reflect rule, zero pad of the transposed conv, ragged tail)
            const TapSeg& sg = p.seg[s_];
            const int R = seg_reload ? BM : BM + sg.J - 1;
            const int jr = seg_reload ? j_ * sg.dil : 0;
            const float* xb = sg.x + (long long)b * sg.bs;
#pragma unroll
            for (int i = 0; i < A_SLOTS; ++i) {
                const int e = tid + i * NT;
                const int row = e / (KC / 4), q = e % (KC / 4);
                const int c = c_ + 4 * q;
                const int tp = sg.cin_shift >= 0 ? (c >> sg.cin_shift) : (c / sg.cin);
                const long long jj = row < R ? src_index(sg, (m0 + row + jr) * sg.s + tp - sg.pad) : -1;
                ra[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (jj >= 0) ra[i] = *reinterpret_cast<const f32x4*>(xb + jj * sg.ts + (c - tp * sg.cin));
            }
        }
    };
    auto load_w = [&](int c_, int j_) {
        const int soff = (seg_kofs + j_ * seg_Cw + c_) * 4;
#pragma unroll
        for (int i = 0; i < W_SLOTS; ++i) rw[i] = bufload16(w_rs, w_boff[i], soff);
    };
    auto store_a = [&](float* dst) {
#pragma unroll
        for (int i = 0; i < A_SLOTS; ++i)
            if (a_lds[i] >= 0) *reinterpret_cast<f32x4*>(&dst[a_lds[i]]) = (a_zero & (1u << i)) ? f32x4{0.f, 0.f, 0.f, 0.f} : ra[i];
    };
    auto store_w = [&](float* dst) {
#pragma unroll
        for (int i = 0; i < W_SLOTS; ++i)
            if (w_lds[i] >= 0) *reinterpret_cast<f32x4*>(&dst[w_lds[i]]) = rw[i];
    };

    // ---- prologue: stage 0 into buffers 0
    enter_segment(0);
    load_a(0, 0, 0);
    load_w(0, 0);
    store_a(As0);
    store_w(Ws0);
    __syncthreads();
    int abuf = 0, wbuf = 0;
    const int a_frag = (wm * WM * 16 + li) * KCP + 4 * kq;   // + (a*16 + j)*KCP + ks*16
    const int w_frag = (wn * WN * 16 + li) * KCP + 4 * kq;

    for (;;) {
        // next stage coordinates (uniform)
        int nsi = si, nc0 = c0, nj = j + 1;
        bool new_chunk = false;
        if (nj == seg_J) {
            nj = 0;
            nc0 = c0 + KC;
            new_chunk = true;
            if (nc0 >= seg_Cw) { nc0 = 0; nsi = si + 1; }
        }
        const bool has_next = nsi < p.nseg;
        const int cur_j = seg_reload ? 0 : j;      // a reloaded slab already starts at the tap's first row
        TRC4(trc_stage, 0);
        if (has_next) {
            if (nsi != si) enter_segment(nsi);
            new_chunk = new_chunk || seg_reload;
            if (new_chunk) load_a(nsi, nc0, nj);
            load_w(nc0, nj);
        }
        TRC4(trc_stage, 1);
        // ---- MFMA over the current stage.  All fragments of the stage are read up front; the LDS
        // writes of the NEXT stage's data (other buffers) sit between the two k-steps, so the end
        // of the stage is just the barrier.
        const float* Ac = As0 + abuf * Cfg::A_FLOATS + a_frag + cur_j * KCP;
        const float* Wc = Ws0 + wbuf * Cfg::W_FLOATS + w_frag;
        f32x4 af[KC / 16][WM], bf[KC / 16][WN];
#pragma unroll
        for (int ks = 0; ks < KC / 16; ++ks) {
#pragma unroll
            for (int a = 0; a < WM; ++a) af[ks][a] = *reinterpret_cast<const f32x4*>(&Ac[a * 16 * KCP + ks * 16]);
#pragma unroll
            for (int c = 0; c < WN; ++c) bf[ks][c] = *reinterpret_cast<const f32x4*>(&Wc[c * 16 * KCP + ks * 16]);
        }
#pragma unroll
        for (int ks = 0; ks < KC / 16; ++ks) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int a = 0; a < WM; ++a)
#pragma unroll
                    for (int c = 0; c < WN; ++c)
                        acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ks][a][u], bf[ks][c][u], acc[a][c], 0, 0, 0);
            if (ks == 0 && has_next) {
                if (new_chunk) {
                    abuf ^= 1;
                    store_a(As0 + abuf * Cfg::A_FLOATS);
                }
                wbuf ^= 1;
                store_w(Ws0 + wbuf * Cfg::W_FLOATS);
            }
        }
        TRC4(trc_stage, 2);
        if (!has_next) break;
        TRC4(trc_stage, 3);
        __syncthreads();
        TRC4(trc_stage, 4);
#ifdef TAP4_TRACE
        ++trc_stage;
#endif
        si = nsi; c0 = nc0; j = nj;
    }

    // ---- epilogue through LDS: Cs[m][n] = acc + bias, then row-contiguous 16-byte stores of y / ELU(y)
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
    const long long yoff = (long long)b * p.y_bs;
    const bool post = p.gelu || p.scale || p.res || p.tanh_out;
    for (int e = tid; e < BM * (BN / 4); e += NT) {
        const int row = e / (BN / 4), q = e % (BN / 4);
        const int m = m0 + row, n = n0 + 4 * q;
        const long long fi = (long long)m * p.y_rs + n + p.y_off;      // flat index inside the item
        if (m < p.M && n < (p.n_valid ? p.n_valid : p.N) && (p.y_len == 0 || (fi >= 0 && fi < p.y_len))) {
            f32x4 v = *reinterpret_cast<const f32x4*>(&Cs[row * CP + 4 * q]);
            if (post) {
                if (p.gelu) { v.x = gelu1(v.x); v.y = gelu1(v.y); v.z = gelu1(v.z); v.w = gelu1(v.w); }
                if (p.scale) {
                    const f32x4 sc = *reinterpret_cast<const f32x4*>(p.scale + n);
                    v.x = __fmul_rn(sc.x, v.x); v.y = __fmul_rn(sc.y, v.y); v.z = __fmul_rn(sc.z, v.z); v.w = __fmul_rn(sc.w, v.w);
                }
                if (p.res) {
                    const f32x4 rv = *reinterpret_cast<const f32x4*>(p.res + (long long)b * p.res_bs + (long long)m * p.res_rs + n);
                    v.x = __fadd_rn(rv.x, v.x); v.y = __fadd_rn(rv.y, v.y); v.z = __fadd_rn(rv.z, v.z); v.w = __fadd_rn(rv.w, v.w);
                }
                if (p.tanh_out) { v.x = tanhf(v.x); v.y = tanhf(v.y); v.z = tanhf(v.z); v.w = tanhf(v.w); }
            }
            const long long o = yoff + fi;
            if (p.y) *reinterpret_cast<f32x4*>(p.y + o) = v;
            if (p.y_elu) {
                f32x4 w;
                if (p.alpha) {
                    const int c = n % p.alpha_n;
                    const f32x4 al = *reinterpret_cast<const f32x4*>(p.alpha + c), ai = *reinterpret_cast<const f32x4*>(p.alpha_inv + c);
                    w.x = snake1(v.x, al.x, ai.x); w.y = snake1(v.y, al.y, ai.y); w.z = snake1(v.z, al.z, ai.z); w.w = snake1(v.w, al.w, ai.w);
                } else {
                    w = elu4(v);
                }
                *reinterpret_cast<f32x4*>(p.y_elu + o) = w;
            }
        }
    }
}

}  // namespace ac
