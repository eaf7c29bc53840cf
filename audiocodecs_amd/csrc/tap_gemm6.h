// tap_gemm6: the tap-GEMM (tap_gemm.h: conv -> GEMM mapping, padding rules, epilogue terms) on the fp16 matrix pipe with
// fp32 fidelity -- split16.h arithmetic: every fp32 operand, scaled by a power of two, is written as hi + lo, two fp16 terms,
// and a product is accumulated in fp32 from 3 of its 4 exact partial products (lo hi, hi lo, hi hi).
// v_mfma_f32_32x32x16_f16 runs 16x the fp32 MFMA rate, so 3 terms cost 0.19x of the fp32-MFMA time: the kernel's roofline moves
// from 157 to 833 fp32-equivalent TFLOP/s.  What binds such a kernel is operand traffic (LDS, L1).  Hence
//   * the weight operand never touches LDS: it is split and packed OFFLINE in MFMA B-fragment order
//       wp[n-tile of 32][k-step of 16][plane 2][lane 64][8 fp16] of the SCALED rows  (a wave-load = 1 KB contiguous per plane)
//     and loaded from L2 straight into registers one k-step ahead;
//   * the activation operand is loaded exactly like in tap_gemm4 (buffer loads with per-slot constant offsets,
//     clip-edge rules) and split ONCE, while it is staged into a double-buffered LDS slab of two fp16 planes.
// Workgroup = 4 or 8 waves as WGM x WGN, wave tile 32*WMT x 32*WN, BM = 128, KC = 32 (two k-steps) per stage:
//   <1,4,4,1> BN = 128;  <1,4,4,2> BN = 256;  <1,8,4,1> BN = 256;  <2,2,2,3> BN = 192, <4,1,1,3> BN = 96 (DAC), <2,2,2,1> BN = 64.
// Requirements beyond tap_gemm4's: N % 32 == 0 and every segment's kofs % 32 == 0 (k-steps align with stages).
// The exact-product kernel tap_gemm4 stays selectable (AC_PRECISION_FP32_EXACT / AC_GEMM=fp32) and serves every other shape.
// (Rounds 1-3 also instantiated NP = 3 -- three bf16 planes, 6 products -- and NP = 1, a rounded-bf16 side mode; both removed in
//  round 4.  The template argument NP = 2 stays: it is part of the kernel names the committed profiles carry.)
#pragma once
#include "tap_gemm4.h"
#include "split16.h"
#include <type_traits>

namespace ac {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every outstanding global access (its release
// fence) -- in a persistent tile loop that is a wait for the previous tile's output stores to reach HBM, once per tile
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

constexpr int T6_PITCH = 40;        // bf16 per LDS row (32 + 8): 80-byte rows keep 16-byte alignment, spread banks

// HALO: rows of the A slab beyond BM.  7 = the taps of a k <= 8 conv on consecutive rows; T6_DIL_HALO = the dilated k7 convs of
// DAC's residual units (6 x 9 rows) read from ONE slab per chunk like any other conv, instead of a slab reload per tap
// (13 - 22 % slower per launch, profiles/r3_tapgemm_trace.md).  The larger slab costs LDS (59 KB instead of 43 KB: two workgroups per CU).
constexpr int T6_DIL_HALO = 56;
template <int WGM, int WGN, int WMT, int WN, int HALO = 7>
struct Tap6Cfg {
    static_assert((WGM * WGN == 4 || WGM * WGN == 8) && (WGM * WMT == 4 || WGM * WMT == 8), "4 or 8 waves, 128 or 256 rows");
    static constexpr int BM = 32 * WGM * WMT, BN = 32 * WGN * WN, NT = 64 * WGM * WGN;
    static constexpr int A_ROWS = BM + HALO;
    static constexpr int A_SLOTS = (A_ROWS * (KC / 4) + NT - 1) / NT;
    static constexpr int PLANE = A_ROWS * T6_PITCH;                 // bf16 elements per plane
    static constexpr int CP = 32 * WGN + 4;      // epilogue staging: one wave column tile per pass
    // the epilogue stages EH 32-row tiles of the workgroup's rows at a time (WGM = 1: two of the four -- half the staging tile)
    static constexpr int EH = WGM == 1 && WMT == 4 ? 2 : WMT;
    static constexpr size_t epi_bytes = (size_t)32 * WGM * EH * CP * 4;
    static constexpr size_t lds_for(int np) {               // two buffers x two planes (np = 2, split16.h)
        const size_t main_bytes = (size_t)2 * np * PLANE * 2;
        return main_bytes > epi_bytes ? main_bytes : epi_bytes;
    }
    static constexpr size_t lds_bytes = lds_for(2);
};
// workgroups per CU the kernel is compiled for: the 128 x 32 wave tile in split16 arithmetic runs the lean main loop (one
// fragment set) and fits three -- the epilogue of a workgroup (as long as its output takes to reach HBM) is then covered by two
// others' main loops (profiles/r2_tapgemm_variants.md)
template <int WGM, int WGN, int WMT, int WN, int NP>
constexpr int tap6_occupancy() {
    return WGM * WGN == 8 ? 1 : (NP == 2 && ((WMT * WN == 4 && WGM == 1) || (WGM == 2 && WGN == 2 && WMT == 2 && WN == 1)) ? 3 : 2);   // (64 x 32 wave tiles, N % 64: 156 VGPRs as they are)
}

// NP = 2: split16.h -- two fp16 planes per operand, 3 partial products, per-clip / per-output-channel power-of-two scales.
// Weight image [n-tile of 32][k-step of 16][plane 2][lane 64][8 fp16] of the SCALED rows; needs seg[].amax and winv.
#ifdef T6_TRACE   // developer build: phase stamps of wave 1 of one workgroup (p.clk[4..15])
#define T6_PHASE(k) do { if (t6_ph) p.clk[4 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#endif
// ---- the epilogue of a tile (shared by tap_gemm6_kernel and tap_gemm8_kernel): accumulators -> bias, 2^-s, activation flavours,
// residual / LayerScale / GELU / tanh terms, amax words -> HBM.  `smem` is the workgroup's whole dynamic LDS (the staged form reuses it
// as its staging tile after a workgroup barrier).
template <int WGM, int WGN, int WMT, int WN, int HALO>
__device__ __forceinline__ void tap6_epilogue(const TapGemmParams& p, f32x16 (&acc)[WMT][WN], float* smem, const int b, const int m0, const int n0,
                                              const float a_inv, const bool rowmode, const unsigned long long clk_t0, const unsigned long long clk_r0, const bool t6_ph) {
    using Cfg = Tap6Cfg<WGM, WGN, WMT, WN, HALO>;
    constexpr int NP = 2, BM = Cfg::BM, BN = Cfg::BN, NT = Cfg::NT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int i32 = lane & 31, kh = lane >> 5;
    (void)t6_ph; (void)BM; (void)BN;
#ifdef T6_TRACE
    T6_PHASE(7);
#endif
    // ---- direct epilogue (split16, plain conv outputs): value r of a 32 x 32 accumulator tile is row 8 (r / 4) + 4 kh + r % 4,
    // column lane & 31 -- for a fixed r the 64 lanes hold two rows x 32 consecutive channels, i.e. two 128-byte segments: stored as
    // they are with one 4-byte buffer store per value (row offset in the scalar offset, rows past the clip fall outside the
    // descriptor's range).  No LDS, no barrier: the staged epilogue below cost 24 % of the kernel (profiles/r2_tapgemm_variants.md).
    if (NP == 2 && p.epi_direct) {
        const long long yoff = (long long)b * p.y_bs;
        const int rs4 = (int)p.y_rs * 4, out_bytes = p.M * rs4;
        // a null output is a descriptor of zero records: its stores fall outside the range check and are dropped
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y ? p.y + yoff : nullptr), 0, p.y ? out_bytes : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t re = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y_elu ? p.y_elu + yoff : nullptr), 0, p.y_elu ? out_bytes : 0, 0x00020000);
        // residual operand (DAC residual units): the same two-rows-x-128-bytes pattern as the stores, one 4-byte load per value
        const int rr4 = (int)p.res_rs * 4;
        const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)(p.res ? p.res + (long long)b * p.res_bs : nullptr), 0, p.res ? p.M * rr4 : 0, 0x00020000);
        unsigned omax = 0;
        // The value loop exists in straight-line copies -- activated flavour (none / ELU / Snake) x residual x (tile inside the
        // clip / last tile) -- chosen ONCE: with the conditions tested per value the compiler emitted four scalar branches around
        // every store and the epilogue took ~170 cycles per value (23 k cycles of a 63 k-cycle tile of EnCodec's up-sampling layers;
        // the LDS-staged epilogue of a DAC 1 x 1 conv, residual loads waited for one by one: 48 k of 67 k cycles;
        // profiles/r3_tapgemm_trace.md).  Inside a copy a value is fma, (residual add), amax, store, (activation, amax, store);
        // the residual values of the NEXT 32 x 32 tile are requested before the current tile is worked on.
        auto values = [&](auto act_tag, auto res_tag, auto is_full) {
            constexpr int ACT = decltype(act_tag)::value;             // 0: raw only, 1: + ELU flavour, 2: + Snake flavour
            constexpr bool RES = decltype(res_tag)::value, FULL = decltype(is_full)::value;
            float rv[2][16];
            auto res_load = [&](int c, int a, float (&dst)[16]) {
                const int ng = n0 + (wn * WN + c) * 32 + i32;
                const int mrow = m0 + (wm * WMT + a) * 32 + 4 * kh;
                const int voff = mrow * rr4 + ng * 4;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    dst[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rr, voff, (8 * (r / 4) + (r % 4)) * rr4, 0));
            };
            if (RES) res_load(0, 0, rv[0]);
#pragma unroll
            for (int c = 0; c < WN; ++c) {
                const int ng = n0 + (wn * WN + c) * 32 + i32;
                const float bv = p.bias ? p.bias[ng] : 0.f;
                const float iv = a_inv * p.winv[ng];
                float al = 0.f, ai = 0.f;
                if (ACT == 2) { const int ca = ng % p.alpha_n; al = p.alpha[ca]; ai = p.alpha_inv[ca]; }
#ifdef T6_TRACE
                if (c == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); T6_PHASE(8); }
                if (c == 1) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); T6_PHASE(10); }
#endif
#pragma unroll
                for (int a = 0; a < WMT; ++a) {
                    const int mrow = m0 + (wm * WMT + a) * 32 + 4 * kh;
                    const int voff = mrow * rs4 + ng * 4;
                    const int cur = (c * WMT + a) & 1;
                    if (RES && c * WMT + a + 1 < WN * WMT) res_load((c * WMT + a + 1) / WMT, (c * WMT + a + 1) % WMT, rv[cur ^ 1]);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int dr = 8 * (r / 4) + (r % 4);
                        float v = __fmaf_rn(acc[a][c][r], iv, bv);
                        if (RES) v = __fadd_rn(rv[cur][r], v);
                        const bool in = FULL || mrow + dr < p.M;
                        if (in) amax_acc(omax, v);
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ry, voff, dr * rs4, 0);
                        if (ACT == 1) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(elu1(v)), re, voff, dr * rs4, 0);   // (ELU never exceeds |v|)
                        if (ACT == 2) {
                            const float w = snake1(v, al, ai);
                            if (in) amax_acc(omax, w);                                                                        // (Snake can)
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(w), re, voff, dr * rs4, 0);
                        }
                    }
                }
            }
        };
        const bool full = m0 + BM <= p.M;
#define T6_VALUES(ACT, RES) do { if (full) values(std::integral_constant<int, ACT>{}, std::integral_constant<bool, RES>{}, std::true_type{}); \
                                 else values(std::integral_constant<int, ACT>{}, std::integral_constant<bool, RES>{}, std::false_type{}); } while (0)
        if (BN < 128 && p.alpha) {          // (Snake layers of 128+ channels go through the staged epilogue: run_tap)
            if constexpr (BN < 128) { if (p.res) T6_VALUES(2, true); else T6_VALUES(2, false); }
        } else if (p.y_elu) {
            T6_VALUES(1, false);
        } else {
            T6_VALUES(0, false);
        }
#undef T6_VALUES
#ifdef T6_TRACE
        T6_PHASE(11);
#endif
        if (p.amax_out) amax_flush(omax, amax_at(p.amax_out, b));
#ifdef T6_TRACE
        T6_PHASE(9);
#endif
        if (p.clk && tid == 0) {
            atomicAdd(&p.clk[0], (unsigned long long)(__builtin_amdgcn_s_memtime() - clk_t0));
            atomicAdd(&p.clk[1], (unsigned long long)(__builtin_amdgcn_s_memrealtime() - clk_r0));
        }
        return;
    }

    // ---- epilogue through LDS (as tap_gemm4), one chunk per wave column tile c: [BM][32 * WGN] staged at a time, so the
    // staging tile never exceeds the main loop's LDS (a 128 x 256 tile keeps two workgroups per CU).  C layout of 32x32:
    // column = lane & 31, row = 8*(r/4) + 4*kh + r%4.  Staging column q <-> global column n0 + (q/32)*32*WN + 32*c + q%32.
    float* Cs = smem;
    constexpr int CP = Cfg::CP, CW = 32 * WGN;
    const long long yoff = (long long)b * p.y_bs;
    const bool post = p.gelu || p.scale || p.res || p.tanh_out;
    const int nvalid = p.n_valid ? p.n_valid : p.N;
    unsigned omax = 0;
    constexpr int EH = Cfg::EH, ER = 32 * WGM * EH;       // 32-row tiles / rows staged per pass
    const bool fastpass = NP == 2 && p.y_len == 0 && p.y_off == 0 && nvalid == p.N && m0 + BM <= p.M && n0 + BN <= p.N && (ER * (CW / 4)) % NT == 0 &&
                          (!p.res || ((long long)p.M * p.res_rs * 4 < 0x7fffffffLL));       // (the residual goes through a 32-bit buffer offset)
#pragma unroll
    for (int c = 0; c < WN; ++c)
#pragma unroll
    for (int a0 = 0; a0 < WMT; a0 += EH) {
        // the residual rows of this pass are requested BEFORE the pass is staged: they arrive under the barriers and LDS writes
        // (the store loop used to wait for each of them in turn: 48 k of the 67 k cycles of a DAC 1 x 1 tile, profiles/r3_tapgemm_trace.md)
        constexpr int EITER = (ER * (CW / 4)) / NT;
        // The residual rows and (row mode) the rows' amax words are requested two iterations ahead of the store loop -- WITHOUT a condition
        // around any request: through buffer descriptors of zero records when the layer has no residual / is not in row mode (such a load
        // returns zeros without touching memory), and with the index clamped instead of `if (i + 2 < EITER)`.  A register that is loaded on
        // one path and carried on the other becomes a copy at the join, and the copy waits for the load: round 5 found `s_waitcnt vmcnt(0)`
        // behind the residual add of EVERY iteration -- the request issued two lines earlier, and every store before it, waited for on the
        // spot (2 340 ticks per iteration of Mimi's o-projection against 850 for a layer without residual, profiles/r5r_*).
        const __amdgpu_buffer_rsrc_t r_res = __builtin_amdgcn_make_buffer_rsrc((void*)(p.res ? p.res + (long long)b * p.res_bs : nullptr), 0,
                                                                                 p.res ? (int)(((long long)(p.M - 1) * p.res_rs + p.N) * 4) : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t r_row = __builtin_amdgcn_make_buffer_rsrc((void*)(rowmode ? p.seg[0].amax : nullptr), 0, rowmode ? p.M * 4 : 0, 0x00020000);
        auto row_of = [&](int i) {
            const int row = (tid + i * NT) / (CW / 4);
            return m0 + (row / (32 * EH)) * (32 * WMT) + a0 * 32 + row % (32 * EH);
        };
        const int fq = tid % (CW / 4), fn = n0 + (fq / 8) * (32 * WN) + 32 * c + 4 * (fq % 8);
        auto res_at = [&](int i) -> f32x4 { return bufload16(r_res, (row_of(i) * (int)p.res_rs + fn) * 4, 0); };
        auto roww_at = [&](int i) -> unsigned { return (unsigned)__builtin_amdgcn_raw_buffer_load_b32(r_row, row_of(i) * 4, 0, 0); };
        f32x4 rn1 = f32x4{0.f, 0.f, 0.f, 0.f}, rn2 = rn1;              // the residual rows two iterations ahead of the store loop
        unsigned rw1 = 0, rw2 = 0;
        f32x4 fb4 = f32x4{0.f, 0.f, 0.f, 0.f}, fsc4 = f32x4{1.f, 1.f, 1.f, 1.f}, fal = fb4, fai = fb4;
        if (fastpass) {
            rn1 = res_at(0); rn2 = res_at(EITER > 1 ? 1 : 0);
            rw1 = roww_at(0); rw2 = roww_at(EITER > 1 ? 1 : 0);
            if (rowmode) { if (p.bias) fb4 = *reinterpret_cast<const f32x4*>(p.bias + fn); }
            if (p.scale) fsc4 = *reinterpret_cast<const f32x4*>(p.scale + fn);
            if (p.alpha && p.y_elu) { const int ca = fn % p.alpha_n; fal = *reinterpret_cast<const f32x4*>(p.alpha + ca); fai = *reinterpret_cast<const f32x4*>(p.alpha_inv + ca); }
        }
        __syncthreads();
#pragma unroll
        for (int a = a0; a < a0 + EH; ++a) {
            const int ng = n0 + (wn * WN + c) * 32 + i32;
            const float bv = (p.bias && ng < p.N) ? p.bias[ng] : 0.f;
            const float iv = NP == 2 ? a_inv * p.winv[ng] : 1.f;     // exact: powers of two
            const float bq = rowmode ? 0.f : bv;                     // row mode: the row's 2^-s and the bias follow in the store pass
#pragma unroll
            for (int r = 0; r < 16; ++r)
                Cs[((wm * EH + a - a0) * 32 + 8 * (r / 4) + 4 * kh + (r % 4)) * CP + wn * 32 + i32] = NP == 2 ? __fmaf_rn(acc[a][c][r], iv, bq) : acc[a][c][r] + bv;
        }
        __syncthreads();
#ifdef T6_TRACE
        if (c == 0 && a0 == 0) T6_PHASE(8);          // staged epilogue: first pass staged
#endif
        if (fastpass) {
            // the full-tile case (every epilogue flavour): no bounds; residual rows and row words requested two iterations ahead,
            // the per-column vectors already in registers -- the generic loop below waits for each of those loads in turn, which
            // was 48 k of the 67 k cycles of a DAC 1 x 1 tile and 150 k of the 250 k cycles of a Mimi o-projection tile
            // (profiles/r3_tapgemm_trace.md)
            // The loop runs in PAIRS with one register set per parity, each set reloaded (for iteration i + 2) right behind its last use: a
            // rotation rn1 = rn2, rn2 = load in a rolled loop became copies at the back edge, and the copy of the register just requested
            // waits for it (`s_waitcnt vmcnt(0)` per iteration, as the conditional request did before).
            auto one = [&](int i, f32x4& rn, unsigned& rw) {
                const int row = (tid + i * NT) / (CW / 4);
                const int m = m0 + (row / (32 * EH)) * (32 * WMT) + a0 * 32 + row % (32 * EH);
                const long long o = yoff + (long long)m * p.y_rs + fn;
                const int inext = i + 2 < EITER ? i + 2 : EITER - 1;       // (unconditional, clamped: see above)
                f32x4 v = *reinterpret_cast<const f32x4*>(&Cs[row * CP + 4 * fq]);
                if (rowmode) {
                    const float ri = s16_pow2(-s16_exponent(rw));
                    v = f32x4{__fmaf_rn(v.x, ri, fb4.x), __fmaf_rn(v.y, ri, fb4.y), __fmaf_rn(v.z, ri, fb4.z), __fmaf_rn(v.w, ri, fb4.w)};
                }
                rw = roww_at(inext);
                if (p.gelu) { v.x = gelu1(v.x); v.y = gelu1(v.y); v.z = gelu1(v.z); v.w = gelu1(v.w); }
                if (p.scale) { v.x = __fmul_rn(fsc4.x, v.x); v.y = __fmul_rn(fsc4.y, v.y); v.z = __fmul_rn(fsc4.z, v.z); v.w = __fmul_rn(fsc4.w, v.w); }
                if (p.res) { v.x = __fadd_rn(rn.x, v.x); v.y = __fadd_rn(rn.y, v.y); v.z = __fadd_rn(rn.z, v.z); v.w = __fadd_rn(rn.w, v.w); }
                rn = res_at(inext);
                if (p.tanh_out) { v.x = tanhf(v.x); v.y = tanhf(v.y); v.z = tanhf(v.z); v.w = tanhf(v.w); }
                if (p.amax_out) amax_acc4(omax, v);
                if (p.y) *reinterpret_cast<f32x4*>(p.y + o) = v;
                if (p.y_elu) {
                    f32x4 w;
                    if (p.alpha) {
                        w.x = snake1(v.x, fal.x, fai.x); w.y = snake1(v.y, fal.y, fai.y); w.z = snake1(v.z, fal.z, fai.z); w.w = snake1(v.w, fal.w, fai.w);
                        if (p.amax_out) amax_acc4(omax, w);
                    } else {
                        w = elu4(v);
                    }
                    *reinterpret_cast<f32x4*>(p.y_elu + o) = w;
                }
                if (p.amax_out_rows) {      // the CW / 4 lanes that share this row (every lane runs every iteration)
                    unsigned rmax = 0;
                    amax_acc4(rmax, v);
                    rmax = group_max_u32<CW / 4>(rmax);
                    if (fq == 0 && rmax) atomicMax(p.amax_out_rows + m, rmax);
                }
            };
            for (int i = 0; i + 1 < EITER; i += 2) { one(i, rn1, rw1); one(i + 1, rn2, rw2); }
            if (EITER & 1) one(EITER - 1, rn1, rw1);
        } else
        for (int e = tid; e < ER * (CW / 4); e += NT) {
            const int row = e / (CW / 4), q = e % (CW / 4);
            // staged row -> row of the workgroup tile: wave row group row / (32 EH), then the pass's EH tiles
            const int m = m0 + (row / (32 * EH)) * (32 * WMT) + a0 * 32 + row % (32 * EH), n = n0 + (q / 8) * (32 * WN) + 32 * c + 4 * (q % 8);
            const long long fi = (long long)m * p.y_rs + n + p.y_off;
            unsigned rmax = 0;
            if (m < p.M && n < nvalid && (p.y_len == 0 || (fi >= 0 && fi < p.y_len))) {
                f32x4 v = *reinterpret_cast<const f32x4*>(&Cs[row * CP + 4 * q]);
                if (rowmode) {
                    const float ri = s16_pow2(-s16_exponent(p.seg[0].amax[m]));
                    const f32x4 b4 = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
                    v = f32x4{__fmaf_rn(v.x, ri, b4.x), __fmaf_rn(v.y, ri, b4.y), __fmaf_rn(v.z, ri, b4.z), __fmaf_rn(v.w, ri, b4.w)};
                }
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
                if (p.amax_out) amax_acc4(omax, v);
                if (NP == 2 && p.amax_out_rows) amax_acc4(rmax, v);
                if (p.y) *reinterpret_cast<f32x4*>(p.y + o) = v;
                if (p.y_elu) {
                    f32x4 w;
                    if (p.alpha) {
                        const int ca = n % p.alpha_n;
                        const f32x4 al = *reinterpret_cast<const f32x4*>(p.alpha + ca), ai = *reinterpret_cast<const f32x4*>(p.alpha_inv + ca);
                        w.x = snake1(v.x, al.x, ai.x); w.y = snake1(v.y, al.y, ai.y); w.z = snake1(v.z, al.z, ai.z); w.w = snake1(v.w, al.w, ai.w);
                        if (p.amax_out) amax_acc4(omax, w);      // Snake can exceed |v| (ELU cannot)
                    } else {
                        w = elu4(v);
                    }
                    *reinterpret_cast<f32x4*>(p.y_elu + o) = w;
                }
            }
            if (NP == 2 && p.amax_out_rows) {      // the CW / 4 lanes that share this row (every lane runs every iteration)
                rmax = group_max_u32<CW / 4>(rmax);
                if (q == 0 && m < p.M && rmax) atomicMax(p.amax_out_rows + m, rmax);
            }
        }
#ifdef T6_TRACE
        if (c == 0 && a0 == 0) T6_PHASE(10);         // staged epilogue: first pass stored
#endif
    }
#ifdef T6_TRACE
    T6_PHASE(11);
#endif
    if (p.amax_out) amax_flush(omax, amax_at(p.amax_out, b));
#ifdef T6_TRACE
    T6_PHASE(9);
#endif
    if (p.clk && tid == 0) {   // shader clock while this workgroup lived: sum of ticks / sum of 100 MHz real-time ticks
        atomicAdd(&p.clk[0], (unsigned long long)(__builtin_amdgcn_s_memtime() - clk_t0));
        atomicAdd(&p.clk[1], (unsigned long long)(__builtin_amdgcn_s_memrealtime() - clk_r0));
    }
}

// what the main loop hands to the epilogue of its tile
struct Tap6Tile {
    int b, m0, n0;
    float a_inv;
    bool rowmode;
    unsigned long long clk_t0, clk_r0;
    bool t6_ph;
};

// ---- the main loop of a tile (prologue, K loop): tap_gemm6_kernel = this + tap6_epilogue; dac_unit6_kernel (dac_unit6.h) = this with
// SWAP + a second product + tap6_epilogue.  SWAP: the MFMA's operands exchanged -- the accumulators hold the TRANSPOSED 32 x 32 tiles
// (lane = row of the tile, register r = column 8 (r / 4) + 4 kh + r % 4), same products, same order.
template <int WGM, int WGN, int WMT, int WN, int HALO, bool SWAP>
__device__ __forceinline__ void tap6_mainloop(const TapGemmParams& p, const __bf16* __restrict__ wp, float* smem, f32x16 (&acc)[WMT][WN], Tap6Tile& tl) {
    constexpr int NP = 2;
    using Cfg = Tap6Cfg<WGM, WGN, WMT, WN, HALO>;
    constexpr int BM = Cfg::BM, BN = Cfg::BN, NT = Cfg::NT, A_SLOTS = Cfg::A_SLOTS, PLANE = Cfg::PLANE;
    unsigned long long clk_t0 = 0, clk_r0 = 0;
    if (p.clk) { clk_t0 = __builtin_amdgcn_s_memtime(); clk_r0 = __builtin_amdgcn_s_memrealtime(); }
    if (p.stagger && (int)blockIdx.x < 256 * tap6_occupancy<WGM, WGN, WMT, WN, NP>()) {
        const unsigned long long wait = (unsigned long long)((blockIdx.x * 0x9E3779B1u) >> 24) * (unsigned)p.stagger >> 8;
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        while (__builtin_amdgcn_s_memtime() - t0 < wait) __builtin_amdgcn_s_sleep(32);
    }
    __bf16* As0 = reinterpret_cast<__bf16*>(smem);            // [2 buffers][NPL planes][A_ROWS][T6_PITCH]
    constexpr int NPL = 2;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int i32 = lane & 31, kh = lane >> 5;                // MFMA operand: row / column i32, k = 8*kh .. 8*kh + 7

    int id;
    {   // XCD-aware tile order (tap_gemm4.h)
        const int total = gridDim.x, q = total >> 3, r = total & 7, xcd = blockIdx.x & 7, pos = blockIdx.x >> 3;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + pos;
    }
    const int nt = id % p.ntiles; id /= p.ntiles;
    const int mt = id % p.mtiles;
    const int b = id / p.mtiles;
    const int m0 = mt * BM, n0 = nt * BN;
    // The clip's amax word(s) are REQUESTED here, as soon as the clip is known, and used when the first chunk is split (store_a): the
    // ISA of round 4 had `global_load_dword` / `s_waitcnt vmcnt(0)` back to back in front of the tile's first activation request -- one
    // exposed L2 round trip per tile (round-5 ISA reading; the kernel-argument s_loads in front of it are hipcc's own business).
    // (NO request under a condition -- a loaded register that is carried on the other path becomes a copy at the join, and the copy waits
    //  for the load: the second segment's word is the first segment's again when there is none; in row mode seg[0].amax is the row-word
    //  array and b = 0, so the word read here exists and is simply not used)
    const unsigned am_e0 = *amax_at(p.seg[0].amax, b);
    const unsigned am_e1 = *amax_at(p.seg[p.nseg > 1 ? 1 : 0].amax, b);
#ifdef T6_TRACE   // developer build: phase stamps of wave 1 of one workgroup (p.clk[4..15]), stage stamps below
    const bool t6_ph = p.clk && mt == p.mtiles / 2 && nt == 0 && b == p.B / 2 && tid == 64;      // an interior tile
    if (t6_ph) p.clk[4] = clk_t0;
    T6_PHASE(1);
#endif

#pragma unroll
    for (int a = 0; a < WMT; ++a)
#pragma unroll
        for (int c = 0; c < WN; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;

    // slot i of a thread = element tid + i * NT of the slab = (row tid / 8 + (NT / 8) i, 16-byte column tid % 8): the LDS offsets of a
    // thread's slots are NT / 8 rows apart (ONE register + constants; a per-slot array cost the 128 x 32 arrangement three spilled
    // registers at its 168-register budget, and a kernel with scratch is not safe to replay from a hipGraph -- DESIGN.md section 1);
    // only the last slot can fall outside the slab
    static_assert(NT % (KC / 4) == 0, "slots are whole rows apart");
    int a_boff[A_SLOTS];
    const int a_lds0 = (tid / (KC / 4)) * T6_PITCH + 4 * (tid % (KC / 4));
    const bool last_slot_ok = (tid + (A_SLOTS - 1) * NT) / (KC / 4) < Cfg::A_ROWS;

    // ---- segment state (wave-uniform); identical to tap_gemm4
    int si = 0, c0 = 0, j = 0;
    int seg_J, seg_Cw, seg_kofs;
    int seg_tapoff = 0;
    bool seg_reload = false;
    int seg_rowstep = 1;                // slab rows between the taps of a segment (its dilation when the taps share one slab)
    bool seg_interior;
    unsigned a_zero = 0;
    __amdgpu_buffer_rsrc_t a_rs;
    auto enter_segment = [&](int s_) {
        const TapSeg& sg = p.seg[s_];
        seg_J = sg.J;
        seg_Cw = sg.s * sg.cin;
        seg_kofs = sg.kofs;
        seg_reload = sg.dil != 1 && (sg.s != 1 || (sg.J - 1) * sg.dil > HALO);     // dilated taps that fit the slab read it like any others
        seg_rowstep = seg_reload ? 1 : sg.dil;
        const long long lo = (long long)m0 * sg.s - sg.pad;
        const long long hi = (long long)(m0 + BM - 1 + (sg.J - 1) * sg.dil) * sg.s + (sg.s - 1) - sg.pad;
        const bool inside = lo >= 0 && hi < sg.L;
        seg_interior = inside || (sg.s == 1 && !seg_reload);
        a_zero = 0;
        const int tsf = sg.s == 1 ? (int)sg.ts : sg.cin;
        seg_tapoff = sg.dil * sg.s * tsf * 4;
        a_rs = __builtin_amdgcn_make_buffer_rsrc((void*)(sg.x + (long long)b * sg.bs), 0,
                                                 (int)(((long long)(sg.L - 1) * sg.ts + sg.cin) * 4), 0x00020000);
        const int R = seg_reload ? BM : BM + (sg.J - 1) * sg.dil;
#pragma unroll
        for (int i = 0; i < A_SLOTS; ++i) {
            const int e = tid + i * NT;
            const int row = e / (KC / 4), q = e % (KC / 4);
            long long t = (long long)(m0 + (row < R ? row : 0)) * sg.s - sg.pad;
            if (!inside) {
                if (sg.s == 1 && !seg_reload) {
                    const long long jj = row < R ? src_index(sg, (int)t) : 0;
                    if (jj < 0) a_zero |= 1u << i;
                    t = jj < 0 ? 0 : jj;
                } else {
                    t = 0;
                }
            }
            a_boff[i] = (int)((t * tsf + 4 * q) * 4);
        }
    };
    // split16: one power-of-two scale for the clip's activation operand(s) (both segments share the accumulator)
    // (row mode: one scale per ROW of the merged row matrix instead -- a_rsc[slot])
    // (row mode is a one-tap, undilated affair -- run_tap --: the wide-slab instantiations never see it, and their per-slot scale
    //  registers collapse to one; they spilled 2 - 7 registers with the array)
    constexpr bool CAN_ROWMODE = HALO == 7;
    constexpr int NRSC = CAN_ROWMODE ? A_SLOTS : 1;
    float a_scale = 1.f, a_inv = 1.f;
    float a_rsc[NRSC];
    const bool rowmode = CAN_ROWMODE && NP == 2 && p.amax_rows;
    unsigned am_rows[NRSC];              // row mode: the rows' words, requested here (unconditionally: word 0 otherwise), used in set_scales
#pragma unroll
    for (int i = 0; i < NRSC; ++i) {
        const int m = m0 + (tid + i * NT) / (KC / 4);
        am_rows[i] = p.seg[0].amax[rowmode ? (m < p.M ? m : p.M - 1) : 0];
    }
    auto set_scales = [&]() {
        if (!rowmode) {
            // (through a VGPR-constrained asm: hipcc otherwise moves the wave-uniform words to SGPRs -- v_readfirstlane, and the wait with it --
            //  right behind their loads)
            unsigned a0 = am_e0, a1 = am_e1;
            asm volatile("" : "+v"(a0), "+v"(a1));
            const int se = s16_exponent(a1 > a0 ? a1 : a0);
            a_scale = s16_pow2(se);
            a_inv = s16_pow2(-se);
        }
#pragma unroll
        for (int i = 0; i < NRSC; ++i) a_rsc[i] = rowmode ? s16_pow2(s16_exponent(am_rows[i])) : a_scale;
    };
    f32x4 ra[A_SLOTS];
    // Exactly A_SLOTS buffer loads, whatever the stage needs (see `stage` below for why the COUNT must not depend on the path):
    //   interior tile : the per-slot offsets of enter_segment + one scalar offset for the chunk / tap
    //   clip-edge tile: per-slot offsets from the padding rule (src_index); rows the rule zero-fills aim past the last record
    //   `live` false  : the stage stays on its chunk -- every slot aims past the last record
    // (a buffer load past num_records returns zeros without touching memory; the range check sees the VGPR offset only)
    constexpr int A_OOB = 0x7fff0000;
    auto load_a = [&](int s_, int c_, int j_, bool live) {
#ifdef T6_ABL_NOALOAD   // (ablation build, wrong results: every activation request aims past the descriptor's last record)
        live = false;
#endif
        int voff[A_SLOTS], soff = 0;
        if (seg_interior || !live) {
            // (readfirstlane: the chunk / tap offset IS wave-uniform, but the compiler could not prove it and wrapped every one of the
            //  stage's activation loads in a waterfall loop -- v_readfirstlane, compare, saveexec, load, loop: ~8 instructions and a
            //  serialisation point per load; round-4 ISA reading, cdna_hip_programming.md T20)
            soff = __builtin_amdgcn_readfirstlane(live ? c_ * 4 + (seg_reload ? j_ * seg_tapoff : 0) : 0);
#pragma unroll
            for (int i = 0; i < A_SLOTS; ++i) voff[i] = live ? a_boff[i] : A_OOB;
        } else {
            const TapSeg& sg = p.seg[s_];
            const int R = seg_reload ? BM : BM + (sg.J - 1) * sg.dil;
            const int jr = seg_reload ? j_ * sg.dil : 0;
#pragma unroll
            for (int i = 0; i < A_SLOTS; ++i) {
                const int e = tid + i * NT;
                const int row = e / (KC / 4), q = e % (KC / 4);
                const int c = c_ + 4 * q;
                const int tp = sg.cin_shift >= 0 ? (c >> sg.cin_shift) : (c / sg.cin);
                const long long jj = row < R ? src_index(sg, (m0 + row + jr) * sg.s + tp - sg.pad) : -1;
                voff[i] = jj < 0 ? A_OOB : (int)((jj * sg.ts + (c - tp * sg.cin)) * 4);
            }
        }
#pragma unroll
        for (int i = 0; i < A_SLOTS; ++i) ra[i] = bufload16(a_rs, voff[i], soff);
    };
    // split 4 fp32 into the two fp16 planes (split16.h: scaled value = hi + lo) and store them
    auto store_a = [&](__bf16* dst) {
#ifdef T6_ABL_NOSTORE   // (ablation build, wrong results: no split, no slab write)
#pragma unroll
        for (int i = 0; i < A_SLOTS; ++i) asm volatile("" ::"v"(ra[i]));
        return;
#endif
#pragma unroll
        for (int i = 0; i < A_SLOTS; ++i)
            if (i + 1 < A_SLOTS || last_slot_ok) {
                const f32x4 v = (a_zero & (1u << i)) ? f32x4{0.f, 0.f, 0.f, 0.f} : ra[i];
                split16_store4s(v, a_rsc[CAN_ROWMODE ? i : 0], dst, PLANE, a_lds0 + i * (NT / (KC / 4)) * T6_PITCH);
            }
    };
    // B fragments of this wave's WN column tiles for ONE k-step: [plane][c]; k-step index inside the packed rows
    const int ksteps = p.Ktot >> 4;
    constexpr int WPL = 2;    // planes in the weight image
    const __bf16* wbase = wp + ((long long)((n0 + wn * 32 * WN) >> 5) * ksteps) * (WPL * 64 * 8) + lane * 8;
    auto load_b = [&](int s_, bf16x8 (&bf)[3][WN]) {
#ifdef T6_ABL_NOBLOAD   // (ablation build, wrong results: no weight fragment is fetched)
        asm volatile("" : "+v"(bf[0][0]), "+v"(bf[1][0]));
        return;
#endif
#pragma unroll
        for (int c = 0; c < WN; ++c)
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
                bf[pl][c] = *reinterpret_cast<const bf16x8*>(wbase + (((long long)c * ksteps + s_) * WPL + pl) * (64 * 8));
    };

    // ---- prologue
#ifdef T6_TRACE
    T6_PHASE(2);
#endif
    enter_segment(0);
#ifdef T6_TRACE
    T6_PHASE(3);
#endif
    load_a(0, 0, 0, true);
    // three rotating B register sets: a stage uses (U0, U1) for its two k-steps and loads the NEXT stage's k-steps
    // into (S, U0) -- each load is issued two k-steps before its use (one k-step is shorter than an L2 round trip)
    bf16x8 bx[3][WN], by[3][WN], bz[3][WN];
    load_b(seg_kofs >> 4, bx);
    load_b((seg_kofs >> 4) + 1, by);
#ifdef T6_TRACE
    T6_PHASE(4);
#endif
    __builtin_amdgcn_sched_barrier(0);      // (the scale arithmetic -- the first use of the amax words -- stays behind the requests)
    set_scales();
    store_a(As0);
#ifdef T6_TRACE
    T6_PHASE(5);
#endif
    __syncthreads();
    int abuf = 0;
    const int a_frag = (wm * WMT * 32 + i32) * T6_PITCH + 8 * kh;    // + (a*32 + j)*T6_PITCH + ks*16, + plane*PLANE
    auto read_a = [&](const __bf16* Ac, int ks, bf16x8 (&af)[3][WMT]) {
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
#pragma unroll
            for (int a = 0; a < WMT; ++a) af[pl][a] = *reinterpret_cast<const bf16x8*>(Ac + pl * PLANE + a * 32 * T6_PITCH + ks * 16);
    };
    auto mfma_step = [&](const bf16x8 (&af)[3][WMT], const bf16x8 (&bf)[3][WN]) {
#pragma unroll
        for (int a = 0; a < WMT; ++a)
#pragma unroll
            for (int c = 0; c < WN; ++c) {
#ifdef T6_ABL_NOMFMA    // (ablation build, wrong results: the fragment reads stay, the MFMAs go)
                asm volatile("" ::"v"(af[0][a]), "v"(af[1][a]), "v"(bf[0][c]), "v"(bf[1][c]));
                continue;
#endif
                f32x16 v = acc[a][c];
                // lo hi, hi lo, hi hi (small terms first)
                if constexpr (SWAP) {
                    v = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bf[0][c]), __builtin_bit_cast(f16x8, af[1][a]), v, 0, 0, 0);
                    v = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bf[1][c]), __builtin_bit_cast(f16x8, af[0][a]), v, 0, 0, 0);
                    acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bf[0][c]), __builtin_bit_cast(f16x8, af[0][a]), v, 0, 0, 0);
                } else {
                v = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[1][a]), __builtin_bit_cast(f16x8, bf[0][c]), v, 0, 0, 0);
                v = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[0][a]), __builtin_bit_cast(f16x8, bf[1][c]), v, 0, 0, 0);
                acc[a][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[0][a]), __builtin_bit_cast(f16x8, bf[0][c]), v, 0, 0, 0);
                }
            }
    };
#ifdef T6_TRACE
    int t6_stage = 0;
    const bool t6_on = p.clk && mt == p.mtiles / 2 && nt == 0 && b == p.B / 2 && lane == 0 && t6_stage < 16;
#define T6_STAMP(k) do { if (t6_on && t6_stage < 16) p.clk[16 + (wave * 16 + t6_stage) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#endif
    // LEAN (wave tile of 6-8 accumulator tiles: 128 x 64, 64 x 96): 96-128 accumulator registers leave room for ONE fragment set and TWO
    // weight sets only -- a weight set is reloaded right after the k-step that used it (one k-step ahead of its next use), the
    // A fragments of a k-step are read just before its MFMAs (the co-resident workgroup covers the LDS latency).
    constexpr bool LEAN = WMT * WN >= 6 || (WGM == 1 && tap6_occupancy<WGM, WGN, WMT, WN, NP>() == 3);
    // one stage; returns true when it was the last one.
    // EVERY stage issues the same loads in the same order -- A_SLOTS activation loads, then a weight set after each k-step --
    // whether or not it needs them: a stage that stays on its A chunk points the activation loads past the end of the buffer
    // resource (they return zeros without touching memory, into registers nobody reads), the last stage reloads k-step 0.
    // The reason is s_waitcnt: vmcnt counts loads in issue order, so "wait for this k-step's weights but not for the younger
    // loads" can only be encoded when the number of younger loads is the same on every path into the wait.  With `if
    // (new_chunk)` / `if (has_next)` around the loads the compiler has to assume the fewest, and round 2's kernel waited for the
    // weight set it had JUST requested before every second k-step (vmcnt(0): one exposed L2 round trip per stage) and for the
    // first activation load before the first MFMA of every new-chunk stage (profiles/r3_tapgemm_trace.md: s_memtime stamps + ISA).
    auto stage = [&](bf16x8 (&u0)[3][WN], bf16x8 (&u1)[3][WN], bf16x8 (&sp)[3][WN]) -> bool {
#ifdef T6_TRACE
        T6_STAMP(6);
#endif
        int nsi = si, nc0 = c0, nj = j + 1;
        bool new_chunk = false;
        if (nj == seg_J) {
            nj = 0;
            nc0 = c0 + KC;
            new_chunk = true;
            if (nc0 >= seg_Cw) { nc0 = 0; nsi = si + 1; }
        }
        const bool has_next = nsi < p.nseg;
        const int cur_j = seg_reload ? 0 : j * seg_rowstep;
        int s_next = 0;
        if (has_next) {
            if (nsi != si) enter_segment(nsi);
            new_chunk = new_chunk || seg_reload;
            s_next = (seg_kofs + nj * seg_Cw + nc0) >> 4;      // first k-step of the next stage in the packed weight rows
        } else {
            new_chunk = false;
        }
        // (sched_barrier: with every load unconditional a stage is one basic block, and the scheduler would sink the loads to
        //  their first use -- the fences keep them where the latency plan needs them)
        load_a(nsi, nc0, nj, new_chunk);
        if (!LEAN) load_b(s_next, sp);
        __builtin_amdgcn_sched_barrier(0);
        const __bf16* Ac = As0 + abuf * NPL * PLANE + a_frag + cur_j * T6_PITCH;
#ifdef T6_TRACE   // developer build: s_memtime stamps of one workgroup's stages (tools/experiments/r3o_trace.py)
        if (t6_on && t6_stage < 16) p.clk[16 + (wave * 16 + t6_stage) * 8 + 7] = (seg_interior ? 1 : 0) | (new_chunk ? 2 : 0) | (mt << 8);
        T6_STAMP(0);
#endif
        if constexpr (LEAN) {
            bf16x8 af[3][WMT];
            read_a(Ac, 0, af);
            mfma_step(af, u0);
#ifdef T6_TRACE
            T6_STAMP(1);
#endif
            __builtin_amdgcn_sched_barrier(0);
            load_b(s_next, u0);
            __builtin_amdgcn_sched_barrier(0);
            read_a(Ac, 1, af);
            mfma_step(af, u1);
#ifdef T6_TRACE
            T6_STAMP(2);
#endif
            __builtin_amdgcn_sched_barrier(0);
            load_b(s_next + 1, u1);
            __builtin_amdgcn_sched_barrier(0);
        } else {
            bf16x8 af0[3][WMT], af1[3][WMT];
            read_a(Ac, 0, af0);
            read_a(Ac, 1, af1);                                // the second k-step's fragments travel under the first one's MFMAs
            mfma_step(af0, u0);
            __builtin_amdgcn_sched_barrier(0);
            load_b(s_next + 1, u0);
            __builtin_amdgcn_sched_barrier(0);
            mfma_step(af1, u1);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!has_next) return true;
#ifdef T6_TRACE
        T6_STAMP(3);
#endif
        // One barrier per CHUNK, not per stage: a slab is written only here, after the barrier that ended the chunk which last
        // read it, and read only after the barrier that follows these writes -- the taps of one chunk need no barrier between them.
        if (new_chunk) {                                       // the A loads had the whole stage to arrive
            abuf ^= 1;
            store_a(As0 + abuf * NPL * PLANE);
        }
#ifdef T6_TRACE
        T6_STAMP(4);
#endif
        if (new_chunk) __syncthreads();
#ifdef T6_TRACE
        T6_STAMP(5);
        ++t6_stage;
#endif
        si = nsi; c0 = nc0; j = nj;
        return false;
    };
#ifdef T6_TRACE
    T6_PHASE(6);
#endif
    if constexpr (LEAN) {
        while (!stage(bx, by, bz)) {}
    } else {
        for (;;) {
            if (stage(bx, by, bz)) break;
            if (stage(bz, bx, by)) break;
            if (stage(by, bz, bx)) break;
        }
    }

    tl.b = b; tl.m0 = m0; tl.n0 = n0; tl.a_inv = a_inv; tl.rowmode = rowmode; tl.clk_t0 = clk_t0; tl.clk_r0 = clk_r0;
#ifdef T6_TRACE
    tl.t6_ph = t6_ph;
#else
    tl.t6_ph = false;
#endif
}

template <int WGM, int WGN, int WMT, int WN, int NP = 2, int HALO = 7>
__global__ __launch_bounds__(64 * WGM * WGN, (tap6_occupancy<WGM, WGN, WMT, WN, NP>())) void tap_gemm6_kernel(const TapGemmParams p, const __bf16* __restrict__ wp) {
    static_assert(NP == 2, "split16 arithmetic only");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x16 acc[WMT][WN];
    Tap6Tile tl;
    tap6_mainloop<WGM, WGN, WMT, WN, HALO, false>(p, wp, smem, acc, tl);
    tap6_epilogue<WGM, WGN, WMT, WN, HALO>(p, acc, smem, tl.b, tl.m0, tl.n0, tl.a_inv, tl.rowmode, tl.clk_t0, tl.clk_r0, tl.t6_ph);
}

}  // namespace ac
