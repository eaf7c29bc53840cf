// Kernels only Mimi needs ([HF] = transformers models/mimi/modeling_mimi.py, the third-party model behind
// /root/reference/audiocodecs/mimi.py):
//   layernorm_kernel   nn.LayerNorm over the hidden axis                          ([HF]:737-738, 771-778)
//   attention_kernel   RoPE + causal sliding-window softmax(QK^T / sqrt(d)) V     ([HF]:582-599, 640-654, 689-727)
//   upsample_dw_kernel depthwise ConvTranspose1d, k = 2*stride, right-trimmed     ([HF]:1209-1217, 388-400)
// The linear layers (q/k/v/o projections, MLP) run through the tap-GEMM as 1-tap convolutions over the
// [B*T][hidden] token matrix with GELU / LayerScale / residual folded into its epilogue (tap_gemm.h).
#pragma once
#include <hip/hip_runtime.h>
#include "tap_gemm.h"
#include "split16.h"

namespace ac {

// ---------------------------------------------------------------------------------------------
// LayerNorm: one wavefront per row; two-pass moments in fp32 (biased variance), wave butterfly reductions.
// Algorithmic bytes: 8*H per row (+ 8*H of parameters once).
// ---------------------------------------------------------------------------------------------
struct LayerNormParams {
    const float* x;      // [rows][H]
    const float* w;      // [H]
    const float* b;      // [H]
    float* y;            // [rows][H]
    long long rows;
    int H;
    float eps;
    unsigned* rowmax;    // optional [rows]: bit pattern of the largest finite |y| of each row (split16.h row mode: the linear layer
                         // that reads y scales by it -- saves that layer's rowmax_kernel launch and its second read of y)
};

constexpr int LN_MAXV = 16;   // H <= 1024

__global__ __launch_bounds__(256) void layernorm_kernel(const LayerNormParams p) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.rows) return;
    const float* xr = p.x + row * p.H;
    float v[LN_MAXV];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < p.H ? xr[c] : 0.f;
        sum += v[i];
    }
#pragma unroll
    for (int sh = 1; sh < 64; sh <<= 1) sum += __shfl_xor(sum, sh);
    const float mean = sum / (float)p.H;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = lane + 64 * i;
        const float d = c < p.H ? v[i] - mean : 0.f;
        sq = fmaf(d, d, sq);
    }
#pragma unroll
    for (int sh = 1; sh < 64; sh <<= 1) sq += __shfl_xor(sq, sh);
    const float rstd = 1.0f / sqrtf(sq / (float)p.H + p.eps);
    float* yr = p.y + row * p.H;
    unsigned rmax = 0;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < p.H) {
            const float o = (v[i] - mean) * rstd * p.w[c] + p.b[c];
            yr[c] = o;
            amax_acc(rmax, o);
        }
    }
    if (p.rowmax) {
        rmax = group_max_u32<64>(rmax);
        if (lane == 0) p.rowmax[row] = rmax;
    }
}

// ---------------------------------------------------------------------------------------------
// Attention.  Grid (ceil(T/64), heads, clips); 4 waves, each owns 16 queries of the 64-query tile.
// Keys are walked in 64-key tiles from max(0, q0 - window + 1) to the diagonal; per tile
//   S = Q K^T (v_mfma_f32_16x16x4_f32, RoPE applied to Q and K while they are staged into LDS),
//   key j is visible to query i iff j <= i and i - j < window ([HF] create_sliding_window_causal_mask),
//   online softmax in fp32 (running max / sum per query row), O += P V on the MFMA.
// The [T][T] score matrix never exists in HBM.  Algorithmic bytes per (clip, head): 16*T*HD (q,k,v in, o out).
// ---------------------------------------------------------------------------------------------
struct AttnParams {
    const float* qkv;    // [B][T][3*A]: q | k | v, A = heads*HD, head h at columns h*HD
    float* out;          // [B][T][A]
    const float* cos;    // [>=T][HD]  (emb = cat(freqs, freqs): [HF]:562-565)
    const float* sin;
    int B, T, A, window;
    float scaling;       // 1/sqrt(HD)
};

template <int HD>
struct AttnCfg {
    static constexpr int QT = 64, KT = 64, HP = HD + 4, PP = KT + 4;
    static constexpr size_t lds_bytes = (size_t)(QT * HP + 2 * KT * HP + 4 * 16 * PP) * 4;
};

template <int HD>
__global__ __launch_bounds__(256) void attention_kernel(const AttnParams p) {
    using Cfg = AttnCfg<HD>;
    constexpr int QT = Cfg::QT, KT = Cfg::KT, HP = Cfg::HP, PP = Cfg::PP, HV = HD / 16, H4 = HD / 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Qs = smem;                     // [QT][HP]
    float* Ks = Qs + QT * HP;             // [KT][HP]
    float* Vs = Ks + KT * HP;             // [KT][HP]
    float* Ps = Vs + KT * HP;             // [4 waves][16][PP]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int q0 = blockIdx.x * QT, h = blockIdx.y, b = blockIdx.z;
    const long long rs = 3LL * p.A;
    const float* base = p.qkv + (long long)b * p.T * rs + (long long)h * HD;

    // rows t0.. of q (which = 0) or k (which = 1) -> dst with RoPE: x*cos + rotate_half(x)*sin  ([HF]:582-599)
    auto stage_rope = [&](float* dst, int t0, int which) {
        for (int e = tid; e < 64 * H4; e += 256) {
            const int row = e / H4, d = (e % H4) * 4;
            const int t = t0 + row;
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (t < p.T) {
                const float* src = base + (long long)t * rs + which * p.A;
                const f32x4 x = *reinterpret_cast<const f32x4*>(src + d);
                const bool lo = d < HD / 2;
                f32x4 xp = *reinterpret_cast<const f32x4*>(src + (lo ? d + HD / 2 : d - HD / 2));
                if (lo) xp = -xp;
                const f32x4 c = *reinterpret_cast<const f32x4*>(p.cos + (long long)t * HD + d);
                const f32x4 s = *reinterpret_cast<const f32x4*>(p.sin + (long long)t * HD + d);
                v.x = __fadd_rn(__fmul_rn(x.x, c.x), __fmul_rn(xp.x, s.x));
                v.y = __fadd_rn(__fmul_rn(x.y, c.y), __fmul_rn(xp.y, s.y));
                v.z = __fadd_rn(__fmul_rn(x.z, c.z), __fmul_rn(xp.z, s.z));
                v.w = __fadd_rn(__fmul_rn(x.w, c.w), __fmul_rn(xp.w, s.w));
            }
            *reinterpret_cast<f32x4*>(&dst[row * HP + d]) = v;
        }
    };

    stage_rope(Qs, q0, 0);
    __syncthreads();
    f32x4 qf[HV];
#pragma unroll
    for (int ks = 0; ks < HV; ++ks) qf[ks] = *reinterpret_cast<const f32x4*>(&Qs[(wave * 16 + li) * HP + ks * 16 + 4 * kq]);

    float m_run[4], l_run[4];
    f32x4 o[HV];
#pragma unroll
    for (int r = 0; r < 4; ++r) { m_run[r] = -INFINITY; l_run[r] = 0.f; }
#pragma unroll
    for (int c = 0; c < HV; ++c) o[c] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int k_lo = max(0, q0 - p.window + 1) / KT * KT;
    const int k_hi = min(p.T, q0 + QT);
    float* Pw = Ps + wave * 16 * PP;
    for (int kb = k_lo; kb < k_hi; kb += KT) {
        __syncthreads();                                   // previous tile's K/V/P fully consumed
        stage_rope(Ks, kb, 1);
        for (int e = tid; e < KT * H4; e += 256) {
            const int row = e / H4, d = (e % H4) * 4;
            f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
            if (kb + row < p.T) v = *reinterpret_cast<const f32x4*>(base + (long long)(kb + row) * rs + 2 * p.A + d);
            *reinterpret_cast<f32x4*>(&Vs[row * HP + d]) = v;
        }
        __syncthreads();
        // ---- scores for this wave's 16 queries x 64 keys
        f32x4 s[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            s[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < HV; ++ks) {
                const f32x4 kf = *reinterpret_cast<const f32x4*>(&Ks[(c * 16 + li) * HP + ks * 16 + 4 * kq]);
#pragma unroll
                for (int u = 0; u < 4; ++u) s[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[ks][u], kf[u], s[c], 0, 0, 0);
            }
        }
        // ---- mask + online softmax; C layout: row (query) = kq*4 + r, column (key) = c*16 + li
        float alpha[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = q0 + wave * 16 + kq * 4 + r;
            float mx = -INFINITY;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int j = kb + c * 16 + li;
                const bool vis = j <= i && i - j < p.window && j < p.T;
                const float sv = vis ? s[c][r] * p.scaling : -INFINITY;
                s[c][r] = sv;
                mx = fmaxf(mx, sv);
            }
#pragma unroll
            for (int sh = 1; sh < 16; sh <<= 1) mx = fmaxf(mx, __shfl_xor(mx, sh));
            const float m_new = fmaxf(m_run[r], mx);
            float sum = 0.f;
            if (m_new == -INFINITY) {                       // nothing visible yet for this query
                alpha[r] = 1.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) s[c][r] = 0.f;
            } else {
                alpha[r] = expf(m_run[r] - m_new);          // exp(-inf) = 0 on the first visible tile
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float pv = expf(s[c][r] - m_new);
                    s[c][r] = pv;
                    sum += pv;
                }
            }
#pragma unroll
            for (int sh = 1; sh < 16; sh <<= 1) sum += __shfl_xor(sum, sh);
            l_run[r] = l_run[r] * alpha[r] + sum;
            m_run[r] = m_new;
#pragma unroll
            for (int c = 0; c < 4; ++c) Pw[(kq * 4 + r) * PP + c * 16 + li] = s[c][r];
        }
#pragma unroll
        for (int c = 0; c < HV; ++c) { o[c][0] *= alpha[0]; o[c][1] *= alpha[1]; o[c][2] *= alpha[2]; o[c][3] *= alpha[3]; }
        __syncthreads();                                   // P visible to the lanes that read it as an A operand
        // ---- O += P V
#pragma unroll
        for (int ks = 0; ks < KT / 16; ++ks) {
            const f32x4 pf = *reinterpret_cast<const f32x4*>(&Pw[li * PP + ks * 16 + 4 * kq]);
#pragma unroll
            for (int c = 0; c < HV; ++c)
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    o[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(pf[u], Vs[(ks * 16 + 4 * kq + u) * HP + c * 16 + li], o[c], 0, 0, 0);
        }
    }
    // ---- normalise and store: row = kq*4 + r, column (dim) = c*16 + li
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = q0 + wave * 16 + kq * 4 + r;
        if (i < p.T) {
            float* dst = p.out + ((long long)b * p.T + i) * p.A + (long long)h * HD;
#pragma unroll
            for (int c = 0; c < HV; ++c) dst[c * 16 + li] = o[c][r] / l_run[r];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// attention16_kernel (round 4, head_dim 64): the same attention -- same grid, same key range, same mask, same online softmax --
// with both products in split16.h arithmetic on v_mfma_f32_16x16x32_f16 (3 partial products per fp32 product, fp32 accumulate) and
// NO score / probability tile in LDS.  attention_kernel spent 357 us per launch (Mimi 128 x 10 s, 16 launches per step) on chains of
// 16 dependent fp32 MFMAs, one LDS word per PV MFMA, libm expf, and a global -> LDS -> barrier sequence per key tile with
// nothing in flight behind it.  Here
//   * both products are computed TRANSPOSED, the key-side matrix as the MFMA's A operand and the query-side one as B:
//       S^T[key][query] = K Q^T,   O^T[dim][query] = V^T P^T
//     so a lane owns ONE query (column li) in every accumulator: running maximum, sum, rescaling and the final division are
//     lane-local (two shuffles across the four kq groups), and the S^T accumulators of two 16-key tiles ARE the B operand of
//     a 32-key step of the second product -- the k order of a dot product is free as long as both operands use the same one
//     (rvq16.h), so V^T is read from LDS in THAT order: keys 16 c0 + 4 kq + 0..3, then 16 c1 + 4 kq + 0..3;
//   * Q is loaded straight into its B fragments (a lane's rotate_half partner is its own other k-step: dims d and d + 32);
//   * K (after RoPE) and V^T of a 64-key tile are staged as fp16 hi / lo planes under one power-of-two scale each -- the tile's
//     largest magnitude, found by a workgroup reduction (LDS atomic, three rotating words) -- and the NEXT tile's global loads
//     are in flight while the current one is multiplied;
//   * P in [0, 1] is split under the fixed scale 2^14; a tile's contribution is accumulated in its own scaled units and added
//     to O^T with the exact factor 2^-14 / scale_V, so tiles with different V scales combine.
// The error of a product is that of an fp32 FMA chain (split16.h); exp is v_exp_f32 after one multiply by log2(e) (ELU's
// choice, tap_gemm.h).  Parity: tests/test_mimi_gpu_parity.py (tokens exact outside fp64 near-ties, features / waveform within tolerance).
// ---------------------------------------------------------------------------------------------
struct Attn16Cfg {
    static constexpr int QT = 64, KT = 64, HD = 64, KP = 72;            // KP: fp16 per LDS row (64 + 8: 144-byte rows)
    static constexpr int PLANE = 64 * KP;
    static constexpr size_t lds_bytes = (size_t)4 * PLANE * 2 + 32;     // K hi / lo [key][dim], V^T hi / lo [dim][key], 6 amax words
};

// 8 fp32 (two quads) x scale -> the hi / lo fp16 fragments of one MFMA operand (split16.h: scaled value = hi + lo)
__device__ __forceinline__ void split16_pack8(const f32x4 a, const f32x4 b, const float sc, f16x8& hi, f16x8& lo) {
    const f32x4 s0 = a * sc, s1 = b * sc;                                // exact: sc is a power of two
    const f16x4_t h0 = __builtin_convertvector(s0, f16x4_t), h1 = __builtin_convertvector(s1, f16x4_t);
    const f16x4_t l0 = __builtin_convertvector(s0 - __builtin_convertvector(h0, f32x4), f16x4_t);
    const f16x4_t l1 = __builtin_convertvector(s1 - __builtin_convertvector(h1, f32x4), f16x4_t);
    hi = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
    lo = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
}

__global__ __launch_bounds__(256, 2) void attention16_kernel(const AttnParams p) {
    using Cfg = Attn16Cfg;
    constexpr int KT = Cfg::KT, HD = Cfg::HD, KP = Cfg::KP, PLANE = Cfg::PLANE;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    _Float16* Kp = reinterpret_cast<_Float16*>(smem);         // [2 planes][64 keys][KP]
    _Float16* Vp = Kp + 2 * PLANE;                             // [2 planes][64 dims][KP]  (keys along the row)
    unsigned* slot = reinterpret_cast<unsigned*>(Vp + 2 * PLANE);   // [3 rotating][K, V]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int q0 = blockIdx.x * Cfg::QT, h = blockIdx.y, b = blockIdx.z;
    const long long rs = 3LL * p.A;
    const float* base = p.qkv + (long long)b * p.T * rs + (long long)h * HD;
    if (tid < 6) slot[tid] = 0u;

    // ---- Q^T fragments: lane (query li, kq) holds dims 8 kq .. + 7 (k-step 0) and 32 + 8 kq .. (k-step 1), RoPE in registers
    const int qi = q0 + wave * 16 + li;
    f16x8 qh[2], ql[2];
    float iq;
    {
        f32x4 xa[2], xb[2], cv[2], sv[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            xa[u] = xb[u] = cv[u] = sv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (qi < p.T) {
                const float* src = base + (long long)qi * rs + 8 * kq + 4 * u;
                xa[u] = *reinterpret_cast<const f32x4*>(src);
                xb[u] = *reinterpret_cast<const f32x4*>(src + HD / 2);
                cv[u] = *reinterpret_cast<const f32x4*>(p.cos + (long long)qi * HD + 8 * kq + 4 * u);
                sv[u] = *reinterpret_cast<const f32x4*>(p.sin + (long long)qi * HD + 8 * kq + 4 * u);
            }
        }
        f32x4 ra[2], rb[2];                                    // x cos + rotate_half(x) sin  ([HF]:582-599), unfused like the reference
        unsigned am = 0;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                ra[u][e] = __fadd_rn(__fmul_rn(xa[u][e], cv[u][e]), __fmul_rn(-xb[u][e], sv[u][e]));
                rb[u][e] = __fadd_rn(__fmul_rn(xb[u][e], cv[u][e]), __fmul_rn(xa[u][e], sv[u][e]));
                amax_acc(am, ra[u][e]);
                amax_acc(am, rb[u][e]);
            }
        am = group_max_u32<64>(am);
        const int eq = s16_exponent(am);
        iq = s16_pow2(-eq);
        split16_pack8(ra[0], ra[1], s16_pow2(eq), qh[0], ql[0]);
        split16_pack8(rb[0], rb[1], s16_pow2(eq), qh[1], ql[1]);
    }

    // ---- staging registers of one key tile: K as (dims 4 g .., 32 + 4 g ..) pairs of two rows per thread, V as a 4-key x 4-dim block
    f32x4 kx0[2], kx1[2], kc[2], ksn[2], vv[4];
    const int kg = tid >> 4, dg = tid & 15;
    auto issue_loads = [&](int kb) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e2 = tid + 256 * i, row = e2 >> 3, g = e2 & 7;
            const int t = kb + row;
            kx0[i] = kx1[i] = kc[i] = ksn[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (t < p.T) {
                const float* src = base + (long long)t * rs + p.A + 4 * g;
                kx0[i] = *reinterpret_cast<const f32x4*>(src);
                kx1[i] = *reinterpret_cast<const f32x4*>(src + HD / 2);
                kc[i] = *reinterpret_cast<const f32x4*>(p.cos + (long long)t * HD + 4 * g);
                ksn[i] = *reinterpret_cast<const f32x4*>(p.sin + (long long)t * HD + 4 * g);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int t = kb + 4 * kg + r;
            vv[r] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (t < p.T) vv[r] = *reinterpret_cast<const f32x4*>(base + (long long)t * rs + 2 * p.A + 4 * dg);
        }
    };

    float m_run = -INFINITY, l_run = 0.f;
    f32x4 o[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) o[d] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int k_lo = max(0, q0 - p.window + 1) / KT * KT;
    const int k_hi = min(p.T, q0 + Cfg::QT);
    issue_loads(k_lo);
    __syncthreads();                                           // the amax words are zero
    int jt = 0;
    for (int kb = k_lo; kb < k_hi; kb += KT, ++jt) {
        // ---- RoPE on K, the tile's two maxima
        f32x4 klo[2], khi[2];
        unsigned amk = 0, amv = 0;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                klo[i][e] = __fadd_rn(__fmul_rn(kx0[i][e], kc[i][e]), __fmul_rn(-kx1[i][e], ksn[i][e]));
                khi[i][e] = __fadd_rn(__fmul_rn(kx1[i][e], kc[i][e]), __fmul_rn(kx0[i][e], ksn[i][e]));
                amax_acc(amk, klo[i][e]);
                amax_acc(amk, khi[i][e]);
            }
#pragma unroll
        for (int r = 0; r < 4; ++r) amax_acc4(amv, vv[r]);
        amk = group_max_u32<64>(amk);
        amv = group_max_u32<64>(amv);
        unsigned* sl = slot + (jt % 3) * 2;
        if (lane == 0) { atomicMax(sl, amk); atomicMax(sl + 1, amv); }
        __syncthreads();                                       // A: maxima complete; every wave is done with the previous tile's planes
        if (tid < 2) slot[((jt + 1) % 3) * 2 + tid] = 0u;      // (next tile's words: last read before the previous tile's barrier B)
        const int ek = s16_exponent(sl[0]), ev = s16_exponent(sl[1]);
        const float sk = s16_pow2(ek), svs = s16_pow2(ev);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e2 = tid + 256 * i, row = e2 >> 3, g = e2 & 7;
            split16_store4s(klo[i], sk, Kp, PLANE, row * KP + 4 * g);
            split16_store4s(khi[i], sk, Kp, PLANE, row * KP + HD / 2 + 4 * g);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)                            // V^T: four consecutive keys of one dim per 8-byte store
            split16_store4s(f32x4{vv[0][u], vv[1][u], vv[2][u], vv[3][u]}, svs, Vp, PLANE, (4 * dg + u) * KP + 4 * kg);
        if (kb + KT < k_hi) issue_loads(kb + KT);              // in flight during this tile's products
        __syncthreads();                                       // B: planes visible

        // ---- S^T = K Q^T: accumulator c, register r <-> key kb + 16 c + 4 kq + r, this lane's query
        f32x4 s[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            s[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const f16x8 kh = *reinterpret_cast<const f16x8*>(Kp + (c * 16 + li) * KP + 32 * ks + 8 * kq);
                const f16x8 kl = *reinterpret_cast<const f16x8*>(Kp + PLANE + (c * 16 + li) * KP + 32 * ks + 8 * kq);
                s[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kl, qh[ks], s[c], 0, 0, 0);
                s[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, ql[ks], s[c], 0, 0, 0);
                s[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh, qh[ks], s[c], 0, 0, 0);
            }
        }
        // ---- mask, online softmax (lane-local but for the four kq groups of the query)
        const float ds = iq * s16_pow2(-ek) * p.scaling;
        float mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = kb + c * 16 + kq * 4 + r;
                const bool vis = j <= qi && qi - j < p.window && j < p.T;
                const float v = vis ? s[c][r] * ds : -INFINITY;
                s[c][r] = v;
                mx = fmaxf(mx, v);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        float alpha = 1.f, sum = 0.f;
        if (m_new == -INFINITY) {                              // nothing visible yet for this query
#pragma unroll
            for (int c = 0; c < 4; ++c) s[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        } else {
            alpha = __expf(m_run - m_new);                     // exp(-inf) = 0 on the first visible tile
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = __expf(s[c][r] - m_new);
                    s[c][r] = pv;
                    sum += pv;
                }
        }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        l_run = l_run * alpha + sum;
        m_run = m_new;
        // ---- O^T += V^T P^T, the tile's contribution in its own units first
        f32x4 tacc[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) tacc[d] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            f16x8 ph, pl;
            split16_pack8(s[2 * kk], s[2 * kk + 1], 16384.0f, ph, pl);
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const _Float16* vr = Vp + (d * 16 + li) * KP + 32 * kk + 4 * kq;
                const f16x4_t vh0 = *reinterpret_cast<const f16x4_t*>(vr), vh1 = *reinterpret_cast<const f16x4_t*>(vr + 16);
                const f16x4_t vl0 = *reinterpret_cast<const f16x4_t*>(vr + PLANE), vl1 = *reinterpret_cast<const f16x4_t*>(vr + PLANE + 16);
                const f16x8 vh = __builtin_shufflevector(vh0, vh1, 0, 1, 2, 3, 4, 5, 6, 7);
                const f16x8 vl = __builtin_shufflevector(vl0, vl1, 0, 1, 2, 3, 4, 5, 6, 7);
                tacc[d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vl, ph, tacc[d], 0, 0, 0);
                tacc[d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, pl, tacc[d], 0, 0, 0);
                tacc[d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vh, ph, tacc[d], 0, 0, 0);
            }
        }
        const float dv = s16_pow2(-ev) * (1.0f / 16384.0f);
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[d][r] = __fmaf_rn(tacc[d][r], dv, o[d][r] * alpha);
    }
    // ---- normalise and store: this lane's query, dims 16 d + 4 kq .. + 3
    if (qi < p.T) {
        float* dst = p.out + ((long long)b * p.T + qi) * p.A + (long long)h * HD + 4 * kq;
#pragma unroll
        for (int d = 0; d < 4; ++d)
            *reinterpret_cast<f32x4*>(dst + d * 16) = f32x4{o[d][0] / l_run, o[d][1] / l_run, o[d][2] / l_run, o[d][3] / l_run};
    }
}

// ---------------------------------------------------------------------------------------------
// Depthwise transposed conv (groups == channels), kernel 2*s, stride s, k - s samples trimmed on the right:
//   y[i*s + ph][c] = x[i][c] * w[c][ph] + x[i-1][c] * w[c][ph + s]          (x[-1] = 0)
// HBM-bound: 4*C*(1 + s) bytes per input frame.  One thread per (output step, 4 channels).
// ---------------------------------------------------------------------------------------------
struct UpsampleParams {
    const float* x;     // [B][N][C]
    const float* w;     // [C][2*s]  (HF layout [C][1][k])
    float* y;           // [B][N*s][C]
    int B, N, C, s;
};

__global__ __launch_bounds__(256) void upsample_dw_kernel(const UpsampleParams p) {
    const int c4 = p.C / 4;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)p.B * p.N * p.s * c4;
    if (gid >= total) return;
    const int q = (int)(gid % c4);
    const long long row = gid / c4;                         // b*(N*s) + t
    const int t = (int)(row % ((long long)p.N * p.s));
    const long long b = row / ((long long)p.N * p.s);
    const int i = t / p.s, ph = t % p.s, k = 2 * p.s;
    const float* xb = p.x + b * p.N * p.C;
    const f32x4 x1 = *reinterpret_cast<const f32x4*>(xb + (long long)i * p.C + 4 * q);
    f32x4 x0 = f32x4{0.f, 0.f, 0.f, 0.f};
    if (i > 0) x0 = *reinterpret_cast<const f32x4*>(xb + (long long)(i - 1) * p.C + 4 * q);
    f32x4 y;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const float* wc = p.w + (long long)(4 * q + u) * k;
        y[u] = __fadd_rn(__fmul_rn(x0[u], wc[ph + p.s]), __fmul_rn(x1[u], wc[ph]));
    }
    *reinterpret_cast<f32x4*>(p.y + row * p.C + 4 * q) = y;
}

}  // namespace ac
