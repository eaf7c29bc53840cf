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
#pragma unroll
        for (int sh = 1; sh < 64; sh <<= 1) {
            const unsigned t = (unsigned)__shfl_xor((int)rmax, sh);
            rmax = t > rmax ? t : rmax;
        }
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
