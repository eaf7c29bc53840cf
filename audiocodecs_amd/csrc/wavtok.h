// Kernels only the WavTokenizer decoder needs (oracle/wavtokenizer_oracle.py restates the
// modules and says where each comes from -- the backend package is not on disk: PARITY UNPINNED):
//   gn_stats_kernel / gn_apply_kernel   GroupNorm(32, C, eps 1e-6) over (channels of a group x all frames), optional swish
//   attn1_kernel                         single-head softmax(Q K^T / sqrt(C)) V over all frames of a clip (AttnBlock)
//   dwconv_ln_kernel                     depthwise Conv1d(k7, zero pad 3) + layer_norm * scale + shift (ConvNeXt front half)
//   polar_kernel                         (log-magnitude, phase) pairs -> (re, im) = min(exp(m), 100) * (cos p, sin p)
//   istft_env_kernel                     division by the overlap-added squared window (ISTFT padding="same")
// Every dense layer (embed conv, k3 convs, q/k/v/proj, pwconv1/2, head Linear, inverse-rFFT + overlap-add) runs through
// the tap-GEMM; layouts are channels-last [B][N][C] fp32 like everywhere else.
#pragma once
#include <hip/hip_runtime.h>
#include "tap_gemm.h"

namespace ac {

// ---------------------------------------------------------------------------------------------
// GroupNorm statistics: workgroup = (clip, GN_GPW consecutive groups) -- 512 workgroups at 64 clips x 32 groups.  It reads
// its [N][GN_GPW * C/G] column block with 16-byte loads (thread = (row lane, 4 channels)), two passes (mean, then centred
// second moment; the second pass hits L2) in fp32 -- biased variance like torch.  stats[b][g] = (mean, rstd).
// Algorithmic bytes: 4*N*C per clip.
// ---------------------------------------------------------------------------------------------
struct GnStatsParams {
    const float* x;      // [B][N][C]
    float* stats;        // [B][G][2]
    int B, N, C, G;
    float eps;
};

constexpr int GN_GPW = 4;    // groups per workgroup; needs (C/G * GN_GPW) % 4 == 0 and C/G * GN_GPW <= 1024

__global__ __launch_bounds__(256) void gn_stats_kernel(const GnStatsParams p) {
    __shared__ float red[256 * 4];
    __shared__ float gmean[GN_GPW];
    const int b = blockIdx.y, g0 = blockIdx.x * GN_GPW, tid = threadIdx.x;
    const int cpg = p.C / p.G, width = cpg * GN_GPW, cols4 = width / 4;
    const int rl = 256 / cols4, rlane = tid / cols4, c4 = (tid % cols4) * 4;
    const bool active = rlane < rl;
    const float* xb = p.x + (long long)b * p.N * p.C + (long long)g0 * cpg + c4;
    const float cnt = (float)p.N * (float)cpg;
    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
    if (active)
        for (int t = rlane; t < p.N; t += rl) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(xb + (long long)t * p.C);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    *reinterpret_cast<f32x4*>(&red[tid * 4]) = s;
    __syncthreads();
    if (tid < GN_GPW) {          // fixed summation order: row lanes outer, the group's channels inner
        float m = 0.f;
        for (int r = 0; r < rl; ++r)
            for (int c = 0; c < cpg; ++c) m += red[(r * cols4) * 4 + tid * cpg + c];
        gmean[tid] = m / cnt;
    }
    __syncthreads();
    s = f32x4{0.f, 0.f, 0.f, 0.f};
    if (active) {
        const f32x4 mu = f32x4{gmean[c4 / cpg], gmean[(c4 + 1) / cpg], gmean[(c4 + 2) / cpg], gmean[(c4 + 3) / cpg]};
        for (int t = rlane; t < p.N; t += rl) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(xb + (long long)t * p.C);
            const f32x4 d = f32x4{v.x - mu.x, v.y - mu.y, v.z - mu.z, v.w - mu.w};
            s.x = fmaf(d.x, d.x, s.x); s.y = fmaf(d.y, d.y, s.y); s.z = fmaf(d.z, d.z, s.z); s.w = fmaf(d.w, d.w, s.w);
        }
    }
    __syncthreads();
    *reinterpret_cast<f32x4*>(&red[tid * 4]) = s;
    __syncthreads();
    if (tid < GN_GPW) {
        float v = 0.f;
        for (int r = 0; r < rl; ++r)
            for (int c = 0; c < cpg; ++c) v += red[(r * cols4) * 4 + tid * cpg + c];
        float* o = p.stats + ((long long)b * p.G + g0 + tid) * 2;
        o[0] = gmean[tid];
        o[1] = 1.0f / sqrtf(v / cnt + p.eps);
    }
}

struct GnApplyParams {
    const float* x;      // [B][N][C]
    const float* stats;  // [B][G][2]
    const float* w;      // [C]
    const float* b;      // [C]
    float* y;            // [B][N][C]
    int B, N, C, G;
    int swish;           // 1: y = v * sigmoid(v)
};

__global__ __launch_bounds__(256) void gn_apply_kernel(const GnApplyParams p) {
    const int c4 = p.C / 4;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)p.B * p.N * c4;
    if (gid >= total) return;
    const int q = (int)(gid % c4);
    const long long row = gid / c4;
    const int b = (int)(row / p.N);
    const int cpg = p.C / p.G;
    const f32x4 v = *reinterpret_cast<const f32x4*>(p.x + row * p.C + 4 * q);
    f32x4 o;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int c = 4 * q + u;
        const float* st = p.stats + ((long long)b * p.G + c / cpg) * 2;
        float r = (v[u] - st[0]) * st[1] * p.w[c] + p.b[c];
        if (p.swish) r = r * (1.0f / (1.0f + expf(-r)));
        o[u] = r;
    }
    *reinterpret_cast<f32x4*>(p.y + row * p.C + 4 * q) = o;
}

// ---------------------------------------------------------------------------------------------
// Single-head attention over all N frames of a clip (AttnBlock of pos_net): out = softmax(q k^T * scale) v.
// Workgroup = 16 queries of one clip, 4 waves; wave w owns the feature quarter [w*DW, (w+1)*DW) -- of the contraction in
// q k^T (partial scores meet in LDS) and of the output columns in P v.  Keys are walked in 64-key tiles with an online
// softmax (fp32, running max / sum per query; every wave keeps the same copy).  K and V fragments come straight from
// L2 as 16-byte loads: the k order inside an MFMA chain is a fixed permutation, and in P v one 16-byte load of V feeds 4
// output tiles (tile e, column li <-> feature 64*cg + 4*li + e).  v_mfma_f32_16x16x4_f32 (exact fp32 products).
// The [N][N] score matrix never exists in HBM.  Algorithmic bytes per clip: 16*N*C.
// ---------------------------------------------------------------------------------------------
struct Attn1Params {
    const float* qkv;    // [B][N][3*C]: q | k | v
    float* out;          // [B][N][C]
    int B, N, C;
    float scaling;       // C^-0.5
};

template <int DW>   // C / 4
__global__ __launch_bounds__(256) void attn1_kernel(const Attn1Params p) {
    constexpr int KSQ = DW / 16, CG = DW / 64, KT = 64, SP = KT + 4;
    __shared__ __attribute__((aligned(16))) float Sp[4][16][SP];     // partial scores per wave
    __shared__ __attribute__((aligned(16))) float Pw[4][16][SP];     // probabilities, one private copy per wave
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int q0 = blockIdx.x * 16, b = blockIdx.y;
    const long long rs = 3LL * p.C;
    const float* base = p.qkv + (long long)b * p.N * rs;
    const int d0 = wave * DW;

    f32x4 qf[KSQ];
    {
        const int qi = min(q0 + li, p.N - 1);
#pragma unroll
        for (int ks = 0; ks < KSQ; ++ks) qf[ks] = *reinterpret_cast<const f32x4*>(base + (long long)qi * rs + d0 + ks * 16 + 4 * kq);
    }
    float m_run[4], l_run[4];
    f32x4 o[CG][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { m_run[r] = -INFINITY; l_run[r] = 0.f; }
#pragma unroll
    for (int g = 0; g < CG; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) o[g][e] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int kb = 0; kb < p.N; kb += KT) {
        // ---- partial scores of this wave's feature quarter: 16 queries x 64 keys
        f32x4 s[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            s[c] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int kj = min(kb + c * 16 + li, p.N - 1);
            const float* kr = base + (long long)kj * rs + p.C + d0 + 4 * kq;
#pragma unroll
            for (int ks = 0; ks < KSQ; ++ks) {
                const f32x4 kf = *reinterpret_cast<const f32x4*>(kr + ks * 16);
#pragma unroll
                for (int u = 0; u < 4; ++u) s[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[ks][u], kf[u], s[c], 0, 0, 0);
            }
        }
        __syncthreads();                                   // previous tile's Sp fully consumed
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) Sp[wave][kq * 4 + r][c * 16 + li] = s[c][r];
        __syncthreads();
        // ---- full scores (quarters summed in a fixed order), mask, online softmax.  C layout: row = kq*4 + r, col = c*16 + li
        float alpha[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float mx = -INFINITY;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int row = kq * 4 + r, col = c * 16 + li;
                const float sv = ((Sp[0][row][col] + Sp[1][row][col]) + (Sp[2][row][col] + Sp[3][row][col])) * p.scaling;
                s[c][r] = kb + col < p.N ? sv : -INFINITY;
                mx = fmaxf(mx, s[c][r]);
            }
#pragma unroll
            for (int sh = 1; sh < 16; sh <<= 1) mx = fmaxf(mx, __shfl_xor(mx, sh));
            const float m_new = fmaxf(m_run[r], mx);       // finite: every tile has at least one key < N
            alpha[r] = expf(m_run[r] - m_new);
            float sum = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float pv = expf(s[c][r] - m_new);
                s[c][r] = pv;
                sum += pv;
            }
#pragma unroll
            for (int sh = 1; sh < 16; sh <<= 1) sum += __shfl_xor(sum, sh);
            l_run[r] = l_run[r] * alpha[r] + sum;
            m_run[r] = m_new;
#pragma unroll
            for (int c = 0; c < 4; ++c) Pw[wave][kq * 4 + r][c * 16 + li] = s[c][r];
        }
#pragma unroll
        for (int g = 0; g < CG; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) { o[g][e][0] *= alpha[0]; o[g][e][1] *= alpha[1]; o[g][e][2] *= alpha[2]; o[g][e][3] *= alpha[3]; }
        __syncthreads();                                   // P visible to the lanes that read it as an A operand
        // ---- O += P V for this wave's output columns
#pragma unroll
        for (int ks = 0; ks < KT / 16; ++ks) {
            const f32x4 pf = *reinterpret_cast<const f32x4*>(&Pw[wave][li][ks * 16 + 4 * kq]);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int kj = min(kb + ks * 16 + 4 * kq + u, p.N - 1);    // keys >= N carry probability 0
                const float* vr = base + (long long)kj * rs + 2 * p.C + d0 + 4 * li;
#pragma unroll
                for (int g = 0; g < CG; ++g) {
                    const f32x4 vf = *reinterpret_cast<const f32x4*>(vr + g * 64);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[g][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(pf[u], vf[e], o[g][e], 0, 0, 0);
                }
            }
        }
    }
    // ---- normalise and store: row = kq*4 + r; tile (g, e) column li <-> feature d0 + 64 g + 4 li + e
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = q0 + kq * 4 + r;
        if (i < p.N) {
            float* dst = p.out + ((long long)b * p.N + i) * p.C + d0 + 4 * li;
            const float inv = 1.0f / l_run[r];
#pragma unroll
            for (int g = 0; g < CG; ++g)
                *reinterpret_cast<f32x4*>(dst + g * 64) = f32x4{o[g][0][r] * inv, o[g][1][r] * inv, o[g][2][r] * inv, o[g][3][r] * inv};
        }
    }
}

// ---------------------------------------------------------------------------------------------
// ConvNeXt front half: y[t][c] = LN_c( bias[c] + sum_j w[c][j] * x[t - 3 + j][c] ) * scale[c] + shift[c]   (zero padding,
// layer_norm without affine, eps; scale/shift = row `cond` of the AdaLayerNorm embeddings).  One wavefront per frame, a
// lane owns the 16-byte channel vectors lane + 64 i (weights tap-major so that they vectorise too); two-pass moments.  HBM-bound: 8*C bytes per frame (the 7 taps of neighbouring frames
// come from L2).  Taps accumulate in ascending order from the bias.
// ---------------------------------------------------------------------------------------------
struct DwLnParams {
    const float* x;      // [B][N][C]
    const float* w;      // [7][C]  (tap-major: re-packed from the checkpoint's [C][1][7])
    const float* bias;   // [C]
    const float* scale;  // [C]
    const float* shift;  // [C]
    float* y;            // [B][N][C]
    int B, N, C;
    float eps;
    unsigned* rowmax;    // optional [B*N]: bit pattern of the largest finite |y| of each row (split16.h row mode, as LayerNormParams::rowmax)
};

constexpr int DWLN_MAXV = 4;    // 16-byte vectors per lane: C <= 1024, C % 4 == 0

__global__ __launch_bounds__(256) void dwconv_ln_kernel(const DwLnParams p) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long long)p.B * p.N) return;
    const int t = (int)(row % p.N);
    const float* xc = p.x + (row - t) * p.C;     // the clip's first frame
    const int cv = p.C / 4;                      // vectors per row; lane owns vectors lane + 64 i
    f32x4 v[DWLN_MAXV];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < DWLN_MAXV; ++i) {
        const int q = lane + 64 * i;
        v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (q < cv) {
            f32x4 acc = *reinterpret_cast<const f32x4*>(p.bias + 4 * q);
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const int tj = t - 3 + j;
                if (tj >= 0 && tj < p.N) {
                    const f32x4 xv = *reinterpret_cast<const f32x4*>(xc + (long long)tj * p.C + 4 * q);
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(p.w + (long long)j * p.C + 4 * q);
                    acc.x = fmaf(wv.x, xv.x, acc.x); acc.y = fmaf(wv.y, xv.y, acc.y);
                    acc.z = fmaf(wv.z, xv.z, acc.z); acc.w = fmaf(wv.w, xv.w, acc.w);
                }
            }
            v[i] = acc;
            sum += (acc.x + acc.y) + (acc.z + acc.w);
        }
    }
#pragma unroll
    for (int sh = 1; sh < 64; sh <<= 1) sum += __shfl_xor(sum, sh);
    const float mean = sum / (float)p.C;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < DWLN_MAXV; ++i) {
        if (lane + 64 * i < cv) {
            const f32x4 d = f32x4{v[i].x - mean, v[i].y - mean, v[i].z - mean, v[i].w - mean};
            sq = fmaf(d.x, d.x, sq); sq = fmaf(d.y, d.y, sq); sq = fmaf(d.z, d.z, sq); sq = fmaf(d.w, d.w, sq);
        }
    }
#pragma unroll
    for (int sh = 1; sh < 64; sh <<= 1) sq += __shfl_xor(sq, sh);
    const float rstd = 1.0f / sqrtf(sq / (float)p.C + p.eps);
    float* yr = p.y + row * p.C;
    unsigned rmax = 0;
#pragma unroll
    for (int i = 0; i < DWLN_MAXV; ++i) {
        const int q = lane + 64 * i;
        if (q < cv) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(p.scale + 4 * q), sh = *reinterpret_cast<const f32x4*>(p.shift + 4 * q);
            const f32x4 o = f32x4{(v[i].x - mean) * rstd * sc.x + sh.x, (v[i].y - mean) * rstd * sc.y + sh.y,
                                  (v[i].z - mean) * rstd * sc.z + sh.z, (v[i].w - mean) * rstd * sc.w + sh.w};
            *reinterpret_cast<f32x4*>(yr + 4 * q) = o;
            amax_acc4(rmax, o);
        }
    }
    if (p.rowmax) {
#pragma unroll
        for (int sh = 1; sh < 64; sh <<= 1) {
            const unsigned t2 = (unsigned)__shfl_xor((int)rmax, sh);
            rmax = t2 > rmax ? t2 : rmax;
        }
        if (lane == 0) p.rowmax[row] = rmax;
    }
}

// ---------------------------------------------------------------------------------------------
// ISTFTHead non-linearity.  The head Linear is packed so that column 2c is the log-magnitude and 2c+1 the phase of bin c
// (columns >= 2*bins are padding and stay 0): in place, (m, p) -> (min(exp(m), 100) * cos p, min(exp(m), 100) * sin p).
// ---------------------------------------------------------------------------------------------
struct PolarParams {
    float* x;            // [rows][pitch]
    long long rows;
    int pitch, bins;
};

__global__ __launch_bounds__(256) void polar_kernel(const PolarParams p) {
    const int pairs = p.pitch / 2;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= p.rows * pairs) return;
    const int c = (int)(gid % pairs);
    float* q = p.x + (gid / pairs) * p.pitch + 2 * c;
    float re = 0.f, im = 0.f;
    if (c < p.bins) {
        const float mag = fminf(expf(q[0]), 100.0f);
        float sn, cs;
        sincosf(q[1], &sn, &cs);
        re = mag * cs;
        im = mag * sn;
    }
    q[0] = re;
    q[1] = im;
}

// ---------------------------------------------------------------------------------------------
// ISTFT(padding="same") tail: the inverse-rFFT + window + overlap-add GEMM wrote y[b][n] for the trimmed range; divide by
// the overlap-added squared window  env(n) = sum over frames f covering position n + trim of w^2[n + trim - f*hop].
// env2[nfft] = w^2 (host-precomputed).
// ---------------------------------------------------------------------------------------------
struct EnvParams {
    float* y;            // [B][N*hop]
    const float* w2;     // [nfft]
    int B, N, hop, nfft;
};

__global__ __launch_bounds__(256) void istft_env_kernel(const EnvParams p) {
    const long long L = (long long)p.N * p.hop;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= L) return;
    const int pos = (int)gid + (p.nfft - p.hop) / 2;        // position in the un-trimmed overlap-add buffer
    const int f_hi = min(p.N - 1, pos / p.hop);
    const int f_lo = pos >= p.nfft ? (pos - p.nfft) / p.hop + 1 : 0;
    float env = 0.f;
    for (int f = f_lo; f <= f_hi; ++f) {                    // ascending frames, like fold's accumulation
        const int o = pos - f * p.hop;
        if (o >= 0 && o < p.nfft) env += p.w2[o];
    }
    for (int b = 0; b < p.B; ++b) p.y[(long long)b * L + gid] /= env;
}

}  // namespace ac
