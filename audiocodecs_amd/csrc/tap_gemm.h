// tap_gemm: every convolution of the SEANet stack as ONE fp32-MFMA GEMM over channels-last rows.
//
// Activations live in HBM channels-last, [B][time][C] (C contiguous).  In that layout
//   * a stride-1 conv with k taps reads, for output row m, the k consecutive input rows
//     m-(k-1)..m  (causal left pad)                     -> A row = k*Cin contiguous-ish floats;
//   * a strided conv with k = 2*s (every down-sampler of EnCodec) is a 2-tap stride-1 conv over
//     the input viewed as rows of s time steps (s*Cin floats): out[m] = Xr[m-1]*W0 + Xr[m]*W1;
//   * a transposed conv with k = 2*s is a 2-tap stride-1 conv whose output row m holds the s
//     output time steps m*s..m*s+s-1 (s*Cout floats): exactly the channels-last output layout,
//     and the reference's right-trim of k-s samples is "the row after the last" -- never computed;
//   * a linear layer (LSTM input projection) is the 1-tap case.
// So all of [HF] EncodecConv1d (:157-176), EncodecConvTranspose1d (:206-233) and the ResBlock's
// 1x1 convs + shortcut (:277-282, fused as a K-concatenation of two sources) run through this one
// kernel: out[b][m][n] = bias[n] + sum_seg sum_j sum_c act(Xr_seg[b][m-(J-1)+j][c]) * W[n][k(seg,j,c)].
//
// Padding never materialises: the loader maps a (row, column) of the LDS slab to a source time
// index with the reference's reflect rule -- including the small-input workaround of
// [HF]:148-155 (zero-extend to max_pad+1, reflect, drop) -- or to zero (transposed conv, masked
// samples, rows past the end).  ELU ([HF] nn.ELU before every conv but the first) is applied once
// per staged element.
//
// MFMA: v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulate; 157 TF peak).  Per 16-wide
// k-step a lane reads ONE 16-byte vector of A and of B (its 4 consecutive k's) from LDS and issues
// 4 MFMAs; MFMA u of the step contracts k = 16*ks + 4*(lane>>4) + u.  The k order inside the
// accumulation chain is therefore a fixed permutation -- deterministic run to run.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ac {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct TapSeg {
    const float* x;     // [B][L][cin] with strides below
    long long bs;       // batch stride (floats)
    long long ts;       // time-step stride (floats)
    const float* rel_len;  // optional [B] relative lengths: zero samples with !(t < L*rel_len[b])
    int L;              // time steps per item
    int cin;            // channels per time step
    int cin_shift;      // log2(cin) or -1
    int s;              // time steps per reshaped row (row width Cw = s*cin)
    int J;              // taps = reshaped rows per output row
    int Lp;             // reflect period base: L if L > max_pad else max_pad+1 ([HF]:148-155)
    int lim;            // valid padded range is [-(J-1)*s, lim): lim = L + extra_padding
    int reflect;        // padding rule for indices outside [0, L): 0 zero, 1 reflect, 2 replicate (PAD_*)
    int elu;            // 1: ELU(alpha=1) on load
    int kofs;           // offset of this segment inside a packed weight row
    int pad;            // left padding in time steps: output row m, tap j, in-row step tp reads time
    int dil;            //   t = (m + j*dil)*s + tp - pad      (causal k-tap conv: pad = (J-1)*s, dil = 1;
                        //   DAC: symmetric pad 3*dil for the dilated k7 conv, ceil(s/2) for the strided ones)
    const unsigned* amax;   // split16.h: [B] largest-magnitude bits of this tensor (fp16 two-plane arithmetic only)
    int amax_n;             //   clips the slot covers (host side only)
};

struct TapGemmParams {
    TapSeg seg[2];
    int nseg;
    const float* w;     // packed [N][Ktot], K contiguous; k = kofs + j*Cw + c
    const float* bias;  // [N]
    float* y;           // out[b*y_bs + m*y_rs + n] (may be null when only y_elu is wanted)
    float* y_elu;       // optional second output ELU(out), same strides: the consumer layers of the
                        // SEANet stack all start with nn.ELU, so activations are activated ONCE here, in
                        // the producing layer's epilogue, not on every staged load (profiles/r1_tapgemm_investigation.md)
    long long y_bs, y_rs;
    int B, M, N, Ktot;
    int mtiles, ntiles;
    // optional epilogue terms (Mimi): v = acc + bias; gelu -> v = GELU(v) (exact erf form); scale -> v *= scale[n]
    // (LayerScale); res -> v = res[b][m][n] + v (identity shortcut / residual stream; may alias y)
    const float* scale;
    const float* res;
    long long res_bs, res_rs;
    int gelu;
    // DAC: the activated flavour is Snake (x + sin^2(alpha x)/(alpha + 1e-9), per channel n % alpha_n) when
    // `alpha` is set, ELU otherwise; `tanh_out` applies tanh to the raw output (decoder head).  A padded
    // transposed conv writes rows that start p*cout floats before the item: y_off shifts the flat index and
    // only indices in [0, y_len) are stored (y_len = 0: no check).
    const float* alpha;
    const float* alpha_inv;
    int alpha_n;
    int tanh_out;
    long long y_off, y_len;
    int n_valid;        // > 0: only columns n < n_valid are stored (N is padded to the tile width with zero weight rows)
    // split16.h (tap_gemm6 NP = 2): per-output-channel 2^-s of the weight rows; optional amax slot [B] of the output
    // (largest magnitude over the raw and the activated flavour)
    const float* winv;
    unsigned* amax_out;
    // ROW mode (a linear layer over a merged row matrix: B = 1, one segment, one tap): seg[0].amax holds one word per ROW, the
    // scale is the row's own -- the result of a row then depends on nothing but that row (clips stay independent of their
    // batch neighbours); amax_out_rows: optional per-row words of the output
    int amax_rows;
    unsigned* amax_out_rows;
    int epi_direct;     // tap_gemm6, split16: the epilogue stores straight from the accumulators (no LDS staging, no barriers): a lane's
                        // 32 x 32 tile column is one output channel, so a 4-byte store instruction covers two rows x 128 contiguous bytes
    int stagger;        // experiment (AC_TAP_STAGGER cycles): the first resident workgroups start after a pseudo-random share of it
    unsigned long long* clk;   // diagnostics (ac_debug_clock): [0] += shader-clock ticks, [1] += 100 MHz real-time ticks of
                               // every workgroup of a tap_gemm6 launch; null = off
};

enum { PAD_ZERO = 0, PAD_REFLECT = 1, PAD_REPLICATE = 2 };

// GELU as torch.nn.functional.gelu(approximate="none") defines it: 0.5 x (1 + erf(x / sqrt(2))).
// Round 4: erf without ocml's erff.  erff is two polynomial branches under a divergent branch (a wave with arguments on both sides of
// |u| = 1 runs both: ~28 vector instructions per element), and the GELU epilogue of a K = 768 / 512 linear layer (WavTokenizer's
// ConvNeXt pwconv1, Mimi's fc1) is 60 % as long as its main loop.  One branch instead:
//     erf(t) = 1 - exp(-q(t)),  q(t) = -ln(erfc(t)) = t P7(t)   (t = |u|; P7 fitted on [0, 4], log2(e) folded in: 2^-s on v_exp_f32)
// 14 instructions, one transcendental.  Measured over 6.4e6 arguments (operation-by-operation fp32 emulation against float64,
// tests/test_gelu_emulation.py): |erf error| <= 1.0e-7, |GELU error| <= 4.7e-7 at |x| = 4.4 (one ulp of the result) -- torch's own
// fp32 GELU: 1.2e-6 --, relative error where |GELU| > 1e-3: 7.6e-5 (torch: 1.0e-3, the cancellation in 1 + erf for x < -3).
// Beyond the fitted range the polynomial keeps s >= 24.7: erf stays within 4e-8 of +-1 up to inf; NaN propagates.
__device__ __forceinline__ float gelu1(float v) {
    const float u = v * 0.70710678118654752440f;
    const float t = fabsf(u);
    float p = 4.536090636975132e-05f;
    p = fmaf(p, t, -0.00044552396866492927f);
    p = fmaf(p, t, 0.001489486894570291f);
    p = fmaf(p, t, 0.0007745670736767352f);
    p = fmaf(p, t, -0.02825363539159298f);
    p = fmaf(p, t, 0.1484815925359726f);
    p = fmaf(p, t, 0.9184163808822632f);
    p = fmaf(p, t, 1.6279085874557495f);
    const float er = copysignf(1.0f - __builtin_amdgcn_exp2f(-(p * t)), u);
    const float hv = 0.5f * v;
    return fmaf(hv, er, hv);
}

// ELU(alpha=1) as torch evaluates it on CPU: x > 0 ? x : exp(x) - 1 (not expm1).  exp goes through the
// hardware v_exp_f32 (2^x, ~1 ulp) after one multiply by log2(e): absolute error of the result
// <= ~1.2e-7, the same class as the reference's own vectorised expf, at 5 instructions per element
// instead of ~15 -- staging instruction count bounds the MFMA kernels (profiles/r1_tapgemm_trace.md).
__device__ __forceinline__ float elu1(float v) { return v > 0.f ? v : (__expf(v) - 1.0f); }

__device__ __forceinline__ f32x4 elu4(f32x4 v) {
    v.x = elu1(v.x); v.y = elu1(v.y); v.z = elu1(v.z); v.w = elu1(v.w);
    return v;
}

// Snake1d ([HF] dac :95-100): x + (alpha + 1e-9)^-1 * sin(alpha*x)^2; ainv is precomputed in fp32 on the host.
// sin(t)^2 for Snake: three-term Cody-Waite reduction by pi/2 (t - k pi/2 is exact in the first fma for |t| < 2^15), sin on
// [-pi/4, pi/4] as r + r z P(z) (degree-3 minimax in z = r^2), cos^2 = 1 - sin^2 for odd k.  Measured against float64 over 8e6
// arguments up to |t| ~ 2e4 (tools/experiments/sin2_check.py): max abs error 1.2e-7 (the fp32 sinf(t)^2 of the reference: 0.9e-7)
// at ~19 instructions; ocml's sinf, which this replaces in every Snake epilogue of the DAC path, is ~3x that.  Arguments beyond
// 2^15, infinities and NaNs take sinf.
__device__ __forceinline__ float sin2_f32(float t) {
    if (__builtin_expect(!(__builtin_fabsf(t) < 32768.f), 0)) {
        const float s = sinf(t);
        return __fmul_rn(s, s);
    }
    const float k = __builtin_rintf(__fmul_rn(t, 0.636619747f));
    float r = __fmaf_rn(k, -0x1.921fb6p+0f, t);
    r = __fmaf_rn(k, 0x1.777a5cp-25f, r);
    r = __fmaf_rn(k, 0x1.ee59dap-50f, r);
    const float z = __fmul_rn(r, r);
    float q = __fmaf_rn(2.7181215500604594e-06f, z, -0.00019839312881231308f);
    q = __fmaf_rn(q, z, 0.008333329111337662f);
    q = __fmaf_rn(q, z, -0.1666666716337204f);
    const float sn = __fmaf_rn(__fmul_rn(r, z), q, r);
    const float s2 = __fmul_rn(sn, sn);
    return ((int)k & 1) ? __fsub_rn(1.f, s2) : s2;
}
__device__ __forceinline__ float snake1(float v, float a, float ainv) {
    return __fadd_rn(v, __fmul_rn(ainv, sin2_f32(__fmul_rn(a, v))));
}

// the activated flavour of an output element (what the consuming conv reads)
__device__ __forceinline__ float act1(const TapGemmParams& p, float v, int n) {
    if (p.alpha) {
        const int c = n % p.alpha_n;
        return snake1(v, p.alpha[c], p.alpha_inv[c]);
    }
    return elu1(v);
}

// bias-added accumulator -> layer output, in the reference's operation order: scale * x, then residual + (.)
__device__ __forceinline__ float epilogue1(const TapGemmParams& p, float v, int n, long long res_off) {
    if (p.gelu) v = gelu1(v);
    if (p.scale) v = __fmul_rn(p.scale[n], v);
    if (p.res) v = __fadd_rn(p.res[res_off], v);
    if (p.tanh_out) v = tanhf(v);
    return v;
}

constexpr int GEN_EXTRA = 56; // halo rows of the generic kernel's A slab: (J-1)*dil <= 6*9
constexpr int KC = 32;        // K chunk staged per iteration
constexpr int KCP = KC + 4;   // LDS pitch (floats): keeps 16-B alignment, spreads banks

// Source value for padded time index i (may be <0 or >= L) and channel ci of item b.
__device__ __forceinline__ long long src_index(const TapSeg& sg, int i) {
    if (i >= sg.lim) return -1;
    int j = i;
    if (sg.reflect == PAD_REFLECT) {
        if (j < 0) j = -j;
        else if (j >= sg.Lp) j = 2 * (sg.Lp - 1) - j;
    } else if (sg.reflect == PAD_REPLICATE) {     // [HF] mimi :1199-1208 down-sampler: F.pad(mode="replicate")
        j = j < 0 ? 0 : (j >= sg.L ? sg.L - 1 : j);
    }
    if (j < 0 || j >= sg.L) return -1;   // zero padding / zero-extension of the small-input rule / guard
    return j;
}

template <int WGM, int WGN, int WM, int WN, bool VEC>
__global__ __launch_bounds__(WGM* WGN * 64) void tap_gemm_kernel(const TapGemmParams p) {
    constexpr int BM = WGM * WM * 16, BN = WGN * WN * 16, NT = WGM * WGN * 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                              // [(BM + GEN_EXTRA)][KCP]
    float* Ws = smem + (BM + GEN_EXTRA) * KCP;     // [BN][KCP]

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

    for (int si = 0; si < p.nseg; ++si) {
        const TapSeg& sg = p.seg[si];
        const int Cw = sg.s * sg.cin;
        const int R = BM + (sg.J - 1) * sg.dil;
        const float* xb = sg.x + (long long)b * sg.bs;
        float alen = 3.0e38f;
        if (sg.rel_len) alen = (float)sg.L * sg.rel_len[b];
        for (int c0 = 0; c0 < Cw; c0 += KC) {
            // ---- stage the A slab: rows m0-(J-1) .. m0+BM-1, columns c0..c0+KC of the reshaped input
            if (VEC) {
                for (int e = tid; e < R * (KC / 4); e += NT) {
                    const int row = e / (KC / 4), q = e % (KC / 4);
                    const int c = c0 + 4 * q;
                    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (c < Cw) {
                        const int r = m0 + row;
                        const int tp = sg.cin_shift >= 0 ? (c >> sg.cin_shift) : (c / sg.cin);
                        const int ci = c - tp * sg.cin;
                        const long long j = src_index(sg, r * sg.s + tp - sg.pad);
                        if (j >= 0 && (float)j < alen) {
                            v = *reinterpret_cast<const f32x4*>(xb + j * sg.ts + ci);
                            if (sg.elu) { v.x = elu1(v.x); v.y = elu1(v.y); v.z = elu1(v.z); v.w = elu1(v.w); }
                        }
                    }
                    *reinterpret_cast<f32x4*>(&As[row * KCP + 4 * q]) = v;
                }
            } else {
                for (int e = tid; e < R * KC; e += NT) {
                    const int row = e / KC, cc = e % KC;
                    const int c = c0 + cc;
                    float v = 0.f;
                    if (c < Cw) {
                        const int r = m0 + row;
                        const int tp = c / sg.cin;
                        const int ci = c - tp * sg.cin;
                        const long long j = src_index(sg, r * sg.s + tp - sg.pad);
                        if (j >= 0 && (float)j < alen) {
                            v = xb[j * sg.ts + ci];
                            if (sg.elu) v = elu1(v);
                        }
                    }
                    As[row * KCP + cc] = v;
                }
            }
            for (int j = 0; j < sg.J; ++j) {
                // ---- stage the weight chunk for tap j: Ws[n][kk] = w[n0+n][kofs + j*Cw + c0 + kk]
                const long long kbase = (long long)sg.kofs + (long long)j * Cw + c0;
                if (VEC) {
                    for (int e = tid; e < BN * (KC / 4); e += NT) {
                        const int n = e / (KC / 4), q = e % (KC / 4);
                        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
                        if (n0 + n < p.N && c0 + 4 * q < Cw)
                            v = *reinterpret_cast<const f32x4*>(p.w + (long long)(n0 + n) * p.Ktot + kbase + 4 * q);
                        *reinterpret_cast<f32x4*>(&Ws[n * KCP + 4 * q]) = v;
                    }
                } else {
                    for (int e = tid; e < BN * KC; e += NT) {
                        const int n = e / KC, cc = e % KC;
                        float v = 0.f;
                        if (n0 + n < p.N && c0 + cc < Cw) v = p.w[(long long)(n0 + n) * p.Ktot + kbase + cc];
                        Ws[n * KCP + cc] = v;
                    }
                }
                __syncthreads();
                // ---- MFMA over this (chunk, tap)
#pragma unroll
                for (int ks = 0; ks < KC / 16; ++ks) {
                    f32x4 af[WM], bf[WN];
#pragma unroll
                    for (int a = 0; a < WM; ++a)
                        af[a] = *reinterpret_cast<const f32x4*>(&As[((wm * WM + a) * 16 + li + j * sg.dil) * KCP + ks * 16 + 4 * kq]);
#pragma unroll
                    for (int c = 0; c < WN; ++c)
                        bf[c] = *reinterpret_cast<const f32x4*>(&Ws[((wn * WN + c) * 16 + li) * KCP + ks * 16 + 4 * kq]);
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int a = 0; a < WM; ++a)
#pragma unroll
                            for (int c = 0; c < WN; ++c)
                                acc[a][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[a][u], bf[c][u], acc[a][c], 0, 0, 0);
                }
                __syncthreads();
            }
        }
    }
    // ---- epilogue: C layout of 16x16x4: col = lane&15, row = (lane>>4)*4 + reg
    const long long yoff = (long long)b * p.y_bs;
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int c = 0; c < WN; ++c) {
            const int n = n0 + (wn * WN + c) * 16 + li;
            if (n < (p.n_valid ? p.n_valid : p.N)) {
                const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + (wm * WM + a) * 16 + kq * 4 + r;
                    const long long fi = (long long)m * p.y_rs + n + p.y_off;     // flat index inside the item
                    if (m < p.M && (p.y_len == 0 || (fi >= 0 && fi < p.y_len))) {
                        const float v = epilogue1(p, acc[a][c][r] + bv, n, (long long)b * p.res_bs + (long long)m * p.res_rs + n);
                        if (p.y) p.y[yoff + fi] = v;
                        if (p.y_elu) p.y_elu[yoff + fi] = act1(p, v, n);
                    }
                }
            }
        }
}

template <int WGM, int WGN, int WM, int WN>
constexpr size_t tap_gemm_lds_bytes() {
    return (size_t)((WGM * WM * 16 + GEN_EXTRA) * KCP + WGN * WN * 16 * KCP) * sizeof(float);
}

}  // namespace ac
