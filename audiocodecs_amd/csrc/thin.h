// HBM-bound ends of the SEANet stack, where one side of the layer has 1 channel and a GEMM tile would
// be almost empty:
//   stem_kernel : [HF] EncodecEncoder.layers[0]  Conv1d(1, F, k)  sig[B][T] -> x[B][T][F] (raw and ELU'd)
//   head_kernel : [HF] EncodecDecoder.layers[-1] Conv1d(F, 1, k)  ELU'd y[B][T][F] -> sig[B][T]
// Both are causal with reflect padding ([HF] modeling_encodec.py:157-176, small-input rule :148-155) or, for
// Mimi's SEANet, zero padding; the stem also applies the length mask (audiocodecs/encodec.py:84-92, [HF]:589-590).
// Algorithmic bytes: stem 4*T*(1 + F*flavours), head 4*T*(F + 1) per clip; plain fp32 FMAs (the
// arithmetic is ~1 flop/B).  Accumulation order: bias first, then taps (and channels) ascending.
#pragma once
#include <hip/hip_runtime.h>
#include "tap_gemm.h"
#include "split16.h"

namespace ac {

struct ThinParams {
    const float* x;        // stem: [B][T]; head: [B][T][F]
    const float* w;        // stem: [F][k] (HF layout [F][1][k]); head: [k][F] (re-packed from [1][F][k])
    const float* bias;     // stem: [F]; head: [1]
    const float* rel_len;  // stem only, optional [B]
    float* y;              // stem: raw [B][T][F] (may be null); head: [B][T]
    float* y_elu;          // stem: ELU'd [B][T][F] (may be null)
    int B, T, F, k;
    int Lp;                // reflect base length: T if T > k-1 else k ([HF]:148-155)
    int pad;               // PAD_REFLECT (EnCodec) or PAD_ZERO (Mimi: pad_mode "constant", [HF] mimi :336-338)
    int padl;              // left padding in samples: k-1 (causal) or (k-1)/2 (DAC, symmetric zero padding)
    const float* alpha;    // stem: Snake parameters of the consumer -> y_elu = snake(y) instead of ELU(y) (DAC)
    const float* alpha_inv;
    int tanh_out;          // head: tanh on the output (DAC decoder, [HF] dac :439-440)
    unsigned* amax_out;    // stem: optional split16.h amax slot [B] of the output (raw and activated flavour)
};

__device__ __forceinline__ int reflect_src(int i, int T, int Lp, int pad) {
    int j = i;
    if (pad == PAD_REFLECT) j = i < 0 ? -i : (i >= Lp ? 2 * (Lp - 1) - i : i);
    return (j >= 0 && j < T) ? j : -1;
}

constexpr int STEM_TT = 2048;   // time steps per workgroup

// F must be a multiple of 4 (<= 128): thread = (time step, 4 channels); k <= 8.
__global__ __launch_bounds__(256) void stem_kernel(const ThinParams p) {
    __shared__ float xs[STEM_TT + THIN_MAXK];
    const int b = blockIdx.y, t0 = blockIdx.x * STEM_TT, tid = threadIdx.x;
    const float* xb = p.x + (long long)b * p.T;
    float alen = 3.0e38f;
    if (p.rel_len) alen = (float)p.T * p.rel_len[b];
    const int pad = p.k - 1;
    for (int e = tid; e < STEM_TT + pad; e += 256) {
        const int j = reflect_src(t0 - p.padl + e, p.T, p.Lp, p.pad);
        xs[e] = (j >= 0 && (float)j < alen) ? xb[j] : 0.f;
    }
    const int cq = p.F / 4;                 // float4 groups per time step
    const int c4 = (tid % cq) * 4;
    float w[4][THIN_MAXK], bv[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        bv[c] = p.bias[c4 + c];
#pragma unroll
        for (int j = 0; j < THIN_MAXK; ++j) w[c][j] = j < p.k ? p.w[(c4 + c) * p.k + j] : 0.f;
    }
    __syncthreads();
    const int tstep = 256 / cq;
    const long long ob = (long long)b * p.T * p.F;
    unsigned omax = 0;
    for (int tl = tid / cq; tl < STEM_TT; tl += tstep) {
        const int t = t0 + tl;
        if (t >= p.T) break;
        f32x4 acc = f32x4{bv[0], bv[1], bv[2], bv[3]};
#pragma unroll
        for (int j = 0; j < THIN_MAXK; ++j) {
            if (j < p.k) {
                const float xv = xs[tl + j];
                acc.x = fmaf(w[0][j], xv, acc.x); acc.y = fmaf(w[1][j], xv, acc.y);
                acc.z = fmaf(w[2][j], xv, acc.z); acc.w = fmaf(w[3][j], xv, acc.w);
            }
        }
        const long long o = ob + (long long)t * p.F + c4;
        amax_acc4(omax, acc);
        if (p.y) *reinterpret_cast<f32x4*>(p.y + o) = acc;
        if (p.y_elu) {
            f32x4 w;
            if (p.alpha) {
                const f32x4 al = *reinterpret_cast<const f32x4*>(p.alpha + c4), ai = *reinterpret_cast<const f32x4*>(p.alpha_inv + c4);
                w.x = snake1(acc.x, al.x, ai.x); w.y = snake1(acc.y, al.y, ai.y); w.z = snake1(acc.z, al.z, ai.z); w.w = snake1(acc.w, al.w, ai.w);
                amax_acc4(omax, w);                  // Snake can exceed |acc| (ELU cannot)
            } else {
                w = elu4(acc);
            }
            *reinterpret_cast<f32x4*>(p.y_elu + o) = w;
        }
    }
    if (p.amax_out) amax_flush(omax, amax_at(p.amax_out, b));
}

constexpr int HEAD_TT = 256;    // outputs per workgroup (one per thread)

// F multiple of 4, F <= 128, k <= 8.  LDS: (HEAD_TT + k - 1) rows of F (+4 pad) floats + the weights.
__global__ __launch_bounds__(256) void head_kernel(const ThinParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int F = p.F, FP = F + 4;
    float* xs = smem;                                   // [(HEAD_TT + pad)][FP]
    float* ws = smem + (HEAD_TT + THIN_MAXK) * FP;      // [k][F]
    const int b = blockIdx.y, t0 = blockIdx.x * HEAD_TT, tid = threadIdx.x;
    const int pad = p.k - 1, fq = F / 4;
    const float* xb = p.x + (long long)b * p.T * F;
    for (int e = tid; e < (HEAD_TT + pad) * fq; e += 256) {
        const int row = e / fq, q = e % fq;
        const int j = reflect_src(t0 - p.padl + row, p.T, p.Lp, p.pad);
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (j >= 0) v = *reinterpret_cast<const f32x4*>(xb + (long long)j * F + 4 * q);
        *reinterpret_cast<f32x4*>(&xs[row * FP + 4 * q]) = v;
    }
    for (int e = tid; e < p.k * F; e += 256) ws[e] = p.w[e];
    __syncthreads();
    const int t = t0 + tid;
    if (t >= p.T) return;
    float acc = p.bias[0];
    for (int j = 0; j < p.k; ++j) {
        const float* xr = &xs[(tid + j) * FP];
        const float* wr = &ws[j * F];
        for (int q = 0; q < fq; ++q) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(xr + 4 * q);
            const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + 4 * q);
            acc = fmaf(wv.x, xv.x, acc); acc = fmaf(wv.y, xv.y, acc);
            acc = fmaf(wv.z, xv.z, acc); acc = fmaf(wv.w, xv.w, acc);
        }
    }
    p.y[(long long)b * p.T + t] = p.tanh_out ? tanhf(acc) : acc;
}

// head4_kernel (round 4): the same layer for F % 16 == 0 with FOUR lanes per output sample, each over a quarter of the channels
// (all taps), the quarters added pairwise ((q0 + q1) + (q2 + q3)), the bias last.  head_kernel walks a sample's k F products as
// ONE sequential fp32 chain per thread behind a 256-row staging tile: two workgroups per CU, 448 dependent FMAs per thread for
// Mimi's 64 channels -- 2.7 TB/s on a layer that only reads (Mimi 128 x 10 s: 2.9 ms; DAC: 3 % of its step).  Here a workgroup
// stages 64 + k - 1 rows (21 KB: seven workgroups per CU), a thread's chain is k F / 4 long; LDS pitch F + 4 floats: the 16 lanes
// of a 16-byte read group (4 samples x 4 quarters) cover all 64 banks.  The summation ORDER differs from head_kernel's (fp32
// rounding only; both are within the parity tolerance of the waveform, tests/test_*_gpu_parity.py).
constexpr int HEAD4_TT = 64;    // outputs per workgroup

__global__ __launch_bounds__(256) void head4_kernel(const ThinParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int F = p.F, FP = F + 4;
    float* xs = smem;                                    // [(HEAD4_TT + pad)][FP]
    float* ws = smem + (HEAD4_TT + THIN_MAXK) * FP;      // [k][F]
    const int b = blockIdx.y, t0 = blockIdx.x * HEAD4_TT, tid = threadIdx.x;
    const int pad = p.k - 1, fq = F / 4, fqq = fq / 4;
    const float* xb = p.x + (long long)b * p.T * F;
    for (int e = tid; e < (HEAD4_TT + pad) * fq; e += 256) {
        const int row = e / fq, q = e % fq;
        const int j = reflect_src(t0 - p.padl + row, p.T, p.Lp, p.pad);
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (j >= 0) v = *reinterpret_cast<const f32x4*>(xb + (long long)j * F + 4 * q);
        *reinterpret_cast<f32x4*>(&xs[row * FP + 4 * q]) = v;
    }
    for (int e = tid; e < p.k * F; e += 256) ws[e] = p.w[e];
    __syncthreads();
    const int tl = tid >> 2, q = tid & 3;                // four adjacent lanes share a sample
    float acc = 0.f;
    for (int j = 0; j < p.k; ++j) {
        const float* xr = &xs[(tl + j) * FP + q * fqq * 4];
        const float* wr = &ws[j * F + q * fqq * 4];
        for (int g = 0; g < fqq; ++g) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(xr + 4 * g);
            const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + 4 * g);
            acc = fmaf(wv.x, xv.x, acc); acc = fmaf(wv.y, xv.y, acc);
            acc = fmaf(wv.z, xv.z, acc); acc = fmaf(wv.w, xv.w, acc);
        }
    }
    acc += __shfl_xor(acc, 1);                           // q0 + q1 | q2 + q3
    acc += __shfl_xor(acc, 2);                           // (q0 + q1) + (q2 + q3): the same value in the four lanes (fp32 + is commutative)
    const int t = t0 + tl;
    if (q == 0 && t < p.T) {
        const float v = acc + p.bias[0];
        p.y[(long long)b * p.T + t] = p.tanh_out ? tanhf(v) : v;
    }
}

// Polyphase windowed-sinc sample-rate conversion: what torchaudio.functional.resample applies at the
// Codec boundary (audiocodecs/codec.py:59-63,95-99): zero-pad (width, width + o), conv1d with the
// [n phases][taps] kernel at stride o, interleave the phases, truncate.  out[i*n + ph] =
// sum_k kern[ph][k] * x[i*o + k - width].  One thread per output sample; HBM-bound (4*(L + L_out) B/clip).
struct ResampleParams {
    const float* x;      // [B][L]
    const float* kern;   // [n][taps]
    float* y;            // [B][L_out]
    int B, L, L_out, n, o, taps, width;
};

__global__ __launch_bounds__(256) void resample_kernel(const ResampleParams p) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (gid >= p.L_out) return;
    const int i = (int)(gid / p.n), ph = (int)(gid % p.n);
    const float* xb = p.x + (long long)b * p.L;
    const float* kr = p.kern + (long long)ph * p.taps;
    const int base = i * p.o - p.width;
    float acc = 0.f;
    for (int k = 0; k < p.taps; ++k) {
        const int m = base + k;
        if (m >= 0 && m < p.L) acc = fmaf(kr[k], xb[m], acc);
    }
    p.y[(long long)b * p.L_out + gid] = acc;
}

}  // namespace ac
