// DAC residual vector quantiser with factorised, L2-normalised codes (descript-audio-codec
// dac/nn/quantize.py, mirrored by [HF] transformers models/dac/modeling_dac.py:103-172, 283-345), as
// /root/reference/audiocodecs/dac.py:96-99,117-119 runs it through `model.encode(..., n_quantizers=K)`:
//   per stage k:  z_e = in_proj_k(residual)                 1x1 conv H -> 8 (+ bias)
//                 idx = argmax_c -(|e|^2 - 2 e.c + |c|^2)   e = normalize(z_e), c = normalize(codebook_k[c])
//                 q   = z_e + (codebook_k[idx] - z_e)       (straight-through form, kept: it rounds)
//                 z_q = out_proj_k(q)                        1x1 conv 8 -> H (+ bias)
//                 residual -= z_q;   quantised += z_q
// One workgroup owns 16 frames for ALL K stages; the [16][H] residual tile lives in LDS (64 KB at H = 1024), so
// the residual makes ONE HBM round trip instead of 2K (18 x 0.9 GB at BASELINE configs[2]).  in_proj and the
// 1024-code search run on v_mfma_f32_16x16x4_f32 (K split over the 4 waves / code tiles interleaved over the
// waves), the rank-8 residual update on the VALU.  Decode (`quantizer.from_codes`, dac.py:126-128) is a gather
// over the pre-projected code table (rvq_decode_kernel over out_proj_k(codebook_k) + bias).
#pragma once
#include <hip/hip_runtime.h>
#include "tap_gemm.h"

namespace ac {

struct DacVqParams {
    const float* z;      // [F][H] encoder output, channels-last
    const float* win;    // [K][8][H]
    const float* bin;    // [K][8]
    const float* wout;   // [K][H][8]
    const float* bout;   // [K][H]
    const float* cb;     // [K][C][8] codebooks (un-normalised)
    const float* cbn;    // [K][C/16][2][64] normalised codebooks in MFMA B order: [t][h][l] = cn[16t + (l&15)][4h + (l>>4)]
    const float* c2;     // [K][C] |cn|^2
    long long* toks;     // [F][K]
    float* qsum;         // optional [F][H]: the quantised representation model.encode returns
    int F, H, C, K;
};

constexpr int DAC_D = 8;       // codebook_dim of every published DAC model
constexpr int DAC_FR = 16;     // frames per workgroup

inline size_t dac_vq_lds_bytes(int H) { return (size_t)(DAC_FR * (H + 4) + 512 + 3 * 128 + 16 + 64 + 64 + 16) * 4; }

__global__ __launch_bounds__(256) void dac_vq_encode_kernel(const DacVqParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int H = p.H, HP = H + 4;
    float* R = smem;                        // [16][HP] residual
    float* part = R + DAC_FR * HP;          // [4 waves][16][8] in_proj partial sums
    float* pl = part + 512;                 // [16][8] z_e
    float* pn = pl + 128;                   // [16][8] normalize(z_e)
    float* qs = pn + 128;                   // [16][8] straight-through code vectors
    float* e2 = qs + 128;                   // [16]    |e|^2
    float* wv = e2 + 16;                    // [4][16] per-wave best score
    int* wi = reinterpret_cast<int*>(wv + 64);   // [4][16] per-wave best index
    int* tokf = wi + 64;                    // [16]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kq = lane >> 4;
    const int f0 = blockIdx.x * DAC_FR;
    for (int e = tid; e < DAC_FR * (H / 4); e += 256) {
        const int row = e / (H / 4), q = e % (H / 4);
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (f0 + row < p.F) v = *reinterpret_cast<const f32x4*>(p.z + (long long)(f0 + row) * H + 4 * q);
        *reinterpret_cast<f32x4*>(&R[row * HP + 4 * q]) = v;
    }
    float qacc[4][DAC_FR];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int f = 0; f < DAC_FR; ++f) qacc[i][f] = 0.f;
    __syncthreads();

    for (int k = 0; k < p.K; ++k) {
        // ---- (a) z_e = in_proj(residual): wave w contracts hidden dims [w*H/4, (w+1)*H/4)
        {
            const float* win = p.win + (long long)k * DAC_D * H;
            f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
            const int kb = wave * (H / 4);
            for (int ks = 0; ks < H / 64; ++ks) {
                const int kk = kb + ks * 16 + 4 * kq;
                const f32x4 a = *reinterpret_cast<const f32x4*>(&R[li * HP + kk]);
                f32x4 b = f32x4{0.f, 0.f, 0.f, 0.f};
                if (li < DAC_D) b = *reinterpret_cast<const f32x4*>(win + (long long)li * H + kk);
#pragma unroll
                for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc, 0, 0, 0);
            }
            if (li < DAC_D) {
#pragma unroll
                for (int r = 0; r < 4; ++r) part[(wave * 16 + kq * 4 + r) * 8 + li] = acc[r];   // C: col = li, row = kq*4+r
            }
        }
        __syncthreads();
        if (tid < 128) {
            const int f = tid >> 3, n = tid & 7;
            float v = ((part[(0 * 16 + f) * 8 + n] + part[(1 * 16 + f) * 8 + n]) + part[(2 * 16 + f) * 8 + n]) + part[(3 * 16 + f) * 8 + n];
            v += p.bin[k * DAC_D + n];
            pl[f * 8 + n] = v;
            // F.normalize: x / max(|x|_2, 1e-12) over the 8 code dims (8 neighbouring lanes)
            float ss = v * v;
            ss += __shfl_xor(ss, 1); ss += __shfl_xor(ss, 2); ss += __shfl_xor(ss, 4);
            const float nv = v / fmaxf(sqrtf(ss), 1e-12f);
            pn[f * 8 + n] = nv;
            float s2 = nv * nv;
            s2 += __shfl_xor(s2, 1); s2 += __shfl_xor(s2, 2); s2 += __shfl_xor(s2, 4);
            if (n == 0) e2[f] = s2;
        }
        __syncthreads();
        // ---- (b) nearest code: code tiles interleaved over the waves, first index wins ties
        {
            const float* cbn = p.cbn + (long long)k * p.C * DAC_D;
            const float* c2 = p.c2 + (long long)k * p.C;
            const float a0 = pn[li * 8 + kq], a1 = pn[li * 8 + 4 + kq];
            float e2r[4], best[4];
            int bi[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { e2r[r] = e2[kq * 4 + r]; best[r] = -3.0e38f; bi[r] = 0; }
            for (int t = wave; t < p.C / 16; t += 4) {
                const float b0 = cbn[(t * 2 + 0) * 64 + lane], b1 = cbn[(t * 2 + 1) * 64 + lane];
                f32x4 d = f32x4{0.f, 0.f, 0.f, 0.f};
                d = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, d, 0, 0, 0);
                const int code = t * 16 + li;
                const float cc = c2[code];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float sc = -((e2r[r] - 2.0f * d[r]) + cc);      // (-dist) of dac/nn/quantize.py
                    if (sc > best[r]) { best[r] = sc; bi[r] = code; }
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int sh = 1; sh < 16; sh <<= 1) {
                    const float ob = __shfl_xor(best[r], sh);
                    const int oi = __shfl_xor(bi[r], sh);
                    if (ob > best[r] || (ob == best[r] && oi < bi[r])) { best[r] = ob; bi[r] = oi; }
                }
                if (li == 0) { wv[wave * 16 + kq * 4 + r] = best[r]; wi[wave * 16 + kq * 4 + r] = bi[r]; }
            }
        }
        __syncthreads();
        if (tid < DAC_FR) {
            float b = wv[tid];
            int ix = wi[tid];
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const float ob = wv[w * 16 + tid];
                const int oi = wi[w * 16 + tid];
                if (ob > b || (ob == b && oi < ix)) { b = ob; ix = oi; }
            }
            tokf[tid] = ix;
            if (f0 + tid < p.F) p.toks[(long long)(f0 + tid) * p.K + k] = ix;
        }
        __syncthreads();
        // ---- (c) q = z_e + (codebook[idx] - z_e);  z_q = out_proj(q);  residual -= z_q
        if (tid < 128) {
            const int f = tid >> 3, n = tid & 7;
            const float q = p.cb[((long long)k * p.C + tokf[f]) * DAC_D + n];
            const float pv = pl[f * 8 + n];
            qs[f * 8 + n] = __fadd_rn(pv, __fsub_rn(q, pv));
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int d = tid + 256 * i;
            if (d < H) {
                const float* wr = p.wout + ((long long)k * H + d) * DAC_D;
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(wr), w1 = *reinterpret_cast<const f32x4*>(wr + 4);
                const float bo = p.bout[(long long)k * H + d];
#pragma unroll
                for (int f = 0; f < DAC_FR; ++f) {
                    const f32x4 q0 = *reinterpret_cast<const f32x4*>(&qs[f * 8]), q1 = *reinterpret_cast<const f32x4*>(&qs[f * 8 + 4]);
                    float zq = w0.x * q0.x;
                    zq = fmaf(w0.y, q0.y, zq); zq = fmaf(w0.z, q0.z, zq); zq = fmaf(w0.w, q0.w, zq);
                    zq = fmaf(w1.x, q1.x, zq); zq = fmaf(w1.y, q1.y, zq); zq = fmaf(w1.z, q1.z, zq); zq = fmaf(w1.w, q1.w, zq);
                    zq = __fadd_rn(zq, bo);
                    R[f * HP + d] = __fsub_rn(R[f * HP + d], zq);
                    qacc[i][f] = __fadd_rn(qacc[i][f], zq);
                }
            }
        }
        __syncthreads();
    }
    if (p.qsum) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int d = tid + 256 * i;
            if (d < H) {
#pragma unroll
                for (int f = 0; f < DAC_FR; ++f)
                    if (f0 + f < p.F) p.qsum[(long long)(f0 + f) * H + d] = qacc[i][f];
            }
        }
    }
}

}  // namespace ac
