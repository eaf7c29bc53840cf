// LSTM recurrence of [HF] EncodecLSTM (:236-249): nn.LSTM(D, D, num_layers) over [T,B,D], zero
// initial state, gate order i,f,g,o, followed by the module's skip connection (out + in).
//
// The input projection of layer 0, x_t * W_ih^T + (b_ih + b_hh) for ALL t, is one tap_gemm (the 1-tap
// case) written as gin[t][b][4D].  What is left is inherently sequential in t.  One launch per time
// step (a dependent kernel boundary costs ~1.5-4 us on MI355X, no more than an in-kernel grid barrier
// and without its residency hazards); the layers are pipelined ACROSS launches ("wavefront"): launch s
// runs, as independent workgroup roles,
//     role 0: layer-0 cell step            t = s
//     role 1: layer-1 input projection     t = s-1   (W_ih1 * h0[t], needs only launch s-1's result)
//     role 2: layer-1 cell step            t = s-2
// so a 2-layer LSTM over T steps takes T+2 launches instead of 2T (+ no separate projection GEMM).
//
// Per role and step: workgroup = 32 clips x 4 hidden units (= 16 gate columns: 4 gates x 4 units, so
// the cell update is local to the workgroup).  4 waves = 2 clip sub-tiles x 2 K-halves; each wave
// runs v_mfma_f32_16x16x4_f32 over its half of K = D with the A operand (h_{t-1} or h0[t]) and the B
// operand (weight slice, pre-packed in fragment order) loaded straight from L2 into registers as
// 16-byte vectors -- all loads of the step issued up front; the K-halves meet in LDS, then 128
// threads finish the cell (their gin / c / skip operands were prefetched before the MFMAs).
#pragma once
#include <hip/hip_runtime.h>
#include "tap_gemm.h"

namespace ac {

struct LstmRole {
    int active;          // 0: this role idles in this launch
    int kind;            // 0: cell step, 1: input projection (writes gin of the next layer)
    const float* a;      // A operand (h_{t-1} for a step, lower layer's h_t for a projection) in MFMA A-fragment
                         // order [ceil(B/16)][D/16 ksteps][64 lanes][4] (see hfrag_index); null: zero
    const float* wpk;    // packed weights: [D/4 unit groups][D/16 ksteps][64 lanes][4]
    const float* bias;   // projection only: [4D] (b_ih + b_hh)
    const float* gin;    // step: [B][4D] pre-activations from the projection (row b at gin + b*4D)
    float* gout;         // projection: [B][4D] destination
    float* hnext;        // step: h_t, written in the same A-fragment order
    float* c;            // step: [B][D] cell state
    int first;           // step: t == 0 (no recurrent term, c = 0)
    const float* skip;   // last layer: module input row (row b at skip + b*skip_bs)
    float* yout;         // last layer: h + skip            (row b at yout + b*y_bs)
    float* yout_elu;     // last layer: ELU(h + skip), the flavour the following conv reads
    long long skip_bs, y_bs;
};

struct LstmLaunchParams {
    LstmRole role[3];
    int B, D;
};

// Gate non-linearities on the hardware exp2 / rcp units (v_exp_f32, v_rcp_f32: ~1 ulp each): absolute
// error ~1e-7, the same class as the reference's vectorised CPU kernels; they sit on the critical path
// of every one of the 1504 dependent launches.
__device__ __forceinline__ float sigmoidf_(float x) { return __frcp_rn(1.0f + __expf(-x)); }
// the same two functions with v_rcp_f32 (1 ulp) in place of the correctly rounded reciprocal (an 11-instruction dependent
// sequence, five of them per gate thread and time step): lstm_persist16.h, whose time step IS this dependency chain
__device__ __forceinline__ float sigmoid_rcp(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_rcp(float x) {
    const float a = fminf(fabsf(x), 15.0f);
    const float t = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * a));
    return copysignf(t, x);
}
__device__ __forceinline__ float tanhf_(float x) {
    const float a = fminf(fabsf(x), 15.0f);                 // tanh(15) == 1 in fp32; avoids exp overflow
    const float t = 1.0f - 2.0f * __frcp_rn(1.0f + __expf(2.0f * a));
    return copysignf(t, x);
}

// h is only ever read back as the A operand of the next launch, so it is STORED in A-fragment order:
// element (clip b, unit k) lives where lane (b%16, (k%16)/4) of k-step k/16 loads its 16-byte vector.
// A wave's A loads are then 1 KiB contiguous each (measured: row-major h cost 4.2 of 13 us per launch
// as 16 x 64-byte segments per load instruction).
__device__ __forceinline__ long long hfrag_index(int b, int k, int D) {
    return ((((long long)(b >> 4) * (D >> 4) + (k >> 4)) * 64 + (((k & 15) >> 2) << 4) + (b & 15)) << 2) + (k & 3);
}

template <int KS>   // 16-wide k-steps per K-half: D = 32 * KS
__global__ __launch_bounds__(256) void lstm_step_kernel(const LstmLaunchParams p) {
    __shared__ float red[2][2][16][17];
    const LstmRole& R = p.role[blockIdx.z];
    if (!R.active) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int msub = wave & 1, khalf = wave >> 1;
    const int li = lane & 15, kq = lane >> 4;
    const int ug = blockIdx.x;
    const int b0 = blockIdx.y * 32;
    constexpr int D = 32 * KS;
    const bool step = R.kind == 0;

    // ---- epilogue operands first (128 threads: clip bl, unit ul), so their latency hides under the MFMAs
    const int bl = tid >> 2, ul = tid & 3;
    const int eb = b0 + bl, eu = ug * 4 + ul;
    const bool ethread = tid < 128 && eb < p.B;
    float gpre[4] = {0.f, 0.f, 0.f, 0.f}, cprev = 0.f, skipv = 0.f;
    if (ethread) {
        if (step) {
            const float* g = R.gin + (long long)eb * (4 * D);
#pragma unroll
            for (int q = 0; q < 4; ++q) gpre[q] = g[q * D + eu];
            if (!R.first) cprev = R.c[(long long)eb * D + eu];
            if (R.skip) skipv = R.skip[(long long)eb * R.skip_bs + eu];
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) gpre[q] = R.bias[q * D + eu];
        }
    }

    f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    if (R.a && !(step && R.first)) {
        const float* hrow = R.a + (((long long)((b0 >> 4) + msub) * (D / 16) + (long long)khalf * KS) * 64 + lane) * 4;
        const float* wrow = R.wpk + ((long long)ug * (D / 16) + (long long)khalf * KS) * 256 + lane * 4;
        f32x4 a[KS], w[KS];
#pragma unroll
        for (int i = 0; i < KS; ++i) {
#if defined(LSTM_DBG_SKIP) && (LSTM_DBG_SKIP & 2)
            a[i] = f32x4{1.f, 2.f, 3.f, (float)i};
#else
            a[i] = *reinterpret_cast<const f32x4*>(hrow + (long long)i * 256);
#endif
#if defined(LSTM_DBG_SKIP) && (LSTM_DBG_SKIP & 1)
            w[i] = f32x4{1.f, 2.f, 3.f, (float)i};
#else
            w[i] = *reinterpret_cast<const f32x4*>(wrow + (long long)i * 256);
#endif
        }
        // keep ALL loads of the step in flight before the first MFMA (hipcc otherwise re-serialises
        // them behind the MFMAs to save registers: ~8 dependent L2 round trips per step)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < KS; ++i) {
#pragma unroll
            for (int u = 0; u < 4; u += 2) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][u], w[i][u], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][u + 1], w[i][u + 1], acc1, 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[khalf][msub][kq * 4 + r][li] = acc0[r] + acc1[r];
    __syncthreads();
    if (ethread) {
        const int ms = bl >> 4, row = bl & 15;
        float pre[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) pre[q] = gpre[q] + (red[0][ms][row][q * 4 + ul] + red[1][ms][row][q * 4 + ul]);
        if (step) {
            const float ig = sigmoidf_(pre[0]), fg = sigmoidf_(pre[1]), gg = tanhf_(pre[2]), og = sigmoidf_(pre[3]);
            const long long o = (long long)eb * D + eu;
            const float cn = fg * cprev + ig * gg;
            const float hn = og * tanhf_(cn);
            R.c[o] = cn;
            R.hnext[hfrag_index(eb, eu, D)] = hn;
            if (R.yout || R.yout_elu) {
                const float yv = hn + skipv;
                if (R.yout) R.yout[(long long)eb * R.y_bs + eu] = yv;
                if (R.yout_elu) R.yout_elu[(long long)eb * R.y_bs + eu] = elu1(yv);
            }
        } else {
            float* g = R.gout + (long long)eb * (4 * D);
#pragma unroll
            for (int q = 0; q < 4; ++q) g[q * D + eu] = pre[q];
        }
    }
}

}  // namespace ac
