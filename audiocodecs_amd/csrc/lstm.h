// LSTM recurrence of [HF] EncodecLSTM (:236-249): nn.LSTM(D, D, num_layers) over [T,B,D], zero
// initial state, gate order i,f,g,o, followed by the module's skip connection (out + in).
//
// The input projection x_t * W_ih^T + (b_ih + b_hh) for ALL t is one tap_gemm (the 1-tap case)
// written as gin[t][b][4D].  What is left is inherently sequential in t: one launch per time step
// (a dependent kernel boundary costs ~1.5 us on MI355X -- cheaper than an in-kernel grid barrier).
//
// Decomposition per step: workgroup = 32 clips x 4 hidden units (= 16 gate columns: 4 gates x 4
// units, so the cell update is local to the workgroup).  4 waves = 2 clip sub-tiles x 2 K-halves;
// each wave runs v_mfma_f32_16x16x4_f32 over its half of K = D with the A operand (h_{t-1}) and
// the B operand (W_hh slice, pre-packed in fragment order) loaded straight from L2 into registers
// as 16-byte vectors; the two K-halves meet in LDS, then 128 threads finish the cell.
#pragma once
#include <hip/hip_runtime.h>
#include "tap_gemm.h"

namespace ac {

struct LstmStepParams {
    const float* gin;    // [B][4D] slice for this t (row b at gin + b*gin_bs)
    long long gin_bs;
    const float* hprev;  // [B][D] or nullptr (t == 0: h = 0)
    float* hnext;        // [B][D]
    float* c;            // [B][D] cell state (zeroed before t == 0)
    const float* wpk;    // packed W_hh: [D/4 unit groups][D/16 ksteps][64 lanes][4]
    const float* skip;   // optional: module input x[b][t][:] (row b at skip + b*skip_bs)
    float* yout;         // optional: yout[b][:] = h + skip (row b at yout + b*y_bs)
    float* yout_elu;     // optional: ELU(h + skip), same layout (the consumer conv starts with nn.ELU)
    long long skip_bs, y_bs;
    int B, D, first;
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ __launch_bounds__(256) void lstm_step_kernel(const LstmStepParams p) {
    __shared__ float red[2][2][16][17];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int msub = wave & 1, khalf = wave >> 1;
    const int li = lane & 15, kq = lane >> 4;
    const int ug = blockIdx.x;
    const int b0 = blockIdx.y * 32;
    const int D = p.D;

    f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    if (!p.first) {
        const int ksteps = D / 32;                 // per K-half
        const int brow = b0 + msub * 16 + li;
        const bool valid = brow < p.B;
        const float* hrow = p.hprev + (long long)(valid ? brow : 0) * D + khalf * (D / 2) + 4 * kq;
        const float* wrow = p.wpk + ((long long)ug * (D / 16) + (long long)khalf * ksteps) * 256 + lane * 4;
        for (int k0 = 0; k0 < ksteps; k0 += 4) {
            f32x4 a[4], w[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool in = k0 + i < ksteps;
                a[i] = (valid && in) ? *reinterpret_cast<const f32x4*>(hrow + (k0 + i) * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
                w[i] = in ? *reinterpret_cast<const f32x4*>(wrow + (long long)(k0 + i) * 256) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
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
    if (tid < 128) {
        const int bl = tid >> 2, ul = tid & 3;
        const int b = b0 + bl;
        if (b < p.B) {
            const int ms = bl >> 4, row = bl & 15;
            const int u = ug * 4 + ul;
            const float* g = p.gin + (long long)b * p.gin_bs;
            float pre[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                pre[q] = g[q * D + u] + (red[0][ms][row][q * 4 + ul] + red[1][ms][row][q * 4 + ul]);
            const float ig = sigmoidf_(pre[0]), fg = sigmoidf_(pre[1]), gg = tanhf(pre[2]), og = sigmoidf_(pre[3]);
            const long long o = (long long)b * D + u;
            const float cprev = p.first ? 0.f : p.c[o];
            const float cn = fg * cprev + ig * gg;
            const float hn = og * tanhf(cn);
            p.c[o] = cn;
            p.hnext[o] = hn;
            if (p.yout || p.yout_elu) {
                const float yv = hn + p.skip[(long long)b * p.skip_bs + u];
                if (p.yout) p.yout[(long long)b * p.y_bs + u] = yv;
                if (p.yout_elu) p.yout_elu[(long long)b * p.y_bs + u] = elu1(yv);
            }
        }
    }
}

}  // namespace ac
