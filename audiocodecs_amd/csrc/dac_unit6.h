// dac_unit6: DAC's ResidualUnit ([HF] dac :103-133)
//     y = x + conv_1x1(Snake_2(conv_k7_dilated(Snake_1(x))))
// as ONE kernel for the 96-channel units (the decoder's last block: the time-longest layers of the path), round 4.
// As two tap-GEMM launches a unit moves six tensors through HBM (Snake_1(x) in, the hidden activation out and in again, x in, y and
// Snake_next(y) out); the 1 x 1 conv is 25 % of DAC's step and runs at 3.0 - 4.4 TB/s of HBM.  Here the hidden activation never
// leaves the CU.  MEASURED (32 clips x 10 s, same process, ac_debug_set "dac_unit"): 14.5 -> 13.9 ms per 96-channel unit (-4 %), i.e.
// 0.7 % of DAC's step -- the unit is not bound by those bytes but by instruction issue: two Snakes per output element (~22 vector
// instructions each, the mid one and the consumer's) are as many VALU cycles as the k7 conv's MFMAs take, and one kernel serialises
// per wave what two launches overlapped across workgroups.  The 64-channel units lose 26 % in this form (32 x 64 wave tiles at two
// workgroups per CU against 64 x 32 at three) and keep two launches.  What makes the form cheap at all for narrow units: in the
// 4 x 1 wave arrangement of tap_gemm6 (four waves along time, each 32 rows x all C channels) a wave owns every channel of its rows,
// so the second product -- a contraction over channels -- is wave-local: no exchange between waves, no barrier but the one that
// retires the main loop's slabs.
//   1. tap6_mainloop<4, 1, 1, WN, HALO, SWAP = true>: the k7 conv exactly as tap_gemm6 runs it (same slab, same stages, same
//      products in the same order), with the MFMA's operands exchanged: the accumulators hold the TRANSPOSED 32 x 32 tiles, a lane
//      owns ONE time row and register r the channel 8 (r / 4) + 4 kh + r % 4 -- four consecutive channels per register quad;
//   2. hidden = Snake_2(acc 2^-s + b7) in registers; its split16 scale is the WAVE tile's own largest magnitude (two shuffles
//      short of a butterfly; the two-launch path uses the clip's); hi / lo planes -> the wave's private LDS region [32 rows][C + 8]
//      with 8-byte stores, which is the A-operand layout of the second product;
//   3. acc2 = hidden x W1 (v_mfma_f32_32x32x16_f16, 3 partial products, weight fragments L2 -> registers one k-step ahead);
//   4. tap6_epilogue (the direct form, shared source): 2^-s, bias, + x, amax, store y, Snake_next, amax, store.
// Wider units (128 .. 1536 channels) keep two launches: their rows span several waves, the second product would need every wave's
// share of the hidden activation -- the accumulators twice or 66 - 131 KB of LDS per 128-row tile (profiles/r4_variants.md).
// The result differs from the two-launch path by the rounding of the hidden activation's lo plane (a different power-of-two
// scale); both are fp32-grade (split16.h) and meet the same parity bar (tests/test_dac_gpu_parity.py, test_round4_kernels_gpu.py).
#pragma once
#include "tap_gemm6.h"

namespace ac {

struct DacUnitParams {
    TapGemmParams g;            // the k7 conv as run_tap prepares it: seg[0] = Snake_1(x) with its amax, bias / winv of the k7 conv;
                                // res* = x (raw), alpha* = the Snake of whatever consumes y, y / y_elu / amax_out = the unit's outputs
    const float* a_mid;         // Snake_2: alpha, (alpha + 1e-9)^-1  [C]
    const float* a_mid_inv;
    const float* bias2;         // 1 x 1 conv: bias [C], 2^-s of its rows [C]
    const float* winv2;
};

template <int WN>
struct DacUnitCfg {
    static constexpr int C = 32 * WN, KP = C + 8, PLANE = 32 * KP;      // a wave's region: [2 planes][32 rows][KP] fp16
    static constexpr size_t region_bytes = (size_t)4 * 2 * PLANE * 2;
    template <int HALO>
    static constexpr size_t lds_bytes() {
        return Tap6Cfg<4, 1, 1, WN, HALO>::lds_for(2) > region_bytes ? Tap6Cfg<4, 1, 1, WN, HALO>::lds_for(2) : region_bytes;
    }
};

template <int WN, int HALO>
__global__ __launch_bounds__(256, 2) void dac_unit6_kernel(const DacUnitParams q, const __bf16* __restrict__ wp1, const __bf16* __restrict__ wp2) {
    using UC = DacUnitCfg<WN>;
    constexpr int C = UC::C, KP = UC::KP, PLANE = UC::PLANE, KS2 = C / 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const TapGemmParams& p = q.g;
    f32x16 acc[1][WN];
    Tap6Tile tl;
    tap6_mainloop<4, 1, 1, WN, HALO, true>(p, wp1, smem, acc, tl);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i32 = lane & 31, kh = lane >> 5;
    const bool rin = tl.m0 + wave * 32 + i32 < p.M;            // this lane's time row exists

    // ---- 2. hidden activation in registers (transposed tiles: register 4 g + e <-> channel 32 c + 8 g + 4 kh + e)
    unsigned am = 0;
#pragma unroll
    for (int c = 0; c < WN; ++c)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = tl.n0 + 32 * c + 8 * g + 4 * kh;
            const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + n);
            const f32x4 wv = *reinterpret_cast<const f32x4*>(p.winv + n);
            const f32x4 al = *reinterpret_cast<const f32x4*>(q.a_mid + n);
            const f32x4 ai = *reinterpret_cast<const f32x4*>(q.a_mid_inv + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = __fmaf_rn(acc[0][c][4 * g + e], tl.a_inv * wv[e], bv[e]);
                const float w = rin ? snake1(v, al[e], ai[e]) : 0.f;
                acc[0][c][4 * g + e] = w;
                amax_acc(am, w);
            }
        }
#pragma unroll
    for (int sh = 1; sh < 64; sh <<= 1) {
        const unsigned t = (unsigned)__shfl_xor((int)am, sh);
        am = t > am ? t : am;
    }
    const int eh = s16_exponent(am);
    const float sh_ = s16_pow2(eh), ih = s16_pow2(-eh);
    __syncthreads();                                            // every wave is done with the main loop's slabs
    _Float16* reg = reinterpret_cast<_Float16*>(smem) + wave * 2 * PLANE;
#pragma unroll
    for (int c = 0; c < WN; ++c)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            split16_store4s(f32x4{acc[0][c][4 * g], acc[0][c][4 * g + 1], acc[0][c][4 * g + 2], acc[0][c][4 * g + 3]}, sh_, reg, PLANE,
                            i32 * KP + 32 * c + 8 * g + 4 * kh);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // (the wave reads back its own rows only: no barrier)

    // ---- 3. acc2 = hidden x W1: tiles in the usual orientation (register = row, lane = channel) for the shared epilogue
    f32x16 acc2[1][WN];
#pragma unroll
    for (int c = 0; c < WN; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[0][c][r] = 0.f;
    const __bf16* wb = wp2 + lane * 8;                          // image [column tile][k-step][plane][64 lanes][8]
    auto load_b = [&](int ks, bf16x8 (&bf)[WN][2]) {
#pragma unroll
        for (int c = 0; c < WN; ++c)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) bf[c][pl] = *reinterpret_cast<const bf16x8*>(wb + (((long long)c * KS2 + ks) * 2 + pl) * 512);
    };
    auto step = [&](int ks, const bf16x8 (&bf)[WN][2]) {
        const f16x8 ah = *reinterpret_cast<const f16x8*>(reg + i32 * KP + ks * 16 + 8 * kh);
        const f16x8 al = *reinterpret_cast<const f16x8*>(reg + PLANE + i32 * KP + ks * 16 + 8 * kh);
#pragma unroll
        for (int c = 0; c < WN; ++c) {
            f32x16 v = acc2[0][c];
            v = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, __builtin_bit_cast(f16x8, bf[c][0]), v, 0, 0, 0);
            v = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, __builtin_bit_cast(f16x8, bf[c][1]), v, 0, 0, 0);
            acc2[0][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, __builtin_bit_cast(f16x8, bf[c][0]), v, 0, 0, 0);
        }
    };
    bf16x8 b0[WN][2], b1[WN][2];
    load_b(0, b0);
#pragma unroll
    for (int ks = 0; ks < KS2; ks += 2) {
        load_b(ks + 1, b1);
        step(ks, b0);
        if (ks + 2 < KS2) load_b(ks + 2, b0);
        step(ks + 1, b1);
    }

    // ---- 4. the unit's output through the tap-GEMM's own (direct) epilogue: 2^-s, bias, + x, amax, y, Snake_next, amax
    TapGemmParams p2 = p;
    p2.bias = q.bias2;
    p2.winv = q.winv2;
    p2.epi_direct = 1;
    tap6_epilogue<4, 1, 1, WN, HALO>(p2, acc2, smem, tl.b, tl.m0, tl.n0, ih, false, tl.clk_t0, tl.clk_r0, false);
}

}  // namespace ac
