// thin_conv6: the two 64-wide layers next to the 32-channel stage of the SEANet stack, where K is only 128:
//   * the stride-2 down-sampling conv  Conv1d(32, 64, k = 4, s = 2)   ([HF] EncodecEncoder.layers[3]),
//   * the stride-2 transposed conv     ConvTranspose1d(64, 32, k = 4, s = 2) ([HF] EncodecDecoder, last up-sampler),
// both of which are, on channels-last data viewed as SUPER-ROWS of 64 floats (two 32-channel time steps, or one
// 64-channel time step),
//       out[m][0..63] = W[64][128] * [S[m-1] | S[m]] + b           (tap_gemm.h's mapping, K order = memory order).
// As tap-GEMMs they are 4 K-stages long and spend their time in prologue / epilogue (3.2 TB/s, HBM-bound layers);
// here the weights (split offline into three bf16 terms, rb_fused6.h fragment order) stay in registers, persistent
// workgroups walk (clip, 64-row tile), the input tile is split once into LDS planes and the transposed MFMA tile
// puts 4 consecutive output channels in a lane: 16-byte stores straight to HBM.  Split-operand arithmetic as in
// tap_gemm6.h (6 of 9 partial products on v_mfma_f32_16x16x32_bf16, fp32 accumulate).
// Edge rule for S[-1]: zeros (transposed conv: x[-1] = 0) or the reflected pair [x[2] | x[1]] of 32-wide rows
// (causal reflect padding of the strided conv, [HF]:157-176).  Input rows are read as stored (the producers write
// the activated flavour), outputs raw and / or ELU'd.
#pragma once
#include <hip/hip_runtime.h>
#include "rb_fused6.h"

namespace ac {

struct ThinConv6Params {
    const float* x;         // [B][Ls][64]
    const __bf16* wf;       // [4 n-tiles of 16][4 k-steps of 32][plane 3][lane 64][8 bf16]
    const float* bias;      // [64]
    float* y;               // optional raw output [B][M][64]
    float* y_elu;           // optional ELU'd output
    int B, M, Ls;           // outputs / input super-rows per clip
    int ntiles;             // 64-row tiles per clip
    int edge;               // S[-1]: 0 zeros, 1 reflected pair of 32-wide rows
    // split16.h (NP = 2): amax slot [B] of x, optional slot of the output, per-output-channel 2^-s of the weight rows
    const unsigned* amax_in;
    unsigned* amax_out;
    const float* winv;      // [64]
};

constexpr int T6_BM = 64, T6_XP = 72;                       // rows per tile; LDS pitch in bf16 (64 + 8)
constexpr int T6_ROWS = T6_BM + 1;
constexpr int T6_PLANE = T6_ROWS * T6_XP;
constexpr int T6_SLOTS = (T6_ROWS * 16 + 255) / 256;        // float4 slots per thread
constexpr size_t T6_LDS = (size_t)3 * T6_PLANE * 2;

template <int NP = 2>
__global__ __launch_bounds__(256, 4) void thin_conv6_kernel(const ThinConv6Params p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __bf16* Xs = reinterpret_cast<__bf16*>(smem);           // [3][T6_ROWS][T6_XP]: slab row r = super-row m0 - 1 + r
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave = 16-channel tile of the output
    const int li = lane & 15, kq = lane >> 4;
    const int total = p.B * p.ntiles;

    constexpr int WPL = NP == 2 ? 2 : 3;                    // planes in the weight image
    bf16x8 wr[4][3];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
            wr[ks][pl] = *reinterpret_cast<const bf16x8*>(p.wf + ((((long long)wave * 4 + ks) * WPL + pl) * 64 + lane) * 8);
    const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + wave * 16 + 4 * kq);
    f32x4 wiv = {1.f, 1.f, 1.f, 1.f};
    if (NP == 2) wiv = *reinterpret_cast<const f32x4*>(p.winv + wave * 16 + 4 * kq);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    int s_row[T6_SLOTS], s_q[T6_SLOTS];
#pragma unroll
    for (int i = 0; i < T6_SLOTS; ++i) {
        const int e = tid + i * 256;
        s_row[i] = e >> 4;
        s_q[i] = e & 15;
    }
    const int in_bytes = p.Ls * 256, out_bytes = p.M * 256;
    f32x4 rx[T6_SLOTS];
    auto load_tile = [&](int tile) {
        const int b = tile / p.ntiles, m0 = (tile % p.ntiles) * T6_BM;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (long long)b * p.Ls * 64), 0, in_bytes, 0x00020000);
#pragma unroll
        for (int i = 0; i < T6_SLOTS; ++i) {
            const int sr = m0 - 1 + s_row[i];
            int off = sr * 256 + s_q[i] * 16;
            if (sr < 0) off = p.edge ? (s_q[i] < 8 ? 256 + s_q[i] * 16 : 128 + (s_q[i] - 8) * 16) : 0x7fff0000;   // [x[2] | x[1]] or zeros
            if (s_row[i] >= T6_ROWS) off = 0x7fff0000;
            rx[i] = bufload16(rs, off, 0);                  // super-rows past the clip: out of range -> 0
        }
    };
    float sx = 1.f, ix = 1.f;                               // split16 scale of the clip staged in LDS
    auto store_tile = [&](int tile) {
        if (NP == 2) {
            const int se = s16_exponent(*amax_at(p.amax_in, tile / p.ntiles));
            sx = s16_pow2(se);
            ix = s16_pow2(-se);
        }
#pragma unroll
        for (int i = 0; i < T6_SLOTS; ++i)
            if (s_row[i] < T6_ROWS) split_store4<NP>(rx[i], Xs, T6_PLANE, s_row[i] * T6_XP + 4 * s_q[i], sx);
    };

    int tile = blockIdx.x;
    if (tile >= total) return;
    load_tile(tile);
    store_tile(tile);
    __syncthreads();
    unsigned omax = 0;
    int omax_b = tile / p.ntiles;
    for (; tile < total; tile += gridDim.x) {
        const int next = tile + gridDim.x;
        const f32x4 iv = wiv * ix;                          // this tile's inverse scales (store_tile(next) replaces ix)
        if (next < total) load_tile(next);
        f32x4 acc[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[a] = NP == 2 ? zero4 : bv;
        // k-steps 0, 1: S[m-1] (slab row r), k-steps 2, 3: S[m] (slab row r + 1)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 xf[4][3];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int pl = 0; pl < NP; ++pl)
                    xf[a][pl] = *reinterpret_cast<const bf16x8*>(Xs + pl * T6_PLANE + (a * 16 + li + (ks >> 1)) * T6_XP + (ks & 1) * 32 + 8 * kq);
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[a] = mma6<NP>(wr[ks], xf[a], acc[a]);
        }
        lds_barrier();                                      // every wave is done reading the slab
        if (next < total) store_tile(next);                 // staged before the output stores are issued (rb_fused6.h)
        {
            const int b = tile / p.ntiles, m0 = (tile % p.ntiles) * T6_BM;
            if (p.amax_out && b != omax_b) {
                amax_flush(omax, amax_at(p.amax_out, omax_b));
                omax = 0;
                omax_b = b;
            }
            const long long ob = (long long)b * p.M * 64;
            const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y ? p.y + ob : nullptr), 0, p.y ? out_bytes : 0, 0x00020000);
            const __amdgpu_buffer_rsrc_t re = __builtin_amdgcn_make_buffer_rsrc((void*)(p.y_elu ? p.y_elu + ob : nullptr), 0, p.y_elu ? out_bytes : 0, 0x00020000);
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int m = m0 + a * 16 + li;
                const int o = (m < p.M ? m * 256 : 0x7fff0000) + (wave * 16 + 4 * kq) * 4;
                f32x4 v = acc[a];
                if (NP == 2) v = f32x4{__fmaf_rn(v.x, iv.x, bv.x), __fmaf_rn(v.y, iv.y, bv.y), __fmaf_rn(v.z, iv.z, bv.z), __fmaf_rn(v.w, iv.w, bv.w)};
                if (p.amax_out && m < p.M) amax_acc4(omax, v);
                if (p.y) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), ry, o, 0, 0);
                if (p.y_elu) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, elu4(v)), re, o, 0, 0);
            }
        }
        lds_barrier();  
    }
    if (p.amax_out) amax_flush(omax, amax_at(p.amax_out, omax_b));
}

}  // namespace ac
