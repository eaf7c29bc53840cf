// The launch parameters and the per-clip scales the fused residual-block kernels share (rb_fused6.h, rb_fused6_128.h, rb_stream6.h).
#pragma once
#include <hip/hip_runtime.h>
#include "tap_gemm4.h"
#include "split16.h"

namespace ac {

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

struct RbFused6Params {
    const float* xr;        // [B][L][C] raw input
    const __bf16* w3f;      // fragment-packed k3 conv
    const __bf16* wff;      // fragment-packed [1x1 over the hidden | shortcut over x]
    const float* b3;        // [C/2]
    const float* bf;        // [C]  (b_1x1 + b_shortcut)
    float* y;               // optional raw output [B][L][C]
    float* y_elu;           // optional ELU'd output
    int B, L, Lp;           // Lp: reflect base length (L, or 3 when L <= 2: [HF]:148-155)
    int ntiles;             // tiles per clip
    int nseg, seg_rows;     // rb_stream6.h: segments per clip, rows per segment (a multiple of 16)
    int pad;                // PAD_REFLECT (EnCodec) / PAD_ZERO (Mimi)
    int lpad;               // rows of left padding of the k3 conv: 2 = causal (EnCodec, Mimi), 1 = centred (WavTokenizer's non-causal SEANet)
    int dbg;                // developer timing modes (AC_RB6_DBG): 1 no stage-A MFMAs, 2 no stage-B MFMAs, 4 no staging,
                            // 8 no output stores, 16 no loads -- results are wrong in every mode but 0
    // split16.h (NP = 2): amax slot [B] of x, optional slot of the output, per-output-channel 2^-s of the two weight
    // matrices, and the bound of the hidden activation |ELU(conv_k3(ELU(x)) + b3)| <= hb0 + hb1 * amax(x)
    // (hb0 = max |b3|, hb1 = largest row 1-norm of the k3 weights) from which its scale is taken
    const unsigned* amax_in;
    unsigned* amax_out;
    const float* winv3;     // [C/2]
    const float* winvf;     // [C]
    float hb0, hb1;
    // HEAD (round 4; C = 64, identity shortcut: Mimi's last block): the block's ELU'd output does not go to HBM -- the decoder's
    // final Conv1d(C, 1, k) (causal, zero-padded) is applied to the tile in LDS and one float per sample is stored.  Tiles then
    // advance by BM - (k - 1) rows and start k - 1 rows early (the conv's left context is recomputed, 3 % of the block's work);
    // the conv's arithmetic is head4_kernel's (thin.h), in the same order: bit-identical samples.  Saves the 2 x 7.9 GB round
    // trip of Mimi's widest tensor (128 clips x 10 s).
    const float* head_w;    // [k][C] tap-major, null = off
    const float* head_b;    // [1]
    float* head_y;          // [B][L]
    int head_k;
};

// split16.h scales of one clip of a fused block: sx for ELU(x) in the k3 conv; sb for the hidden activation AND the raw x of
// the second stage (they share an accumulator, so they share a scale)
struct Rb16Scale {
    float sx, sb, ix, ib;   // 2^s and 2^-s
};
__device__ __forceinline__ Rb16Scale rb16_scale(unsigned amax_x, float hb0, float hb1) {
    const int ex = s16_exponent(amax_x);
    const float hb = __fmaf_rn(hb1, __uint_as_float(amax_x), hb0) * 1.0000005f;
    const int eh = s16_exponent(__float_as_uint(hb));
    const int eb = eh < ex ? eh : ex;
    return Rb16Scale{s16_pow2(ex), s16_pow2(eb), s16_pow2(-ex), s16_pow2(-eb)};
}

}  // namespace ac
