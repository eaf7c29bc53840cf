// enc_front: the thin-channel head of the EnCodec encoder as ONE kernel
//     sig [B][T]  ->  Conv1d(1, 32, k7)  ->  ResnetBlock(32)  ->  ELU  ->  Conv1d(32, 64, k4, s2)  ->  y [B][ceil(T/2)][64]
// ([HF] modeling_encodec.py:290-301 EncodecEncoder.layers[0..3], called from audiocodecs/encodec.py:90).  As separate kernels
// (stem_kernel, rb_fused6_kernel<32>, thin_conv6_kernel) these layers each move their algorithmic bytes and nothing more, but
// the layer boundaries are 3.9 GB round trips at 64 x 10 s: 11.9 GB of HBM traffic for 2 GB of compulsory output.
//
// Structure: one WAVE = one stream.  A wave walks a segment of one clip in chunks of 32 input samples and carries every
// intermediate in its own LDS slab; no workgroup barrier in the loop, no cross-wave traffic:
//   stem      x0 = b0 + W0 * sig[t-6 .. t]                       fp32 FMAs (bias first, taps ascending: stem_kernel's order);
//             ELU(x0) and x0 are split (split16.h) into the slabs Xe (34 rows: 2 rows of causal halo carried over) and Xr
//   stage A   h  = ELU(W3 * Xe[t-2 .. t] + b3)                   3 k-steps, v_mfma_f32_16x16x32_f16, 3 partial products
//   stage B   y1 = [W1 | Ws] * [h | x0] + (b1 + bs)              2 k-steps; ELU(y1) split into the slab Y1e (aliases Xe)
//   stage C   y  = Wd * Y1e[2m-2 .. 2m+1] + bd                   4 k-steps, 16 output rows per chunk -> HBM (raw) + amax
// The halo rows of Xe and Y1e are the last two rows of the previous chunk (kept in the slab), so nothing is recomputed
// inside a stream; a segment that starts inside a clip runs ONE warm-up chunk without output.
// Clip edges follow [HF]:157-176: reflect padding on the left of every conv (the sample window for the stem, row copies for
// Xe and Y1e), one reflected step on the right of the strided conv when T is odd, length mask on the samples.
// The weights of stages B and C (rb_fused6.h / thin_conv6.h fragment images) live in registers (160 VGPRs), stage A's are read from
// LDS per chunk; two waves per SIMD.
//
// split16 scales: the intermediates never exist as whole tensors, so their amax cannot be known; each takes a BOUND
// derived from amax(sig) through the layers (|conv(x)| <= max|b| + max ||w||_1 amax(x), ELU never grows a magnitude):
//     A0 = sb0 + sb1 amax(sig);   H = hb0 + hb1 A0;   Y1 = fb0 + fb1h H + fb1x A0
// A bound that is 2^w too large costs w of the 16 bits of headroom split16 has below the largest element before relative
// precision degrades (split16.h); these chains measure 2^5 .. 2^7 (ac_debug_bounds, tests/test_split16_gpu.py).
// The output's amax is exact (accumulated in registers, one atomic per stream).
#pragma once
#include <hip/hip_runtime.h>
#include "rb_fused6.h"
#include "thin.h"

namespace ac {

struct EncFrontParams {
    const float* sig;        // [B][T]
    const float* rel_len;    // optional [B]
    const float* w0;         // stem weights [32][7]
    const float* b0;         // [32]
    const __bf16* w3f;       // rb_fused6.h image of the k3 conv:        [1 n-tile][3 k-steps][2 planes][64 lanes][8 fp16]
    const __bf16* wff;       // ... of [1x1 over hidden (16 + 16 zero columns) | shortcut over x]: [2][2][2][64][8]
    const __bf16* wdf;       // thin_conv6.h image of the strided conv:  [4][4][2][64][8]
    const float* b3;         // [16]
    const float* winv3;      // [16] 2^-s of the image rows
    const float* bf;         // [32]
    const float* winvf;      // [32]
    const float* bd;         // [64]
    const float* winvd;      // [64]
    float* y;                // [B][M][64] raw
    float* dbg_x0;           // test hook: optional raw stem output [B][T][32]
    float* dbg_y1;           // test hook: optional raw block output [B][T][32]
    int B, T, M;             // M = ceil(T / 2)
    int seg_chunks;          // chunks (of 32 samples) per stream
    int segs_per_clip;
    const unsigned* amax_sig;   // split16.h slot [B]: largest finite |sig|
    unsigned* amax_out;         // optional slot of y
    float sb0, sb1;          // |x0| <= sb0 + sb1 amax(sig)
    float hb0, hb1;          // |h|  <= hb0 + hb1 bound(x0)
    float fb0, fb1h, fb1x;   // |y1| <= fb0 + fb1h bound(h) + fb1x bound(x0)
};

constexpr int EF_ROWS = 32;                                                    // input samples per chunk
constexpr int EF_XP = 40, EF_R1_ROWS = 36, EF_R1_PLANE = EF_R1_ROWS * EF_XP;    // Xe rows 2..35 / Y1e rows 0..33, 80-byte rows
constexpr int EF_QP = 48, EF_Q_PLANE = EF_ROWS * EF_QP;                         // Xr, Hs: 96-byte rows (conflict-free 16-byte reads)
constexpr int EF_SIGW = 40;                                                    // sample window of a chunk: t0 - 8 .. t0 + 31
constexpr int EF_WAVE_BYTES = (2 * EF_R1_PLANE + 4 * EF_Q_PLANE) * 2 + EF_SIGW * 4;
constexpr int EF_CONST_FLOATS = 512;                                           // stem weights, biases, inverse weight scales
constexpr int EF_W3_HALFS = 3 * 2 * 512;                                       // the k3 conv's fragment image, shared by the waves (6 KB)
constexpr int EF_SHARED_BYTES = EF_CONST_FLOATS * 4 + EF_W3_HALFS * 2;
constexpr int EF_WAVES = 8;                                                     // one 8-wave workgroup per CU (see the note at the kernel)
constexpr size_t EF_LDS = (size_t)EF_SHARED_BYTES + EF_WAVES * (size_t)EF_WAVE_BYTES;
static_assert(EF_LDS <= 160 * 1024, "one workgroup per CU");
static_assert(EF_WAVE_BYTES % 16 == 0, "wave slabs stay 16-byte aligned");
// offsets (floats) inside the constant block
constexpr int EF_W0 = 0, EF_B0 = 224, EF_B3 = 256, EF_I3 = 272, EF_BF = 288, EF_IF = 320, EF_BD = 352, EF_ID = 416;

__device__ __forceinline__ f32x4 ef_mfma(const f16x8 a, const f16x8 b, const f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 ef_fma4(const f32x4 v, const f32x4 s, const f32x4 b) {
    return f32x4{__fmaf_rn(v.x, s.x, b.x), __fmaf_rn(v.y, s.y, b.y), __fmaf_rn(v.z, s.z, b.z), __fmaf_rn(v.w, s.w, b.w)};
}
// scale exponent of a bound (a positive float, or inf when it overflowed)
__device__ __forceinline__ int ef_exp(float bound) { return s16_exponent(__float_as_uint(bound) & 0x7fffffffu); }

__global__ __launch_bounds__(64 * EF_WAVES, 2) void enc_front_kernel(const EncFrontParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;

    // ---- constants every wave reads per chunk -> LDS (the only workgroup-wide step)
    for (int e = tid; e < 480; e += 64 * EF_WAVES) {
        float v;
        if (e < 224) v = p.w0[(e & 31) * 7 + (e >> 5)];          // [tap][channel]
        else if (e < 256) v = p.b0[e - 224];
        else if (e < 272) v = p.b3[e - 256];
        else if (e < 288) v = p.winv3[e - 272];
        else if (e < 320) v = p.bf[e - 288];
        else if (e < 352) v = p.winvf[e - 320];
        else if (e < 416) v = p.bd[e - 352];
        else v = p.winvd[e - 416];
        smem[e] = v;
    }
    // the k3 conv's weights are read from LDS per chunk (6 fragment reads): 24 registers fewer than holding them, which is what
    // keeps the kernel spill-free at two waves per SIMD
    const _Float16* W3s = reinterpret_cast<const _Float16*>(smem + EF_CONST_FLOATS);
    for (int e = tid; e < EF_W3_HALFS / 8; e += 64 * EF_WAVES)
        *reinterpret_cast<f16x8*>(reinterpret_cast<_Float16*>(smem + EF_CONST_FLOATS) + e * 8) = *reinterpret_cast<const f16x8*>(p.w3f + (long long)e * 8);
    __syncthreads();

    _Float16* R1 = reinterpret_cast<_Float16*>(reinterpret_cast<char*>(smem) + EF_SHARED_BYTES + wave * EF_WAVE_BYTES);
    _Float16* Xr = R1 + 2 * EF_R1_PLANE;
    _Float16* Hs = Xr + 2 * EF_Q_PLANE;
    float* sg = reinterpret_cast<float*>(Hs + 2 * EF_Q_PLANE);

    // ---- this wave's stream: clip b, chunks [c_first, c_last)
    const int sid = blockIdx.x * EF_WAVES + wave;
    if (sid >= p.B * p.segs_per_clip) return;
    const int b = sid / p.segs_per_clip, seg = sid - b * p.segs_per_clip;
    const int nchunks = (p.T + EF_ROWS - 1) / EF_ROWS;
    const int c_first = seg * p.seg_chunks;
    const int c_last = c_first + p.seg_chunks < nchunks ? c_first + p.seg_chunks : nchunks;
    if (c_first >= c_last) return;

    // ---- weights -> registers (once)
    f16x8 wf[2][2][2], wd[4][4][2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) wf[ks][c][pl] = *reinterpret_cast<const f16x8*>(p.wff + ((long long)((c * 2 + ks) * 2 + pl) * 64 + lane) * 8);
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) wd[ks][c][pl] = *reinterpret_cast<const f16x8*>(p.wdf + ((long long)((c * 4 + ks) * 2 + pl) * 64 + lane) * 8);

    // hidden columns 16..31 are K padding of stage B's first k-step (the matching weight columns are zero): zero once
    {
        const f16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = lane + 64 * i;
            *reinterpret_cast<f16x8*>(Hs + (idx >> 6) * EF_Q_PLANE + ((idx >> 1) & 31) * EF_QP + 16 + 8 * (idx & 1)) = z8;
        }
    }

    // ---- split16 scales of this clip's intermediates (bounds, see the header)
    const float a_sig = __uint_as_float(*amax_at(p.amax_sig, b));
    const float A0 = __fmaf_rn(p.sb1, a_sig, p.sb0) * 1.0000005f;
    const float Hb = __fmaf_rn(p.hb1, A0, p.hb0) * 1.0000005f;
    const float Yb = __fmaf_rn(p.fb1h, Hb, __fmaf_rn(p.fb1x, A0, p.fb0)) * 1.000001f;
    const int ex = ef_exp(A0), eh = ef_exp(Hb), eb = eh < ex ? eh : ex, ey = ef_exp(Yb);
    const float sx = s16_pow2(ex), ix = s16_pow2(-ex);       // ELU(x0) in the k3 conv
    const float sb = s16_pow2(eb), ib = s16_pow2(-eb);       // hidden and raw x0 share stage B's accumulator
    const float sy = s16_pow2(ey), iy = s16_pow2(-ey);       // ELU(y1) in the strided conv

    const float* sigb = p.sig + (long long)b * p.T;
    float alen = 3.0e38f;
    if (p.rel_len) alen = (float)p.T * p.rel_len[b];
    auto fetch_sig = [&](int t0) -> float {                  // lane i holds sample t0 - 8 + i (reflected on the left, masked)
        const int q = t0 - 8 + lane;
        const int j = q < 0 ? -q : q;
        return (lane < EF_SIGW && j < p.T && (float)j < alen) ? sigb[j] : 0.f;
    };
    auto copy_row = [&](_Float16* base, int plane, int pitch, int dst, int src, int l16) {   // 32 halfs of both planes, lanes l16 = 0..15
        const int pl = l16 >> 3, c4 = (l16 & 7) * 4;
        const f16x4_t v = *reinterpret_cast<const f16x4_t*>(base + pl * plane + src * pitch + c4);
        *reinterpret_cast<f16x4_t*>(base + pl * plane + dst * pitch + c4) = v;
    };

    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    unsigned omax = 0;
    const long long yb_off = (long long)b * p.M * 64;
    int ch = c_first > 0 ? c_first - 1 : 0;                  // a segment inside a clip warms its slabs up on the chunk before
    float sv = fetch_sig(ch * EF_ROWS);
    for (; ch < c_last; ++ch) {
        const int t0 = ch * EF_ROWS;
        const bool emit = ch >= c_first;
        // Halos = the previous chunk's last two rows: Y1e rows 32, 33 -> 0, 1 and Xe rows 34, 35 -> 2, 3 (Y1e, written over
        // Xe rows 2..33, leaves them alone).  Stale bytes in the first chunk of a stream: fixed below at a clip start, and in a
        // warm-up chunk they reach only rows whose results nothing reads.
        if (lane < 32) copy_row(R1, EF_R1_PLANE, EF_XP, lane >> 4, 32 + (lane >> 4), lane & 15);
        else copy_row(R1, EF_R1_PLANE, EF_XP, 2 + ((lane - 32) >> 4), 34 + ((lane - 32) >> 4), lane & 15);
        if (lane < EF_SIGW) sg[lane] = sv;
        if (ch + 1 < c_last) sv = fetch_sig(t0 + EF_ROWS);   // in flight during the chunk

        // ---- stem: 32 rows (times t0 .. t0 + 31 -> Xe rows 4..35, Xr rows 0..31), lane item = (row, 4 channels)
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int e = lane + 64 * it;
            const int r = e >> 3, g = e & 7;                 // row of the chunk, channel group
            asm volatile("" ::: "memory");      // the stem weights are re-read per item: 32 fewer registers held beside the matrix weights
            f32x4 acc = *reinterpret_cast<const f32x4*>(smem + EF_B0 + 4 * g);
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const float xv = sg[r + 2 + j];              // sample t0 + r - 6 + j  (window index 0 <-> t0 - 8)
                const f32x4 w = *reinterpret_cast<const f32x4*>(smem + EF_W0 + j * 32 + 4 * g);
                acc.x = fma_pinned(w.x, xv, acc.x); acc.y = fma_pinned(w.y, xv, acc.y);      // (pinned: never v_pk_fma_f32, split16.h)
                acc.z = fma_pinned(w.z, xv, acc.z); acc.w = fma_pinned(w.w, xv, acc.w);
            }
            split16_store4s(elu4(acc), sx, R1, EF_R1_PLANE, (4 + r) * EF_XP + 4 * g);
            split16_store4s(acc, sb, Xr, EF_Q_PLANE, r * EF_QP + 4 * g);
            if (p.dbg_x0 && emit && t0 + r < p.T) *reinterpret_cast<f32x4*>(p.dbg_x0 + ((long long)b * p.T + t0 + r) * 32 + 4 * g) = acc;
        }
        // reflect padding of the k3 conv at the clip start ([HF]:157-176): x0e[-1] = x0e[1], x0e[-2] = x0e[2]  (rows 3, 2 <- 5, 6)
        if (t0 == 0 && lane < 32) copy_row(R1, EF_R1_PLANE, EF_XP, 2 + (lane >> 4), 6 - (lane >> 4), lane & 15);

        // ---- stage A: hidden = ELU(W3 * Xe + b3), 32 rows x 16 channels
        {
            f32x4 aH[2] = {zero4, zero4}, aL[2] = {zero4, zero4};      // hi*hi and the two cross terms on separate chains
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) {
                f16x8 xh[2], xl[2];
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int o = (2 + a * 16 + li + ks) * EF_XP + 8 * kq;
                    xh[a] = *reinterpret_cast<const f16x8*>(R1 + o);
                    xl[a] = *reinterpret_cast<const f16x8*>(R1 + EF_R1_PLANE + o);
                }
                const f16x8 w3h = *reinterpret_cast<const f16x8*>(W3s + ((ks * 2 + 0) * 64 + lane) * 8);
                const f16x8 w3l = *reinterpret_cast<const f16x8*>(W3s + ((ks * 2 + 1) * 64 + lane) * 8);
#pragma unroll
                for (int a = 0; a < 2; ++a) aL[a] = ef_mfma(w3l, xh[a], aL[a]);
#pragma unroll
                for (int a = 0; a < 2; ++a) aH[a] = ef_mfma(w3h, xh[a], aH[a]);
#pragma unroll
                for (int a = 0; a < 2; ++a) aL[a] = ef_mfma(w3h, xl[a], aL[a]);
            }
            const f32x4 b3v = *reinterpret_cast<const f32x4*>(smem + EF_B3 + 4 * kq);
            const f32x4 i3v = *reinterpret_cast<const f32x4*>(smem + EF_I3 + 4 * kq) * ix;
#pragma unroll
            for (int a = 0; a < 2; ++a)
                split16_store4s(elu4(ef_fma4(aH[a] + aL[a], i3v, b3v)), sb, Hs, EF_Q_PLANE, (a * 16 + li) * EF_QP + 4 * kq);
        }

        // ---- stage B: y1 = [W1 | Ws] * [hidden | x0] + bf; ELU(y1) -> Y1e rows 2..33 (over Xe: stage A is done with it)
        {
            f32x4 acc[2][2] = {{zero4, zero4}, {zero4, zero4}};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const _Float16* src = ks ? Xr : Hs;
                f16x8 xh[2], xl[2];
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int o = (a * 16 + li) * EF_QP + 8 * kq;
                    xh[a] = *reinterpret_cast<const f16x8*>(src + o);
                    xl[a] = *reinterpret_cast<const f16x8*>(src + EF_Q_PLANE + o);
                }
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int c = 0; c < 2; ++c) acc[a][c] = ef_mfma(wf[ks][c][1], xh[a], acc[a][c]);
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int c = 0; c < 2; ++c) acc[a][c] = ef_mfma(wf[ks][c][0], xl[a], acc[a][c]);
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int c = 0; c < 2; ++c) acc[a][c] = ef_mfma(wf[ks][c][0], xh[a], acc[a][c]);
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const f32x4 bfv = *reinterpret_cast<const f32x4*>(smem + EF_BF + 16 * c + 4 * kq);
                const f32x4 ifv = *reinterpret_cast<const f32x4*>(smem + EF_IF + 16 * c + 4 * kq) * ib;
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const f32x4 v = ef_fma4(acc[a][c], ifv, bfv);
                    const int t = t0 + a * 16 + li;
                    if (p.dbg_y1 && emit && t < p.T) *reinterpret_cast<f32x4*>(p.dbg_y1 + ((long long)b * p.T + t) * 32 + 16 * c + 4 * kq) = v;
                    split16_store4s(elu4(v), sy, R1, EF_R1_PLANE, (2 + a * 16 + li) * EF_XP + 16 * c + 4 * kq);
                }
            }
        }
        // clip edges of the strided conv's input ([HF]:157-176): y1e[-1] = y1e[1], y1e[-2] = y1e[2]; T odd: y1e[T] = y1e[T-2]
        if (t0 == 0 && lane < 32) copy_row(R1, EF_R1_PLANE, EF_XP, lane >> 4, 4 - (lane >> 4), lane & 15);
        if ((p.T & 1) && p.T < t0 + EF_ROWS && lane < 16) copy_row(R1, EF_R1_PLANE, EF_XP, p.T - t0 + 2, p.T - t0, lane);

        // ---- stage C: y[m] = Wd * [y1e[2m-2] | y1e[2m-1] | y1e[2m] | y1e[2m+1]] + bd, 16 rows x 64 channels
        {
            f32x4 acc[4] = {zero4, zero4, zero4, zero4};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int o = (2 * li + ks) * EF_XP + 8 * kq;
                const f16x8 xh = *reinterpret_cast<const f16x8*>(R1 + o);
                const f16x8 xl = *reinterpret_cast<const f16x8*>(R1 + EF_R1_PLANE + o);
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = ef_mfma(wd[ks][c][1], xh, acc[c]);
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = ef_mfma(wd[ks][c][0], xl, acc[c]);
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = ef_mfma(wd[ks][c][0], xh, acc[c]);
            }
            // Every accumulator is read by the VALU on every path (also in a warm-up chunk, whose rows are not stored).  This was the first
            // suspect for the two wrong stem rows in lanes 48..63 (a late matrix write landing on reused registers) and is NOT the
            // cause: with the accumulators always consumed the wrong rows stayed (profiles/r3_pk_fma_hazard.md, fourth table row);
            // they went away with the packed FMAs of the stem (fma_pinned above, -fno-slp-vectorize).  Kept because it is harmless.
            const int m = (t0 >> 1) + li;
            const bool st = emit && m < p.M;
            float* yr = p.y + yb_off + (long long)m * 64 + 4 * kq;
            unsigned cmax = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f32x4 bdv = *reinterpret_cast<const f32x4*>(smem + EF_BD + 16 * c + 4 * kq);
                const f32x4 idv = *reinterpret_cast<const f32x4*>(smem + EF_ID + 16 * c + 4 * kq) * iy;
                const f32x4 v = ef_fma4(acc[c], idv, bdv);
                amax_acc4(cmax, v);
                if (st) *reinterpret_cast<f32x4*>(yr + 16 * c) = v;
            }
            omax = st && cmax > omax ? cmax : omax;
        }
    }
    if (p.amax_out) amax_flush(omax, amax_at(p.amax_out, b));
}

}  // namespace ac
