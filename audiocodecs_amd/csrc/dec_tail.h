// dec_tail: the thin-channel tail of the EnCodec decoder as ONE kernel (the mirror image of enc_front.h)
//     xe [B][L][64] (ELU'd)  ->  ConvTranspose1d(64, 32, k4, s2)  ->  ResnetBlock(32)  ->  ELU  ->  Conv1d(32, 1, k7)  ->  sig [B][2L]
// ([HF] modeling_encodec.py:330-341 EncodecDecoder.layers[-5..-1], called from audiocodecs/encodec.py:139).  As separate
// kernels (thin_conv6_kernel, rb_fused6_kernel<32>, head_kernel) the two 32-channel tensors make 3.9 GB round trips each.
//
// One WAVE = one stream over a segment of one clip, 16 input rows (32 output samples) per chunk, every intermediate in
// the wave's own LDS slab, no workgroup barrier in the loop:
//   stage U   u[2m + ph] = Wp[ph] * [xe[m-1] | xe[m]] + b          4 k-steps (thin_conv6.h mapping; xe[-1] = 0, right trim = rows
//             never computed); ELU(u) and u split (split16.h) into the slabs Ue (2 rows of halo carried over) and Ur
//   stage A   h = ELU(W3 * Ue[t-2 .. t] + b3)                     3 k-steps
//   stage B   v = [W1 | Ws] * [h | u] + (b1 + bs);  ELU(v) kept in fp32 in the slab Ve (6 rows of halo carried over)
//   head      sig[t] = bh + sum_j sum_c wh[j][c] ELU(v)[t-6+j][c]   fp32 FMAs, lane = (sample, channel half), halves added
// Clip start: reflect padding of the k3 conv (ue[-i] = ue[i]) and of the head (ve[-i] = ve[i]) by row copies; a segment
// inside a clip runs one warm-up chunk without output.  Stage U's weights in registers (128 VGPRs); stage A's and B's fragment
// images and the head's weights in LDS shared by the workgroup's 8 waves (no spill: a kernel with scratch cannot be replayed
// safely from a hipGraph once the runtime has resized its scratch buffer).
// split16 scales from BOUNDS (enc_front.h): U = ub0 + ub1 amax(xe) for ELU(u);  min(U, hb0 + hb1 U) for [h | u].
#pragma once
#include <hip/hip_runtime.h>
#include "enc_front.h"

namespace ac {

struct DecTailParams {
    const float* xe;         // [B][L][64] ELU'd output of the 64-channel block
    const __bf16* wuf;       // thin_conv6.h image of the transposed conv: [4 n-tiles][4 k-steps][2 planes][64][8]
    const __bf16* w3f;       // rb_fused6.h images of the 32-channel block
    const __bf16* wff;
    const float* bu;         // [64] (bias repeated per phase)
    const float* winvu;      // [64]
    const float* b3;         // [16]
    const float* winv3;
    const float* bf;         // [32]
    const float* winvf;
    const float* wh;         // head weights [7][32] (ac_api's packing: [tap][channel])
    const float* bh;         // [1]
    float* sig;              // [B][2L]
    float* dbg_u;            // test hook: optional raw transposed-conv output [B][2L][32]
    float* dbg_v;            // test hook: optional raw block output [B][2L][32]
    int B, L;
    int seg_chunks, segs_per_clip;
    const unsigned* amax_x;  // split16.h slot [B] of xe
    float ub0, ub1;          // |u| <= ub0 + ub1 amax(xe)
    float hb0, hb1;          // |h| <= hb0 + hb1 bound(u)
};

constexpr int DT_ROWS = 16;                                                     // input rows per chunk (32 output samples)
constexpr int DT_XSP = 80, DT_XS_PLANE = 17 * DT_XSP;                            // Xs: 17 rows of 64 channels, 160-byte rows
constexpr int DT_VP = 36;                                                       // Ve: fp32 rows of 32 channels, 144-byte rows
constexpr int DT_VE_HALO_BYTES = 6 * DT_VP * 4;                                  // 864
constexpr int DT_X_BYTES = 6144;                                                // Xs (5440) / Hs (6144) / Ve rows 6..37 (4608) share this region
constexpr int DT_UEP = 40, DT_UE_PLANE = 34 * DT_UEP;                            // Ue: rows 0,1 halo, 2..33 this chunk
constexpr int DT_QP = 48, DT_Q_PLANE = 32 * DT_QP;                               // Hs: 96-byte rows
constexpr int DT_URP = 40, DT_UR_PLANE = 32 * DT_URP;                            // Ur: 80-byte rows (the LDS budget pays for stage B's weights)
constexpr int DT_XPREV_BYTES = 2 * DT_XSP * 2;                                   // the chunk's last input row, both planes: the next chunk's xe[m0-1]
constexpr int DT_WAVE_BYTES = DT_VE_HALO_BYTES + DT_X_BYTES + 2 * DT_UE_PLANE * 2 + 2 * DT_UR_PLANE * 2 + DT_XPREV_BYTES;
constexpr int DT_CONST_FLOATS = 512;
constexpr int DT_WF_HALFS = 2 * 2 * 2 * 512;                                     // stage B's fragment image [2 n-tiles][2 k-steps][2 planes][64][8]
constexpr int DT_SHARED_BYTES = DT_CONST_FLOATS * 4 + (EF_W3_HALFS + DT_WF_HALFS) * 2;
constexpr size_t DT_LDS = (size_t)DT_SHARED_BYTES + 8 * (size_t)DT_WAVE_BYTES;
static_assert(DT_WAVE_BYTES % 16 == 0 && DT_VE_HALO_BYTES % 16 == 0 && 2 * DT_XS_PLANE * 2 <= DT_X_BYTES && 2 * DT_Q_PLANE * 2 <= DT_X_BYTES &&
              32 * DT_VP * 4 <= DT_X_BYTES, "slab layout");
static_assert(DT_LDS <= 160 * 1024, "one 8-wave workgroup per CU");
// offsets (floats) inside the constant block
#define DT_FENCE() asm volatile("" ::: "memory")
constexpr int DT_WH = 0, DT_BH = 224, DT_B3 = 256, DT_I3 = 272, DT_BF = 288, DT_IF = 320, DT_BU = 352, DT_IU = 416;

__global__ __launch_bounds__(512, 2) void dec_tail_kernel(const DecTailParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;

    // ---- shared constants and the k3 conv's fragment image -> LDS (the only workgroup-wide step)
    for (int e = tid; e < 480; e += 512) {
        float v;
        if (e < 224) v = p.wh[e];
        else if (e < 256) v = e == 224 ? p.bh[0] : 0.f;
        else if (e < 272) v = p.b3[e - 256];
        else if (e < 288) v = p.winv3[e - 272];
        else if (e < 320) v = p.bf[e - 288];
        else if (e < 352) v = p.winvf[e - 320];
        else if (e < 416) v = p.bu[e - 352];
        else v = p.winvu[e - 416];
        smem[e] = v;
    }
    const _Float16* W3s = reinterpret_cast<const _Float16*>(smem + DT_CONST_FLOATS);
    const _Float16* WFs = W3s + EF_W3_HALFS;
    for (int e = tid; e < EF_W3_HALFS / 8; e += 512)
        *reinterpret_cast<f16x8*>(reinterpret_cast<_Float16*>(smem + DT_CONST_FLOATS) + e * 8) = *reinterpret_cast<const f16x8*>(p.w3f + (long long)e * 8);
    for (int e = tid; e < DT_WF_HALFS / 8; e += 512)
        *reinterpret_cast<f16x8*>(reinterpret_cast<_Float16*>(smem + DT_CONST_FLOATS) + EF_W3_HALFS + e * 8) = *reinterpret_cast<const f16x8*>(p.wff + (long long)e * 8);
    __syncthreads();

    char* wbase = reinterpret_cast<char*>(smem) + DT_SHARED_BYTES + wave * DT_WAVE_BYTES;
    float* Ve = reinterpret_cast<float*>(wbase);                                        // rows 0..5 halo, 6..37 this chunk
    _Float16* Xs = reinterpret_cast<_Float16*>(wbase + DT_VE_HALO_BYTES);              // [2][17][80]
    _Float16* Hs = Xs;                                                                  // [2][32][48]  (Xs is dead by then)
    _Float16* Ue = reinterpret_cast<_Float16*>(wbase + DT_VE_HALO_BYTES + DT_X_BYTES);  // [2][34][40]
    _Float16* Ur = Ue + 2 * DT_UE_PLANE;                                                // [2][32][40]
    _Float16* Xprev = Ur + 2 * DT_UR_PLANE;                                              // [2][80]

    const int sid = blockIdx.x * 8 + wave;
    if (sid >= p.B * p.segs_per_clip) return;
    const int b = sid / p.segs_per_clip, seg = sid - b * p.segs_per_clip;
    const int nchunks = (p.L + DT_ROWS - 1) / DT_ROWS;
    const int c_first = seg * p.seg_chunks;
    const int c_last = c_first + p.seg_chunks < nchunks ? c_first + p.seg_chunks : nchunks;
    if (c_first >= c_last) return;
    const int T = 2 * p.L;

    f16x8 wu[4][4][2];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) wu[ks][c][pl] = *reinterpret_cast<const f16x8*>(p.wuf + ((long long)((c * 4 + ks) * 2 + pl) * 64 + lane) * 8);

    // ---- split16 scales of this clip
    const unsigned am = *amax_at(p.amax_x, b);
    const int exs = s16_exponent(am);
    const float Ub = __fmaf_rn(p.ub1, __uint_as_float(am), p.ub0) * 1.0000005f;
    const float Hb = __fmaf_rn(p.hb1, Ub, p.hb0) * 1.0000005f;
    const int eu = ef_exp(Ub), eh = ef_exp(Hb), eb = eh < eu ? eh : eu;
    const float sxs = s16_pow2(exs), ixs = s16_pow2(-exs);   // xe in the transposed conv (its amax is exact)
    const float su = s16_pow2(eu), iu = s16_pow2(-eu);       // ELU(u) in the k3 conv
    const float sb = s16_pow2(eb), ib = s16_pow2(-eb);       // hidden and raw u share stage B's accumulator

    const float* xb = p.xe + (long long)b * p.L * 64;
    const int clip_bytes = p.L * 256;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, clip_bytes, 0x00020000);
    f32x4 rx[4];
    auto load_x = [&](int m0) {                              // 16 input rows of 64 channels; rows past the clip read 0
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = lane + 64 * i;
            rx[i] = bufload16(rs, (m0 + (e >> 4)) * 256 + (e & 15) * 16, 0);
        }
    };
    auto copy_row = [&](_Float16* base, int plane, int pitch, int dst, int src, int l16) {
        const int pl = l16 >> 3, c4 = (l16 & 7) * 4;
        const f16x4_t v = *reinterpret_cast<const f16x4_t*>(base + pl * plane + src * pitch + c4);
        *reinterpret_cast<f16x4_t*>(base + pl * plane + dst * pitch + c4) = v;
    };

    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const f16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
    const long long ob = (long long)b * T;
    int ch = c_first > 0 ? c_first - 1 : 0;                  // a segment inside a clip warms its halos up on the chunk before
    load_x(ch * DT_ROWS);
    if (lane < 32) *reinterpret_cast<f16x4_t*>(Xprev + (lane >> 4) * DT_XSP + (lane & 15) * 4) = f16x4_t{0, 0, 0, 0};
    for (; ch < c_last; ++ch) {
        const int m0 = ch * DT_ROWS, t0 = 2 * m0;
        const bool emit = ch >= c_first;
        // (Xs / Hs / Ve rows 6..37 are views of ONE region with different element types: compiler fences at the hand-overs keep
        //  type-based alias analysis from moving an access of one view across an access of another)
        DT_FENCE();
        // ---- stage the input rows (split once, the clip's scale); Ue halo = the previous chunk's last two rows
        // slab row 0 = xe[m0-1]: the previous chunk's last row (zeros at a clip start -- the transposed conv's x[-1] -- and at the
        // start of a warm-up chunk, whose first output rows nothing reads)
        if (lane < 32) *reinterpret_cast<f16x4_t*>(Xs + (lane >> 4) * DT_XS_PLANE + (lane & 15) * 4) = *reinterpret_cast<const f16x4_t*>(Xprev + (lane >> 4) * DT_XSP + (lane & 15) * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = lane + 64 * i;
            split16_store4s(rx[i], sxs, Xs, DT_XS_PLANE, (1 + (e >> 4)) * DT_XSP + 4 * (e & 15));
        }
        if (lane < 32) copy_row(Ue, DT_UE_PLANE, DT_UEP, lane >> 4, 32 + (lane >> 4), lane & 15);

        // ---- stage U: 16 input rows -> 32 rows of u (phase ph = tile >> 1), 32 channels
        {
            f32x4 acc[4] = {zero4, zero4, zero4, zero4};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int o = (li + (ks >> 1)) * DT_XSP + (ks & 1) * 32 + 8 * kq;
                const f16x8 xh = *reinterpret_cast<const f16x8*>(Xs + o);
                const f16x8 xl = *reinterpret_cast<const f16x8*>(Xs + DT_XS_PLANE + o);
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = ef_mfma(wu[ks][c][1], xh, acc[c]);
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = ef_mfma(wu[ks][c][0], xl, acc[c]);
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = ef_mfma(wu[ks][c][0], xh, acc[c]);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f32x4 buv = *reinterpret_cast<const f32x4*>(smem + DT_BU + 16 * c + 4 * kq);
                const f32x4 iuv = *reinterpret_cast<const f32x4*>(smem + DT_IU + 16 * c + 4 * kq) * ixs;
                const f32x4 v = ef_fma4(acc[c], iuv, buv);
                const int rho = 2 * li + (c >> 1), co = 16 * (c & 1) + 4 * kq;      // output row of the chunk, first channel
                if (p.dbg_u && emit && t0 + rho < T) *reinterpret_cast<f32x4*>(p.dbg_u + (ob + t0 + rho) * 32 + co) = v;
                split16_store4s(elu4(v), su, Ue, DT_UE_PLANE, (2 + rho) * DT_UEP + co);
                split16_store4s(v, sb, Ur, DT_UR_PLANE, rho * DT_URP + co);
            }
        }
        if (lane < 32) *reinterpret_cast<f16x4_t*>(Xprev + (lane >> 4) * DT_XSP + (lane & 15) * 4) = *reinterpret_cast<const f16x4_t*>(Xs + (lane >> 4) * DT_XS_PLANE + 16 * DT_XSP + (lane & 15) * 4);
        // reflect padding of the k3 conv at the clip start: ue[-1] = ue[1], ue[-2] = ue[2]
        if (t0 == 0 && lane < 32) copy_row(Ue, DT_UE_PLANE, DT_UEP, lane >> 4, 4 - (lane >> 4), lane & 15);

        // ---- stage A: hidden = ELU(W3 * Ue + b3) -> Hs (over Xs)
        DT_FENCE();
        {
            f32x4 aH[2] = {zero4, zero4}, aL[2] = {zero4, zero4};
#pragma unroll
            for (int ks = 0; ks < 3; ++ks) {
                f16x8 xh[2], xl[2];
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int o = (a * 16 + li + ks) * DT_UEP + 8 * kq;
                    xh[a] = *reinterpret_cast<const f16x8*>(Ue + o);
                    xl[a] = *reinterpret_cast<const f16x8*>(Ue + DT_UE_PLANE + o);
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
            const f32x4 b3v = *reinterpret_cast<const f32x4*>(smem + DT_B3 + 4 * kq);
            const f32x4 i3v = *reinterpret_cast<const f32x4*>(smem + DT_I3 + 4 * kq) * iu;
#pragma unroll
            for (int a = 0; a < 2; ++a)
                split16_store4s(elu4(ef_fma4(aH[a] + aL[a], i3v, b3v)), sb, Hs, DT_Q_PLANE, (a * 16 + li) * DT_QP + 4 * kq);
        }
        // ---- stage B: v = [W1 | Ws] * [hidden | u] + bf; ELU(v) -> Ve rows 6..37 in fp32 (over Hs: all its reads are issued first)
        {
            f32x4 acc[2][2] = {{zero4, zero4}, {zero4, zero4}};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                f16x8 xh[2], xl[2];
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    if (ks == 0) {            // k = 0..15 hidden, 16..31 K padding (zero weight columns): lanes kq >= 2 supply zeros
                        const int o = (a * 16 + li) * DT_QP + 8 * kq;
                        xh[a] = kq < 2 ? *reinterpret_cast<const f16x8*>(Hs + o) : z8;
                        xl[a] = kq < 2 ? *reinterpret_cast<const f16x8*>(Hs + DT_Q_PLANE + o) : z8;
                    } else {
                        const int o = (a * 16 + li) * DT_URP + 8 * kq;
                        xh[a] = *reinterpret_cast<const f16x8*>(Ur + o);
                        xl[a] = *reinterpret_cast<const f16x8*>(Ur + DT_UR_PLANE + o);
                    }
                }
                f16x8 wfh[2], wfl[2];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    wfh[c] = *reinterpret_cast<const f16x8*>(WFs + (((c * 2 + ks) * 2 + 0) * 64 + lane) * 8);
                    wfl[c] = *reinterpret_cast<const f16x8*>(WFs + (((c * 2 + ks) * 2 + 1) * 64 + lane) * 8);
                }
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int c = 0; c < 2; ++c) acc[a][c] = ef_mfma(wfl[c], xh[a], acc[a][c]);
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int c = 0; c < 2; ++c) acc[a][c] = ef_mfma(wfh[c], xl[a], acc[a][c]);
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int c = 0; c < 2; ++c) acc[a][c] = ef_mfma(wfh[c], xh[a], acc[a][c]);
            }
            DT_FENCE();
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const f32x4 bfv = *reinterpret_cast<const f32x4*>(smem + DT_BF + 16 * c + 4 * kq);
                const f32x4 ifv = *reinterpret_cast<const f32x4*>(smem + DT_IF + 16 * c + 4 * kq) * ib;
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const f32x4 v = ef_fma4(acc[a][c], ifv, bfv);
                    const int t = t0 + a * 16 + li;
                    if (p.dbg_v && emit && t < T) *reinterpret_cast<f32x4*>(p.dbg_v + (ob + t) * 32 + 16 * c + 4 * kq) = v;
                    *reinterpret_cast<f32x4*>(Ve + (6 + a * 16 + li) * DT_VP + 16 * c + 4 * kq) = elu4(v);
                }
            }
        }
        // reflect padding of the head at the clip start: ve[-i] = ve[i], i = 1..6  (rows 6 - i <- 6 + i)
        if (t0 == 0 && lane < 48) {
            const int i = 1 + (lane >> 3), c4 = (lane & 7) * 4;
            *reinterpret_cast<f32x4*>(Ve + (6 - i) * DT_VP + c4) = *reinterpret_cast<const f32x4*>(Ve + (6 + i) * DT_VP + c4);
        }

        if (ch + 1 < c_last) load_x(m0 + DT_ROWS);            // the next chunk's rows: in flight during the head (the matrix stages have no registers to spare)
        // ---- head: lane = (sample tl, channel half); taps ascending, channels ascending inside a half, halves added last
        {
            const int tl = lane & 31, hf = lane >> 5;
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                asm volatile("" ::: "memory");               // weights re-read per tap (not hoisted into 112 registers)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 xv = *reinterpret_cast<const f32x4*>(Ve + (tl + j) * DT_VP + 16 * hf + 4 * q);
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(smem + DT_WH + j * 32 + 16 * hf + 4 * q);
                    acc = fmaf(wv.x, xv.x, acc); acc = fmaf(wv.y, xv.y, acc);
                    acc = fmaf(wv.z, xv.z, acc); acc = fmaf(wv.w, xv.w, acc);
                }
            }
            const float other = __shfl_xor(acc, 32);
            if (emit && hf == 0 && t0 + tl < T) p.sig[ob + t0 + tl] = smem[DT_BH] + (acc + other);
        }
        DT_FENCE();
        // halo of the next chunk's head: the last six rows of Ve
        if (lane < 48) {
            const int r = lane >> 3, c4 = (lane & 7) * 4;
            *reinterpret_cast<f32x4*>(Ve + r * DT_VP + c4) = *reinterpret_cast<const f32x4*>(Ve + (32 + r) * DT_VP + c4);
        }
    }
}

}  // namespace ac
