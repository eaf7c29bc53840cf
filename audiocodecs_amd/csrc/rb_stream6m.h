// rb_stream6m: Mimi's 64-channel identity-shortcut residual block ([HF] models/mimi/modeling_mimi.py MimiResnetBlock; zero padding, causal)
// as rb_stream6.h builds it (sixteen waves per CU, weights in LDS, one wave = one stream of 16-row tiles, accumulators are operands), with the
// layer on either side of it folded in -- the two widest tensors of Mimi's path (64 channels at 24 kHz: 7.9 GB at 128 clips x 10 s) then
// cross HBM once instead of three times:
//   STEM  (encoder's first block): the block's input is Conv1d(1, 64, k7) of the SIGNAL, evaluated on the matrix pipe per tile exactly as
//         enc_stream.h does (W0 [64 x (7 taps -> K 32)] times a per-lane sample window from the wave's window in LDS).  An accumulator tile
//         of the stem is a lane's loaded float4 of rb_stream6.h (tile c = channels 16 c + 4 kq ..), so nothing else changes: the stem kernel
//         (1.56 ms per step) and the block's 7.9 GB read go away.
//   HEAD  (decoder's last block): the block's ELU'd output never leaves the CU; the final Conv1d(64, 1, k <= 8) runs as
//         G[row][tap] = ELU(y)[row][:] . wh[tap][:]  (two k-steps on the matrix pipe, the operand is the output accumulator split in
//         registers: dec_stream.h) and the diagonal sum  sig[t] = bh + sum_j G[t - (k-1) + j][j]  over a (k - 1 + 16)-row window in LDS.
//         Replaces rb_fused6_head_kernel (rb_fused6.h HEAD: 64-row tiles at two waves per SIMD, the conv as scalar FMAs over an fp32 tile in LDS).
// Scales: ELU(x) and [hidden | -] as rb_stream6.h from the block input's amax -- exact (HEAD, plain) or, for STEM, the bound
// sb0 + sb1 amax(sig) of the stem's output; ELU(y) for the head from the bound |y| <= |x| + fb0 + fb1h bound(hidden).
#pragma once
#include "enc_stream.h"

namespace ac {

struct RbStreamMParams {
    const float* xr;         // [B][L][64] raw input (not STEM)
    const float* sig;        // STEM: [B][L] samples
    const __bf16* w0f;       // STEM: stem image [4 n-tiles][1 k-step][2 planes][64][8] (tap j in column j)
    const __bf16* w3f;       // k3 conv   [2][6][2][64][8]   (permuted columns: core.h perm32)
    const __bf16* wff;       // 1x1 conv  [4][1][2][64][8]
    const __bf16* whf;       // HEAD: head image [1][2][2][64][8] (row = tap, rows k .. 15 zero)
    const float *b0, *winv0; // STEM [64]
    const float *b3, *winv3; // [32]
    const float *bf, *winvf; // [64]
    const float *bh, *winvh; // HEAD [1], [16]
    float* y;                // optional raw output [B][L][64]
    float* y_elu;            // optional ELU'd output
    float* head_y;           // HEAD: [B][L]
    int head_k;
    int B, L;
    int nseg, seg_rows;      // segments per clip, rows per segment (a multiple of 16)
    const unsigned* amax_in; // slot [B]: the block input's amax, or (STEM) the samples'
    unsigned* amax_out;
    float sb0, sb1;          // STEM: |x0| <= sb0 + sb1 amax(sig)
    float hb0, hb1;          // |hidden| <= hb0 + hb1 amax(x)
    float fb0, fb1h;         // HEAD: |y - x| <= fb0 + fb1h bound(hidden)
};

constexpr int RM_WA = 0, RM_WB = 24576, RM_W0 = RM_WB + 8192, RM_WH = RM_W0 + 8192, RM_CONST = RM_WH + 4096;     // byte offsets
constexpr int RM_B3 = 0, RM_I3 = 32, RM_BF = 64, RM_IF = 128, RM_B0 = 192, RM_I0 = 256, RM_IH = 320, RM_BH = 336, RM_CONST_FLOATS = 352;
constexpr int RM_SLAB = 4 * 80 * 16, RM_SG_FLOATS = 48, RM_GP = 9, RM_G_FLOATS = 208;     // sample window 24 + zero area at 40; G window 23 rows x 9
constexpr int RM_WAVE_BYTES = RM_SLAB + RM_SG_FLOATS * 4 + RM_G_FLOATS * 4;
constexpr int RM_SHARED_BYTES = RM_CONST + RM_CONST_FLOATS * 4;
constexpr size_t RM_LDS = (size_t)RM_SHARED_BYTES + 16 * (size_t)RM_WAVE_BYTES;
static_assert(RM_LDS <= 160 * 1024 && RM_SHARED_BYTES % 16 == 0 && RM_WAVE_BYTES % 16 == 0, "one 16-wave workgroup per CU");

template <bool STEM, bool HEAD, bool YR, bool YE>
__global__ __launch_bounds__(1024) void rb_stream6m_kernel(const RbStreamMParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rm_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;

    // ---- weights and constants -> LDS (once per workgroup)
    {
        u32x4_t* d = reinterpret_cast<u32x4_t*>(rm_smem);
        for (int i = tid; i < 24576 / 16; i += 1024) d[RM_WA / 16 + i] = reinterpret_cast<const u32x4_t*>(p.w3f)[i];
        for (int i = tid; i < 8192 / 16; i += 1024) d[RM_WB / 16 + i] = reinterpret_cast<const u32x4_t*>(p.wff)[i];
        if (STEM) for (int i = tid; i < 8192 / 16; i += 1024) d[RM_W0 / 16 + i] = reinterpret_cast<const u32x4_t*>(p.w0f)[i];
        if (HEAD) for (int i = tid; i < 4096 / 16; i += 1024) d[RM_WH / 16 + i] = reinterpret_cast<const u32x4_t*>(p.whf)[i];
        float* cs = reinterpret_cast<float*>(rm_smem + RM_CONST);
        for (int e = tid; e < RM_CONST_FLOATS; e += 1024) {
            float v = 0.f;
            if (e < RM_I3) v = p.b3[e - RM_B3];
            else if (e < RM_BF) v = p.winv3[e - RM_I3];
            else if (e < RM_IF) v = p.bf[e - RM_BF];
            else if (e < RM_B0) v = p.winvf[e - RM_IF];
            else if (e < RM_I0) v = STEM ? p.b0[e - RM_B0] : 0.f;
            else if (e < RM_IH) v = STEM ? p.winv0[e - RM_I0] : 0.f;
            else if (e < RM_BH) v = HEAD ? p.winvh[e - RM_IH] : 0.f;
            else if (e == RM_BH) v = HEAD ? p.bh[0] : 0.f;
            cs[e] = v;
        }
    }
    __syncthreads();

    const unsigned char* wa_l = rm_smem + RM_WA + lane * 16;
    const unsigned char* wb_l = rm_smem + RM_WB + lane * 16;
    const unsigned char* w0_l = rm_smem + RM_W0 + lane * 16;
    const unsigned char* wh_l = rm_smem + RM_WH + lane * 16;
    const float* cs = reinterpret_cast<const float*>(rm_smem + RM_CONST);
    const float* c_l = cs + 4 * kq;
    unsigned char* slab = rm_smem + RM_SHARED_BYTES + wave * RM_WAVE_BYTES;
    float* sg = reinterpret_cast<float*>(slab + RM_SLAB);          // [0..23] samples t - 8 .. t + 15, [40..47] zeros
    float* G = sg + RM_SG_FLOATS;
    unsigned char* sl = slab + (kq * 80 + li) * 16;                 // rb_stream6.h slab: unit = plane 36 + kc 18 + row
    auto unit = [](int pl, int kc, int row) { return (pl * 36 + kc * 18 + row) * 16; };
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const int clip_bytes = p.L * 256;
    const int total = p.B * p.nseg;
    const int hk1 = HEAD ? p.head_k - 1 : 0;
    if (STEM && lane < 8) sg[40 + lane] = 0.f;

    for (int sgi = blockIdx.x * 16 + wave; sgi < total; sgi += gridDim.x * 16) {
        const int b = sgi / p.nseg;
        const int t_own = (sgi - b * p.nseg) * p.seg_rows;          // first row this segment emits
        const int t_beg = HEAD && t_own > 0 ? t_own - 16 : t_own;   // HEAD: one warm-up tile fills the head's window
        const int t_end = t_own + p.seg_rows < p.L ? t_own + p.seg_rows : p.L;
        const long long ob = (long long)b * p.L * 64;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(STEM ? nullptr : p.xr + ob), 0, STEM ? 0 : clip_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)(YR ? p.y + ob : nullptr), 0, YR ? clip_bytes : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t re = __builtin_amdgcn_make_buffer_rsrc((void*)(YE ? p.y_elu + ob : nullptr), 0, YE ? clip_bytes : 0, 0x00020000);
        const unsigned am_in = *amax_at(p.amax_in, b);
        // the block input's magnitude: exact, or the stem's bound
        const float xb = STEM ? __fmaf_rn(p.sb1, __uint_as_float(am_in), p.sb0) * 1.0000005f : __uint_as_float(am_in);
        const Rb16Scale cs16 = rb16_scale(__float_as_uint(xb) & 0x7fffffffu, p.hb0, p.hb1);
        const int esg = s16_exponent(am_in);
        const float ssg = s16_pow2(esg), isg = s16_pow2(-esg);      // STEM: the samples
        float sv = 1.f, iv_ = 1.f;                                  // HEAD: ELU(y) as the head's operand
        if (HEAD) {
            const float Hb = __fmaf_rn(p.hb1, xb, p.hb0) * 1.0000005f;
            const float Yb = (xb + __fmaf_rn(p.fb1h, Hb, p.fb0)) * 1.000001f;
            const int ey = es_exp(Yb);
            sv = s16_pow2(ey); iv_ = s16_pow2(-ey);
        }
        unsigned omax = 0;
        const float* sigb = STEM ? p.sig + (long long)b * p.L : nullptr;

        // the tile's rows in operand shape: r[kc][h] = channels 32 kc + 16 h + 4 kq + {0..3} of row t + li -- loaded, or (STEM) computed:
        // the sample window of the tile is fetched a tile ahead into one register per lane (lane i: sample t - 8 + i, zeros left of the clip)
        auto fetch_sig = [&](int t) -> float {
            const int q = t - 8 + lane;
            return (lane < 24 && q >= 0 && q < p.L) ? sigb[q] : 0.f;
        };
        auto request = [&](int t, f32x4 (&r)[2][2]) {
            const int row = t + li;
            const int ro = row >= 0 && row < p.L ? row * 256 + kq * 16 : 0x7fff0000;
#pragma unroll
            for (int kc = 0; kc < 2; ++kc)
#pragma unroll
                for (int h = 0; h < 2; ++h) r[kc][h] = bufload16(rs, ro + kc * 128 + h * 64, 0);
        };
        auto stem = [&](float win, f32x4 (&r)[2][2]) {              // win: this lane's sample of the window (fetch_sig)
            if (lane < 24) sg[lane] = win;
            const float* wp = kq == 0 ? sg + 2 + li : sg + 40;
            const f32x4 s0 = {wp[0], wp[1], wp[2], wp[3]};
            const f32x4 s1 = {wp[4], wp[5], wp[6], 0.f};
            const Hl8 so = split16_regs8(s0, s1, ssg);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f16x8 wh = *reinterpret_cast<const f16x8*>(w0_l + (c * 2 + 0) * 1024);
                const f16x8 wl = *reinterpret_cast<const f16x8*>(w0_l + (c * 2 + 1) * 1024);
                const f32x4 acc = mma16(wh, wl, so.hi, so.lo, zero4);
                const f32x4 b0v = *reinterpret_cast<const f32x4*>(c_l + RM_B0 + 16 * c);
                const f32x4 i0v = *reinterpret_cast<const f32x4*>(c_l + RM_I0 + 16 * c) * isg;
                r[c >> 1][c & 1] = es_fma4(acc, i0v, b0v);
            }
        };
        // ELU + split of one row set -> slab rows `row0 + li` (both planes)
        auto stage_xe = [&](const f32x4 (&r)[2][2], int row0_bytes) {
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) {
                const Hl8 e = split16_regs8(elu4p(r[kc][0]), elu4p(r[kc][1]), cs16.sx);
                *reinterpret_cast<f16x8*>(sl + row0_bytes + unit(0, kc, 0)) = e.hi;
                *reinterpret_cast<f16x8*>(sl + row0_bytes + unit(1, kc, 0)) = e.lo;
            }
        };

        f32x4 rx[2][2];                     // the tile's raw rows, kept until the output (identity shortcut)
        float win_next = 0.f;
        {   // ---- first tile of the segment: its rows and the two rows in front of them (zeros left of the clip: Mimi pads with zeros)
            f32x4 rh[2][2];
            if (STEM) {
                if (t_beg > 0) {
                    stem(fetch_sig(t_beg - 16), rh);               // rows t_beg - 16 .. t_beg - 1: the last two are the halo
                    if (li >= 14) stage_xe(rh, -14 * 16);
                } else if (li < 2) {
#pragma unroll
                    for (int kc = 0; kc < 2; ++kc) {
                        *reinterpret_cast<u32x4_t*>(sl + unit(0, kc, 0)) = u32x4_t{0u, 0u, 0u, 0u};
                        *reinterpret_cast<u32x4_t*>(sl + unit(1, kc, 0)) = u32x4_t{0u, 0u, 0u, 0u};
                    }
                }
                stem(fetch_sig(t_beg), rx);
                win_next = fetch_sig(t_beg + 16);
            } else {
                request(t_beg, rx);
                const int j = t_beg - 2 + li;
                const int ho = li < 2 && j >= 0 ? j * 256 + kq * 16 : 0x7fff0000;
#pragma unroll
                for (int kc = 0; kc < 2; ++kc)
#pragma unroll
                    for (int h = 0; h < 2; ++h) rh[kc][h] = bufload16(rs, ho + kc * 128 + h * 64, 0);
                if (li < 2) stage_xe(rh, 0);
            }
            stage_xe(rx, 2 * 16);
        }
        if (HEAD && lane < 8 * 8) {                                 // the head's window starts empty: zeros left of the clip, and a warm-up tile fills it
            for (int i = lane; i < 8 * RM_GP; i += 64) G[i] = 0.f;
        }

        for (int t = t_beg; t < t_end; t += 16) {
            f32x4 rn[2][2];
            float win = win_next;
            if (STEM) win_next = fetch_sig(t + 32 < t_end + 16 ? t + 32 : 0x3fffff00);
            else request(t + 16 < t_end ? t + 16 : 0x3fffff00, rn);   // (no next tile: every row out of range, no memory access)
            __builtin_amdgcn_sched_barrier(0);                     // the requests stay HERE: a tile ahead of their use

            // ---- stage A: hidden = ELU(conv_k3(xe) + b3)
            f32x4 accA[2] = {zero4, zero4};
#pragma unroll
            for (int ks = 0; ks < 6; ++ks) {
                const int j = ks >> 1, kc = ks & 1;
                const f16x8 xh = *reinterpret_cast<const f16x8*>(sl + unit(0, kc, j));
                const f16x8 xl = *reinterpret_cast<const f16x8*>(sl + unit(1, kc, j));
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const f16x8 wh = *reinterpret_cast<const f16x8*>(wa_l + ((c * 6 + ks) * 2 + 0) * 1024);
                    const f16x8 wl = *reinterpret_cast<const f16x8*>(wa_l + ((c * 6 + ks) * 2 + 1) * 1024);
                    accA[c] = mma16(wh, wl, xh, xl, accA[c]);
                }
            }
            Hl8 hf;
            {
                f32x4 hv[2];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const f32x4 b3v = *reinterpret_cast<const f32x4*>(c_l + RM_B3 + 16 * c);
                    const f32x4 i3 = *reinterpret_cast<const f32x4*>(c_l + RM_I3 + 16 * c) * cs16.ix;
                    hv[c] = elu4p(es_fma4(accA[c], i3, b3v));
                }
                hf = split16_regs8(hv[0], hv[1], cs16.sb);
            }
            // ---- stage B: y = x + W1 hidden + bf
            f32x4 acc[4];
            const int row = t + li;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f16x8 wh = *reinterpret_cast<const f16x8*>(wb_l + (c * 2 + 0) * 1024);
                const f16x8 wl = *reinterpret_cast<const f16x8*>(wb_l + (c * 2 + 1) * 1024);
                const f32x4 a = mma16(wh, wl, hf.hi, hf.lo, zero4);
                const f32x4 bfv = *reinterpret_cast<const f32x4*>(c_l + RM_BF + 16 * c);
                const f32x4 ifv = *reinterpret_cast<const f32x4*>(c_l + RM_IF + 16 * c) * cs16.ib;
                const f32x4 v = es_fma4(a, ifv, bfv);
                const f32x4 xv = rx[c >> 1][c & 1];
                acc[c] = f32x4{__fadd_rn(xv.x, v.x), __fadd_rn(xv.y, v.y), __fadd_rn(xv.z, v.z), __fadd_rn(xv.w, v.w)};
            }
            // ---- the next tile is staged before this tile's outputs leave (rb_stream6.h): halo = this tile's last two rows
            if (lane < 32) {
                unsigned char* hp = slab + ((lane >> 3) * 80 + ((lane >> 2) & 1) * 36 + ((lane >> 1) & 1) * 18 + (lane & 1)) * 16;
                const u32x4_t hv = *reinterpret_cast<const u32x4_t*>(hp + 16 * 16);
                *reinterpret_cast<u32x4_t*>(hp) = hv;
            }
            if (STEM) stem(win, rn);
            stage_xe(rn, 2 * 16);
            // ---- outputs
            if (HEAD) {
                const bool emit = t >= t_own;
                const float ihv = cs[RM_IH + li] * iv_;
                f32x4 g = zero4;
#pragma unroll
                for (int kc = 0; kc < 2; ++kc) {
                    const f32x4 e0 = elu4p(acc[2 * kc]), e1 = elu4p(acc[2 * kc + 1]);      // (rows past the clip: finite, and never read by a stored sample)
                    const Hl8 yo = split16_regs8(e0, e1, sv);
                    const f16x8 whh = *reinterpret_cast<const f16x8*>(wh_l + (kc * 2 + 0) * 1024);
                    const f16x8 whl = *reinterpret_cast<const f16x8*>(wh_l + (kc * 2 + 1) * 1024);
                    g = es_mfma(yo.lo, whh, g);
                    g = es_mfma(yo.hi, whl, g);
                    g = es_mfma(yo.hi, whh, g);
                }
                if (li < 8) {                                      // window row = time - t + (k - 1) = 4 kq + r + k - 1
                    float* gw = G + (4 * kq + hk1) * RM_GP + li;
                    gw[0] = g.x * ihv; gw[RM_GP] = g.y * ihv; gw[2 * RM_GP] = g.z * ihv; gw[3 * RM_GP] = g.w * ihv;
                }
                float s = 0.f;
                for (int j = 0; j <= hk1; ++j) s += G[(li + j) * RM_GP + j];                        // taps ascending
                if (emit && lane < 16 && row < p.L) p.head_y[(long long)b * p.L + row] = cs[RM_BH] + s;
                if (lane < 56) {                                    // the next tile's halo: the window's last k - 1 rows
                    const int r = lane >> 3, j = lane & 7;
                    if (r < hk1) G[r * RM_GP + j] = G[(16 + r) * RM_GP + j];
                }
            } else {
                const int orow = row < p.L ? row * 256 + kq * 16 : 0x7fff0000;       // rows outside the clip: out of range, dropped
                const unsigned tm = amax16(acc);
                omax = row < p.L && tm > omax ? tm : omax;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (YR) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, acc[c]), ry, orow + c * 64, 0, 0);
                    if (YE) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, elu4p(acc[c])), re, orow + c * 64, 0, 0);
                }
            }
#pragma unroll
            for (int kc = 0; kc < 2; ++kc)
#pragma unroll
                for (int h = 0; h < 2; ++h) rx[kc][h] = rn[kc][h];
        }
        if (!HEAD && p.amax_out) amax_flush(omax, amax_at(p.amax_out, b));
    }
}

}  // namespace ac
