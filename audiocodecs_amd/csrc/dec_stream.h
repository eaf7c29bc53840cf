// dec_stream: the thin-channel tail of the EnCodec decoder as ONE kernel at sixteen waves per CU (round 6; dec_tail.h is the round-3 form)
//     xe [B][L][64] (ELU'd)  ->  ConvTranspose1d(64, 32, k4, s2)  ->  ResnetBlock(32)  ->  ELU  ->  Conv1d(32, 1, k7)  ->  sig [B][2L]
// ([HF] modeling_encodec.py:330-341 EncodecDecoder.layers[-5..-1], called from audiocodecs/encodec.py:139).
// Same construction as enc_stream.h (all weights in LDS, one wave = one stream, accumulators are operands, 16 input rows = 32 output
// samples per chunk), with two things of its own:
//   * the transposed conv's output rows come out of the matrix pipe PHASE-MAJOR: accumulator tile c holds phase c >> 1, i.e. lane li has
//     rows 2 li and 2 li + 1 of u.  Nothing is re-ordered: every later stage takes "the 16 rows of one phase" as its row tile, and a tap of
//     the k3 conv is (phase, lane li or li - 1) -- four operand fragments per plane serve the three taps of both tiles;
//   * the head (one output channel, 7 taps x 32 channels: 112 scalar FMAs + 112 LDS reads per lane and chunk in dec_tail.h) runs on the
//     matrix pipe as  G[row][tap] = ELU(v)[row][:] . wh[tap][:]  (ONE k-step: M = 16 rows, N = 7 taps -> 16, K = 32 channels; the operand
//     is stage B's accumulator, split in registers) followed by the diagonal sum  sig[t] = bh + sum_j G[t - 6 + j][j]  over a 38-row
//     window of G in LDS (9-float rows: conflict-free both ways; the 6 halo rows carried from chunk to chunk).
// Slab per wave, per kq block of 80 units (16 B): [0..3] XP = the previous chunk's last input row (x[m0 - 1] of the transposed conv; zeros at
// a clip start), [4 + 34 pl + 17 g + r] the region: first the input rows (g = 32-channel half kc, r = 0..16), then ELU(u) (g = phase,
// r = 0 the previous chunk's last row of that phase), [72..75] EH = ELU(u) rows r = 16 of the previous chunk.  Then the G window.
// Edges as in dec_tail.h: ue[-i] = ue[i] (k3 conv) and ve[-i] = ve[i] (head) at the clip start by copies, rows past the clip never stored.
#pragma once
#include "enc_stream.h"

namespace ac {

struct DecStreamParams {
    const float* xe;         // [B][L][64] ELU'd output of the 64-channel block
    const __bf16* wuf;       // transposed conv   [4 n-tiles][4 k-steps][2 planes][64][8]   (n = phase * 32 + channel; permuted columns)
    const __bf16* w3f;       // k3 conv           [1][3][2][64][8]
    const __bf16* wff;       // [1x1 | shortcut]  [2][2][2][64][8]
    const __bf16* whf;       // head              [1][1][2][64][8]                          (row = tap, rows 7..15 zero)
    const float *bu, *winvu; // [64]
    const float *b3, *winv3; // [16]
    const float *bf, *winvf; // [32]
    const float *bh, *winvh; // [1], [16]
    float* sig;              // [B][2L]
    float* dbg_u;            // test hook: optional raw transposed-conv output [B][2L][32]
    float* dbg_v;            // test hook: optional raw block output [B][2L][32]
    int B, L;
    int seg_chunks, segs_per_clip;
    const unsigned* amax_x;  // split16.h slot [B] of xe
    float ub0, ub1;          // |u| <= ub0 + ub1 amax(xe)
    float hb0, hb1;          // |h| <= hb0 + hb1 bound(u)
    float fb0, fb1h, fb1x;   // |v| <= fb0 + fb1h bound(h) + fb1x bound(u)
};

constexpr int DS_WAVES = 16;
constexpr int DS_WU = 0, DS_W3 = DS_WU + 32768, DS_WF = DS_W3 + 6144, DS_WH = DS_WF + 8192, DS_CONST = DS_WH + 2048;   // byte offsets
constexpr int DS_BU = 0, DS_IU = 64, DS_B3 = 128, DS_I3 = 144, DS_BF = 160, DS_IF = 192, DS_IH = 224, DS_BH = 240, DS_CONST_FLOATS = 256;
constexpr int DS_SLAB = 4 * 80 * 16, DS_GP = 9, DS_G_FLOATS = 344;            // G window: 38 rows x 9 floats (342), padded
constexpr int DS_WAVE_BYTES = DS_SLAB + DS_G_FLOATS * 4;
constexpr int DS_SHARED_BYTES = DS_CONST + DS_CONST_FLOATS * 4;
constexpr size_t DS_LDS = (size_t)DS_SHARED_BYTES + (size_t)DS_WAVES * DS_WAVE_BYTES;
static_assert(DS_LDS <= 160 * 1024 && DS_SHARED_BYTES % 16 == 0 && DS_WAVE_BYTES % 16 == 0, "one 16-wave workgroup per CU");

__global__ __launch_bounds__(64 * DS_WAVES) void dec_stream_kernel(const DecStreamParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ds_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kq = lane >> 4;

    // ---- weights and constants -> LDS (the only workgroup-wide step)
    {
        u32x4_t* d = reinterpret_cast<u32x4_t*>(ds_smem);
        for (int i = tid; i < 32768 / 16; i += 64 * DS_WAVES) d[DS_WU / 16 + i] = reinterpret_cast<const u32x4_t*>(p.wuf)[i];
        for (int i = tid; i < 6144 / 16; i += 64 * DS_WAVES) d[DS_W3 / 16 + i] = reinterpret_cast<const u32x4_t*>(p.w3f)[i];
        for (int i = tid; i < 8192 / 16; i += 64 * DS_WAVES) d[DS_WF / 16 + i] = reinterpret_cast<const u32x4_t*>(p.wff)[i];
        for (int i = tid; i < 2048 / 16; i += 64 * DS_WAVES) d[DS_WH / 16 + i] = reinterpret_cast<const u32x4_t*>(p.whf)[i];
        float* cs = reinterpret_cast<float*>(ds_smem + DS_CONST);
        for (int e = tid; e < DS_CONST_FLOATS; e += 64 * DS_WAVES) {
            float v;
            if (e < DS_IU) v = p.bu[e - DS_BU];
            else if (e < DS_B3) v = p.winvu[e - DS_IU];
            else if (e < DS_I3) v = p.b3[e - DS_B3];
            else if (e < DS_BF) v = p.winv3[e - DS_I3];
            else if (e < DS_IF) v = p.bf[e - DS_BF];
            else if (e < DS_IH) v = p.winvf[e - DS_IF];
            else if (e < DS_BH) v = p.winvh[e - DS_IH];
            else v = e == DS_BH ? p.bh[0] : 0.f;
            cs[e] = v;
        }
    }
    __syncthreads();

    const unsigned char* wu_l = ds_smem + DS_WU + lane * 16;
    const unsigned char* w3_l = ds_smem + DS_W3 + lane * 16;
    const unsigned char* wf_l = ds_smem + DS_WF + lane * 16;
    const unsigned char* wh_l = ds_smem + DS_WH + lane * 16;
    const float* cs = reinterpret_cast<const float*>(ds_smem + DS_CONST);
    const float* c_l = cs + 4 * kq;
    unsigned char* slab = ds_smem + DS_SHARED_BYTES + wave * DS_WAVE_BYTES;
    float* G = reinterpret_cast<float*>(slab + DS_SLAB);
    unsigned char* m_l = slab + (kq * 80 + 4 + li) * 16;         // unit (pl 0, g 0, row li) of this lane's kq block
    auto munit = [](int pl, int g, int row) { return (pl * 34 + g * 17 + row) * 16; };
    // the row copies move 16 units = (kq, plane, g): lane l < 16
    unsigned char* h_main = slab + ((lane >> 2) * 80 + 4 + ((lane >> 1) & 1) * 34 + (lane & 1) * 17) * 16;   // row 0 of (pl, g)
    unsigned char* h_xp = slab + ((lane >> 2) * 80 + ((lane >> 1) & 1) * 2 + (lane & 1)) * 16;
    unsigned char* h_eh = h_xp + 72 * 16;
    auto copy16 = [](unsigned char* dst, const unsigned char* src) { *reinterpret_cast<u32x4_t*>(dst) = *reinterpret_cast<const u32x4_t*>(src); };

    const int sid = blockIdx.x * DS_WAVES + wave;
    if (sid >= p.B * p.segs_per_clip) return;
    const int b = sid / p.segs_per_clip, seg = sid - b * p.segs_per_clip;
    const int nchunks = (p.L + 15) / 16;
    const int c_first = seg * p.seg_chunks;
    const int c_last = c_first + p.seg_chunks < nchunks ? c_first + p.seg_chunks : nchunks;
    if (c_first >= c_last) return;
    const int T = 2 * p.L;

    // ---- split16 scales of this clip (dec_tail.h; the head's operand ELU(v) from the bound of v)
    const unsigned am = *amax_at(p.amax_x, b);
    const int exs = s16_exponent(am);
    const float Ub = __fmaf_rn(p.ub1, __uint_as_float(am), p.ub0) * 1.0000005f;
    const float Hb = __fmaf_rn(p.hb1, Ub, p.hb0) * 1.0000005f;
    const float Vb = __fmaf_rn(p.fb1h, Hb, __fmaf_rn(p.fb1x, Ub, p.fb0)) * 1.000001f;
    const int eu = es_exp(Ub), eh = es_exp(Hb), eb = eh < eu ? eh : eu, ev = es_exp(Vb);
    const float sxs = s16_pow2(exs), ixs = s16_pow2(-exs);     // xe in the transposed conv (its amax is exact)
    const float su = s16_pow2(eu), iu = s16_pow2(-eu);         // ELU(u) in the k3 conv
    const float sb = s16_pow2(eb), ib = s16_pow2(-eb);         // hidden and raw u share stage B's accumulator
    const float sv = s16_pow2(ev), iv = s16_pow2(-ev);         // ELU(v) in the head

    const int clip_bytes = p.L * 256;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.xe + (long long)b * p.L * 64), 0, clip_bytes, 0x00020000);
    // rows m .. m + 15 in operand shape (rb_stream6.h): r[kc][h] = channels 32 kc + 16 h + 4 kq + {0..3} of row m + li; rows past the clip: zeros
    auto request = [&](int m, f32x4 (&r)[2][2]) {
        const int row = m + li;
        const int ro = row < p.L ? row * 256 + kq * 16 : 0x7fff0000;
#pragma unroll
        for (int kc = 0; kc < 2; ++kc)
#pragma unroll
            for (int h = 0; h < 2; ++h) r[kc][h] = bufload16(rs, ro + kc * 128 + h * 64, 0);
    };

    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const u32x4_t z16 = {0u, 0u, 0u, 0u};
    const long long ob = (long long)b * T;
    int ch = c_first > 0 ? c_first - 1 : 0;                    // a segment inside a clip warms its halos up on the chunk before
    f32x4 rx[2][2];
    request(ch * 16, rx);
    if (lane < 16) *reinterpret_cast<u32x4_t*>(h_xp) = z16;     // x[m0 - 1] = 0 at a clip start (and at the start of a warm-up chunk, whose first rows nothing reads)
    for (; ch < c_last; ++ch) {
        const int m0 = ch * 16, t0 = 2 * m0;
        const bool emit = ch >= c_first;
        // ---- stage the input rows (split once, the clip's scale) -> region rows 1..16; row 0 = the previous chunk's last row
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            const Hl8 e = split16_regs8(rx[kc][0], rx[kc][1], sxs);
            *reinterpret_cast<f16x8*>(m_l + munit(0, kc, 1)) = e.hi;
            *reinterpret_cast<f16x8*>(m_l + munit(1, kc, 1)) = e.lo;
        }
        if (lane < 16) copy16(h_main, h_xp);
        request(ch + 1 < c_last ? m0 + 16 : 0x3fffff00, rx);    // the next chunk's rows: in flight during the whole chunk
        __builtin_amdgcn_sched_barrier(0);                      // (the requests stay HERE: rb_stream6.h)

        // ---- stage U: u[2 m + ph] = Wp[ph] * [xe[m-1] | xe[m]] + b; accumulator tile c = (phase c >> 1, channels 16 (c & 1) ..)
        Hl8 ur[2], hf[2];
        {
            f32x4 acc[4] = {zero4, zero4, zero4, zero4};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {                     // k = tap j * 64 + channel: k-step (j, kc)
                const f16x8 xh = *reinterpret_cast<const f16x8*>(m_l + munit(0, ks & 1, ks >> 1));
                const f16x8 xl = *reinterpret_cast<const f16x8*>(m_l + munit(1, ks & 1, ks >> 1));
                f16x8 wh[4], wl[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    wh[c] = *reinterpret_cast<const f16x8*>(wu_l + ((c * 4 + ks) * 2 + 0) * 1024);
                    wl[c] = *reinterpret_cast<const f16x8*>(wu_l + ((c * 4 + ks) * 2 + 1) * 1024);
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = es_mfma(wl[c], xh, acc[c]);
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = es_mfma(wh[c], xl, acc[c]);
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = es_mfma(wh[c], xh, acc[c]);
            }
            if (lane < 16) copy16(h_xp, h_main + 16 * 16);      // the next chunk's x[m0 - 1], before ELU(u) takes the region
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
                f32x4 u[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int c = 2 * ph + q;
                    const f32x4 buv = *reinterpret_cast<const f32x4*>(c_l + DS_BU + 16 * c);
                    const f32x4 iuv = *reinterpret_cast<const f32x4*>(c_l + DS_IU + 16 * c) * ixs;
                    u[q] = es_fma4(acc[c], iuv, buv);
                    const int t = t0 + 2 * li + ph;
                    if (p.dbg_u && emit && t < T) *reinterpret_cast<f32x4*>(p.dbg_u + (ob + t) * 32 + 16 * q + 4 * kq) = u[q];
                }
                const Hl8 e = split16_regs8(elu4p(u[0]), elu4p(u[1]), su);
                *reinterpret_cast<f16x8*>(m_l + munit(0, ph, 1)) = e.hi;
                *reinterpret_cast<f16x8*>(m_l + munit(1, ph, 1)) = e.lo;
                ur[ph] = split16_regs8(u[0], u[1], sb);
            }
        }
        // ELU(u) row r = 0 of each phase: the previous chunk's last row; at the clip start the k3 conv's reflect padding ([HF]:157-176)
        // ue[-1] = ue[1], ue[-2] = ue[2]:  (phase 1, r 0) <- (phase 1, r 1),  (phase 0, r 0) <- (phase 0, r 2)
        if (lane < 16) copy16(h_main, t0 == 0 ? h_main + (2 - (lane & 1)) * 16 : h_eh);

        // ---- stage A: hidden = ELU(W3 * [ue(t-2) | ue(t-1) | ue(t)] + b3), row tile = phase: the taps of row 2 li + ph are
        //      ph 0: (0, li - 1) (1, li - 1) (0, li);   ph 1: (1, li - 1) (0, li) (1, li)     as (phase, lane) = region row li / li + 1
        {
            f16x8 eh_[2][2], el_[2][2];                          // [phase][r - li]
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    eh_[g][d] = *reinterpret_cast<const f16x8*>(m_l + munit(0, g, d));
                    el_[g][d] = *reinterpret_cast<const f16x8*>(m_l + munit(1, g, d));
                }
            f32x4 aH[2] = {zero4, zero4}, aL[2] = {zero4, zero4};
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const f16x8 w3h = *reinterpret_cast<const f16x8*>(w3_l + (j * 2 + 0) * 1024);
                const f16x8 w3l = *reinterpret_cast<const f16x8*>(w3_l + (j * 2 + 1) * 1024);
#pragma unroll
                for (int ph = 0; ph < 2; ++ph) {
                    const int q = ph + j;                        // steps back from (ph, li): 2 - j rows; q = 0 .. 3 walks (0,li-1) (1,li-1) (0,li) (1,li)
                    aL[ph] = es_mfma(w3l, eh_[q & 1][q >> 1], aL[ph]);
                }
#pragma unroll
                for (int ph = 0; ph < 2; ++ph) {
                    const int q = ph + j;
                    aH[ph] = es_mfma(w3h, eh_[q & 1][q >> 1], aH[ph]);
                }
#pragma unroll
                for (int ph = 0; ph < 2; ++ph) {
                    const int q = ph + j;
                    aL[ph] = es_mfma(w3h, el_[q & 1][q >> 1], aL[ph]);
                }
            }
            if (lane < 16) copy16(h_eh, h_main + 16 * 16);      // the next chunk's ELU(u) rows r = 0
            const f32x4 b3v = *reinterpret_cast<const f32x4*>(c_l + DS_B3);
            const f32x4 i3v = *reinterpret_cast<const f32x4*>(c_l + DS_I3) * iu;
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) hf[ph] = split16_regs8(elu4p(es_fma4(aH[ph] + aL[ph], i3v, b3v)), zero4, sb);
        }

        // ---- stage B: v = [W1 | Ws] * [hidden | u] + bf;  head: G[row][tap] = ELU(v)[row][:] . wh[tap][:]
        {
            f32x4 acc[2][2] = {{zero4, zero4}, {zero4, zero4}};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                f16x8 wh[2], wl[2];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    wh[c] = *reinterpret_cast<const f16x8*>(wf_l + ((c * 2 + ks) * 2 + 0) * 1024);
                    wl[c] = *reinterpret_cast<const f16x8*>(wf_l + ((c * 2 + ks) * 2 + 1) * 1024);
                }
#pragma unroll
                for (int ph = 0; ph < 2; ++ph)
#pragma unroll
                    for (int c = 0; c < 2; ++c) acc[ph][c] = es_mfma(wl[c], ks ? ur[ph].hi : hf[ph].hi, acc[ph][c]);
#pragma unroll
                for (int ph = 0; ph < 2; ++ph)
#pragma unroll
                    for (int c = 0; c < 2; ++c) acc[ph][c] = es_mfma(wh[c], ks ? ur[ph].lo : hf[ph].lo, acc[ph][c]);
#pragma unroll
                for (int ph = 0; ph < 2; ++ph)
#pragma unroll
                    for (int c = 0; c < 2; ++c) acc[ph][c] = es_mfma(wh[c], ks ? ur[ph].hi : hf[ph].hi, acc[ph][c]);
            }
            const f16x8 whh = *reinterpret_cast<const f16x8*>(wh_l);
            const f16x8 whl = *reinterpret_cast<const f16x8*>(wh_l + 1024);
            const float ihv = cs[DS_IH + li] * iv;              // this lane's tap column: 2^-s of its weight row x 2^-s of ELU(v)
            f32x4 bfv[2], ifv[2];                               // (read once per chunk)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                bfv[c] = *reinterpret_cast<const f32x4*>(c_l + DS_BF + 16 * c);
                ifv[c] = *reinterpret_cast<const f32x4*>(c_l + DS_IF + 16 * c) * ib;
            }
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
                f32x4 v[2];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    v[c] = es_fma4(acc[ph][c], ifv[c], bfv[c]);
                    const int t = t0 + 2 * li + ph;
                    if (p.dbg_v && emit && t < T) *reinterpret_cast<f32x4*>(p.dbg_v + (ob + t) * 32 + 16 * c + 4 * kq) = v[c];
                }
                const Hl8 vo = split16_regs8(elu4p(v[0]), elu4p(v[1]), sv);
                // A = ELU(v) (M = this phase's 16 rows), B = head weights (N = taps): lane (tap, kq) gets rows 4 kq + r
                f32x4 g = es_mfma(vo.lo, whh, zero4);
                g = es_mfma(vo.hi, whl, g);
                g = es_mfma(vo.hi, whh, g);
                if (li < 8) {                                    // window row = time - t0 + 6 = 2 (4 kq + r) + ph + 6
                    float* gw = G + (8 * kq + ph + 6) * DS_GP + li;
                    gw[0 * 2 * DS_GP] = g.x * ihv; gw[1 * 2 * DS_GP] = g.y * ihv; gw[2 * 2 * DS_GP] = g.z * ihv; gw[3 * 2 * DS_GP] = g.w * ihv;
                }
            }
        }
        // reflect padding of the head at the clip start: ve[-i] = ve[i], i = 1..6  (window rows 6 - i <- 6 + i)
        if (t0 == 0 && lane < 48) {
            const int i = 1 + (lane >> 3), j = lane & 7;
            G[(6 - i) * DS_GP + j] = G[(6 + i) * DS_GP + j];
        }
        // ---- sig[t0 + tl] = bh + sum_j G[tl + j][j]  (taps ascending)
        {
            const int tl = lane & 31;
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < 7; ++j) acc += G[(tl + j) * DS_GP + j];
            if (emit && lane < 32 && t0 + tl < T) p.sig[ob + t0 + tl] = cs[DS_BH] + acc;
        }
        // halo of the next chunk's head: the last six rows of the window
        if (lane < 48) {
            const int r = lane >> 3, j = lane & 7;
            G[r * DS_GP + j] = G[(32 + r) * DS_GP + j];
        }
    }
}

}  // namespace ac
