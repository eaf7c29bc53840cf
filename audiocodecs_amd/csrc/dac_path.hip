// DAC: model plan, weight packing and the launch sequences of encode / decode.
// One translation unit of the library (core.h has the map): owns the DAC kernels of dac.h and ac_dac_create.  Algorithm: descript-audio-codec 1.0.0
// (dac/model/dac.py, dac/nn/layers.py, dac/nn/quantize.py) as /root/reference/audiocodecs/dac.py calls it;
// that package is not on disk, line cites below are to the same-architecture [HF] transformers
// models/dac/modeling_dac.py (PARITY UNPINNED w.r.t. the reference, oracle/dac_oracle.py header).
//
//   encoder ([HF]:444-474): conv k7 (pad 3) | per stride s: 3 x ResidualUnit(dilation 1,3,9), Snake, conv k=2s
//           stride s pad ceil(s/2) | Snake, conv k3 (pad 1)
//   ResidualUnit ([HF]:175-209): x + conv_k1(Snake(conv_k7,dilated(Snake(x))))   -- full width, no bottleneck
//   decoder ([HF]:407-441): conv k7 | per stride s: Snake, convT k=2s stride s pad ceil(s/2), 3 x ResidualUnit |
//           Snake, conv k7, tanh
// Every conv is the tap-GEMM again: symmetric zero padding is the segment's `pad`, dilation reloads the A slab
// per tap, Snake (per-channel alpha of the CONSUMING layer) is the activated flavour written by the producer's
// epilogue, the padded transposed conv writes shifted rows with a range mask.
#include "core.h"
#include "dac.h"
#include "dac_unit6.h"

namespace acimpl {

static_assert(DAC_D == DAC_CODE_DIM, "core.h mirrors dac.h");

struct SnakeP {
    const float* a = nullptr;
    const float* ai = nullptr;
    int n = 0;
};

SnakeP dac_snake(const ac_handle* h, size_t a_off, size_t ai_off, int n) { return SnakeP{h->blob + a_off, h->blob + ai_off, n}; }

// conv with symmetric zero padding `padl` (time steps) and dilation; stride 1 (any k <= 8) or k = 2*stride.
int dac_conv(ac_handle* h, hipStream_t st, const PackedGemm& g, const Act& x, int k, int s, int dil, int padl, Out out, SnakeP sn, int B,
             Act2* y, const Epi& epi = Epi{}, int tanh_out = 0) {
    if (s != 1 && k != 2 * s) return fail(h, AC_EINVAL, "strided conv needs kernel == 2*stride (got k=%d, s=%d)", k, s);
    const long long span = (long long)x.L + 2LL * padl - (long long)dil * (k - 1) - 1;
    if (span < 0)
        return fail(h, AC_EINVAL, "input too short: %d time steps (+%d padding) for a kernel spanning %d", x.L, 2 * padl, dil * (k - 1) + 1);
    const int M = (int)(span / s) + 1;
    TapGemmParams p{};
    p.nseg = 1;
    p.seg[0] = make_seg(x, s, s == 1 ? k : 2, PAD_ZERO, 0, 0, nullptr);
    p.seg[0].pad = padl;
    p.seg[0].dil = dil;
    p.w = h->blob + g.w_off;
    p.bias = g.has_bias ? h->blob + g.b_off : nullptr;
    p.y = out.raw;
    p.y_elu = out.elu;
    p.y_bs = (long long)M * g.N;
    p.y_rs = g.N;
    p.B = B;
    p.M = M;
    p.N = g.N;
    p.Ktot = g.Ktot;
    p.res = epi.res;
    p.res_bs = epi.res_bs;
    p.res_rs = epi.res_rs;
    p.alpha = sn.a;
    p.alpha_inv = sn.ai;
    p.alpha_n = sn.n;
    p.tanh_out = tanh_out;
    const int rc = run_tap(h, st, p);
    if (y) {
        y->raw = Act{out.raw, p.y_bs, p.y_rs, M, g.N, p.amax_out, p.B};
        y->elu = Act{out.elu, p.y_bs, p.y_rs, M, g.N, p.amax_out, p.B};
    }
    return rc;
}

// transposed conv k = 2s, stride s, padding pp: rows m' = 0..L of s*cout floats, row m' = [x[m'-1] | x[m']] * Wp,
// holding output samples m'*s - pp .. m'*s - pp + s - 1; only samples in [0, Lout) are stored.
int dac_convtr(ac_handle* h, hipStream_t st, const PackedGemm& g, const Act& x, int s, int pp, Out out, SnakeP sn, int B, Act2* y) {
    const int cout = g.N / s;
    const int Lout = (x.L - 1) * s - 2 * pp + 2 * s;
    TapGemmParams p{};
    p.nseg = 1;
    p.seg[0] = make_seg(x, 1, 2, PAD_ZERO, 0, 0, nullptr);
    p.w = h->blob + g.w_off;
    p.bias = h->blob + g.b_off;
    p.y = out.raw;
    p.y_elu = out.elu;
    p.y_bs = (long long)Lout * cout;
    p.y_rs = g.N;
    p.y_off = -(long long)pp * cout;
    p.y_len = (long long)Lout * cout;
    p.B = B;
    p.M = x.L + 1;
    p.N = g.N;
    p.Ktot = g.Ktot;
    p.alpha = sn.a;
    p.alpha_inv = sn.ai;
    p.alpha_n = sn.n;
    const int rc = run_tap(h, st, p);
    y->raw = Act{out.raw, p.y_bs, cout, Lout, cout, p.amax_out, p.B};
    y->elu = Act{out.elu, p.y_bs, cout, Lout, cout, p.amax_out, p.B};
    return rc;
}

// the 96-channel ResidualUnit as ONE kernel (dac_unit6.h); returns false when the shape, the alignment or the arithmetic mode
// asks for the two tap-GEMM launches (then nothing was launched)
bool dac_unit_fused(ac_handle* h, hipStream_t st, const DacResUnitPlan& ru, int C, const Act2& x, Out out, SnakeP next, int B, Act2* y, int* rc_out) {
    *rc_out = AC_OK;
    // (96 channels only: measured per unit, 32 clips -- 14.5 -> 13.9 ms; the 64-channel units LOSE, 6.9 -> 8.7 ms: their two-launch path runs
    //  64 x 32 wave tiles at three workgroups per CU, the wave-local second product needs 32 x 64 ones; profiles/r4_variants.md.
    //  The developer switches that select a tap-GEMM code path -- slab reload per tap, staged epilogue -- mean the tap-GEMM.)
    if (h->gemm_fp32 || h->dbg || h->dev.dac_unit == 0 || h->dev.tap_dil == 0 || h->dev.tap_epi_staged != 0 || C != 96 || ru.c7.N != C || ru.c1.N != C || ru.c1.Ktot != C || ru.c7.Ktot != 7 * C) return false;
    if (!ru.c7.has_bias || !ru.c1.has_bias || !out.elu || !next.a || next.n < C) return false;
    auto w1 = h->w6_of.find(ru.c7.w_off), w2 = h->w6_of.find(ru.c1.w_off);
    auto i1 = h->winv_of.find(ru.c7.w_off), i2 = h->winv_of.find(ru.c1.w_off);
    if (w1 == h->w6_of.end() || w2 == h->w6_of.end() || i1 == h->winv_of.end() || i2 == h->winv_of.end()) return false;
    const Act& xe = x.elu;
    const Act& xr = x.raw;
    const int dil = ru.dil, L = xe.L;
    if (!xe.p || !xr.p || xe.ts != C || xr.ts != C || xe.C != C || L < 1 || 6 * dil > T6_DIL_HALO || !aligned16(xe.p) || !aligned16(xr.p) || !aligned16(out.elu) ||
        (out.raw && !aligned16(out.raw)) || (xe.bs % 4) || (xr.bs % 4) || (long long)L * C * 4 >= 0x7fffffffLL)
        return false;
    DacUnitParams q{};
    TapGemmParams& p = q.g;
    p.nseg = 1;
    p.seg[0] = make_seg(xe, 1, 7, PAD_ZERO, 0, 0, nullptr);
    p.seg[0].pad = 3 * dil;
    p.seg[0].dil = dil;
    p.seg[0].amax = amax_of(h, st, xe.p, xe.bs, xe.ts, xe.L, xe.C, B, xe.amax_n == B ? xe.amax : nullptr);
    if (!p.seg[0].amax) { *rc_out = fail(h, AC_ESTATE, "out of amax slots (split16.h)"); return true; }
    p.w = h->blob + ru.c7.w_off;
    p.bias = h->blob + ru.c7.b_off;
    p.winv = h->blob + i1->second;
    p.y = out.raw;
    p.y_elu = out.elu;
    p.y_bs = (long long)L * C;
    p.y_rs = C;
    p.B = B;
    p.M = L;                                    // "same" padding: 3 dil each side of a k7 conv with dilation dil
    p.N = C;
    p.Ktot = 7 * C;
    p.res = xr.p;
    p.res_bs = xr.bs;
    p.res_rs = xr.ts;
    p.alpha = next.a;
    p.alpha_inv = next.ai;
    p.alpha_n = next.n;
    p.amax_out = amax_new(h);
    if (!p.amax_out) { *rc_out = fail(h, AC_ESTATE, "out of amax slots (split16.h)"); return true; }
    p.epi_direct = 1;
    p.mtiles = cdiv(L, 128);
    p.ntiles = 1;
    p.stagger = 0;
    p.clk = nullptr;
    q.a_mid = h->blob + ru.a2;
    q.a_mid_inv = h->blob + ru.a2i;
    q.bias2 = h->blob + ru.c1.b_off;
    q.winv2 = h->blob + i2->second;
    const __bf16* wp1 = reinterpret_cast<const __bf16*>(h->blob + w1->second);
    const __bf16* wp2 = reinterpret_cast<const __bf16*>(h->blob + w2->second);
    const long long blocks = (long long)B * p.mtiles;
    const double flops = 2.0 * B * (double)L * C * (7.0 * C + C);
    const double bytes = (double)B * L * C * 4.0 * (2 + (out.raw ? 1 : 0) + 1);
#define DAC_UNIT_LAUNCH(WN, HALO, NAME)                                                                                              \
    do {                                                                                                                            \
        const size_t lds = DacUnitCfg<WN>::lds_bytes<HALO>();                                                                       \
        if ((*rc_out = ensure_lds(h, reinterpret_cast<const void*>(dac_unit6_kernel<WN, HALO>), lds))) return true;                \
        ProfScope ps(h, st, NAME, flops, bytes);                                                                                   \
        hipLaunchKernelGGL((dac_unit6_kernel<WN, HALO>), dim3((unsigned)blocks), dim3(256), lds, st, q, wp1, wp2);                  \
    } while (0)
    if (dil == 1) DAC_UNIT_LAUNCH(3, 7, "dac_unit6_kernel<3>"); else DAC_UNIT_LAUNCH(3, T6_DIL_HALO, "dac_unit6_kernel<3, dil>");
#undef DAC_UNIT_LAUNCH
    if (hipGetLastError() != hipSuccess) { *rc_out = fail(h, AC_ESTATE, "dac_unit6 launch failed"); return true; }
    y->raw = Act{out.raw, p.y_bs, C, L, C, p.amax_out, B};
    y->elu = Act{out.elu, p.y_bs, C, L, C, p.amax_out, B};
    return true;
}

// one ResidualUnit; `next`: Snake of whatever consumes the unit's output; `want_raw`: the next layer is another unit
int dac_res_unit(ac_handle* h, hipStream_t st, const DacResUnitPlan& ru, int C, const Act2& x, WsPtrs& ws, SnakeP next, bool want_raw, int B,
                 Act2* y) {
    {
        Out o{want_raw ? ws.take() : nullptr, ws.take()};
        int rc = AC_OK;
        if (dac_unit_fused(h, st, ru, C, x, o, next, B, y, &rc)) return rc;
        if (o.raw) ws.give(o.raw);
        ws.give(o.elu);
    }
    float* hb = ws.take();
    Act2 hv;
    int rc = dac_conv(h, st, ru.c7, x.elu, 7, 1, ru.dil, 3 * ru.dil, Out{nullptr, hb}, dac_snake(h, ru.a2, ru.a2i, C), B, &hv);
    if (rc) return rc;
    Epi e;
    e.res = x.raw.p;
    e.res_bs = x.raw.bs;
    e.res_rs = x.raw.ts;
    rc = dac_conv(h, st, ru.c1, hv.elu, 1, 1, 1, 0, Out{want_raw ? ws.take() : nullptr, ws.take()}, next, B, y, e);
    ws.give(hb);
    return rc;
}

int dac_ceil_half(int s) { return (s + 1) / 2; }

int dac_num_frames(const ac_dac_config& c, long long T) {
    long long L = T;
    for (int i = 0; i < c.num_ratios; ++i) {
        const int s = c.downsampling_ratios[i];
        const long long span = L + 2LL * dac_ceil_half(s) - 2LL * s;
        if (span < 0) return 0;
        L = span / s + 1;
    }
    return (int)L;
}

long long dac_num_samples(const ac_dac_config& c, long long N) {
    long long L = N;
    for (int i = 0; i < c.num_ratios; ++i) {
        const int s = c.upsampling_ratios[i];
        L = (L - 1) * s - 2LL * dac_ceil_half(s) + 2LL * s;
    }
    return L;
}

// floats of the widest activation per clip (encoder for T samples, or decoder for N frames)
size_t dac_act_floats(const ac_handle* h, long long T_in, long long N_frames, bool enc) {
    const ac_dac_config& c = h->dcfg;
    size_t mx = 0;
    if (enc) {
        long long L = T_in;
        int ch = c.encoder_hidden_size;
        mx = std::max(mx, (size_t)L * ch);
        for (int i = 0; i < c.num_ratios; ++i) {
            const int s = c.downsampling_ratios[i];
            L = std::max<long long>((L + 2LL * dac_ceil_half(s) - 2LL * s) / s + 1, 1);
            ch *= 2;
            mx = std::max(mx, (size_t)L * ch);
        }
    } else {
        long long L = N_frames;
        int ch = c.decoder_hidden_size;
        mx = std::max(mx, (size_t)L * std::max(ch, h->dac.H));
        for (int i = 0; i < c.num_ratios; ++i) {
            const int s = c.upsampling_ratios[i];
            L = (L - 1) * s - 2LL * dac_ceil_half(s) + 2LL * s;
            ch /= 2;
            mx = std::max(mx, (size_t)L * ch);
        }
    }
    return mx;
}

constexpr size_t DAC_WS_CAP_BYTES = 40ull << 30;   // clips are processed in chunks that keep the 6 buffers under this

int dac_chunk_clips(const ac_handle* h, int B, long long T_in, long long N_frames, bool enc) {
    const size_t per_clip = NACT * dac_act_floats(h, T_in, N_frames, enc) * sizeof(float);
    const size_t fit = std::max<size_t>(1, DAC_WS_CAP_BYTES / std::max<size_t>(per_clip, 1));
    return (int)std::min<size_t>((size_t)B, fit);
}

Workspace dac_plan_ws(const ac_handle* h, int B, long long T_in, long long N_frames, bool enc) {
    Workspace w;
    const int Bc = dac_chunk_clips(h, B, T_in, N_frames, enc);
    w.act_floats = align_up(dac_act_floats(h, T_in, N_frames, enc) * Bc, 64);
    w.total_bytes = NACT * w.act_floats * sizeof(float) + 256;
    add_pool(w, Bc, 0);                   // amax slots for one chunk of clips (every chunk starts a fresh pass)
    return w;
}

// sig [B][T] -> z [B][N][H] (encoder output, what DAC._sig_to_feats returns, dac.py:109-111)
int dac_encoder_fwd(ac_handle* h, hipStream_t st, const float* sig, int B, int T, float* z, WsPtrs& ws) {
    const ac_dac_config& c = h->dcfg;
    const DacPlan& m = h->dac;
    const bool dbg = h->dbg != nullptr;
    const int F = c.encoder_hidden_size;
    const DacBlockPlan& b0 = m.enc[0];
    const SnakeP s0 = dac_snake(h, b0.ru[0].a1, b0.ru[0].a1i, F);
    Act xin{sig, (long long)T, 1, T, 1};
    Act2 x, y;
    int rc;
    if (F % 4 == 0 && F <= 64)
        rc = thin_stem(h, st, m.enc_stem, F, 7, PAD_ZERO, sig, nullptr, B, T, Out{ws.take(), ws.take()}, &x, 3, s0.a, s0.ai);
    else
        rc = dac_conv(h, st, m.enc_stem, xin, 7, 1, 1, 3, Out{ws.take(), ws.take()}, s0, B, &x);
    if (rc) return rc;
    capture(h, st, x.raw, B);
    const int nb = c.num_ratios;
    for (int i = 0; i < nb; ++i) {
        const DacBlockPlan& blk = m.enc[i];
        const int nu = (int)blk.ru.size();
        for (int u = 0; u < nu; ++u) {
            const bool last_u = u == nu - 1;
            const SnakeP next = last_u ? dac_snake(h, blk.a, blk.ai, blk.C) : dac_snake(h, blk.ru[u + 1].a1, blk.ru[u + 1].a1i, blk.C);
            rc = dac_res_unit(h, st, blk.ru[u], blk.C, x, ws, next, !last_u || dbg, B, &y);
            if (rc) return rc;
            ws.give(x);
            x = y;
            if (dbg) capture(h, st, x.raw, B);
            if (last_u && dbg) { ws.give(x.raw.p); x.raw.p = nullptr; }
        }
        const bool last = i == nb - 1;
        const int s = blk.stride, cout = blk.conv.N;
        const SnakeP next = last ? dac_snake(h, m.enc_a, m.enc_ai, cout) : dac_snake(h, m.enc[i + 1].ru[0].a1, m.enc[i + 1].ru[0].a1i, cout);
        rc = dac_conv(h, st, blk.conv, x.elu, 2 * s, s, 1, dac_ceil_half(s), Out{(!last || dbg) ? ws.take() : nullptr, ws.take()}, next, B, &y);
        if (rc) return rc;
        ws.give(x);
        x = y;
        if (!last || dbg) capture(h, st, x.raw, B);
        if (last && dbg) { ws.give(x.raw.p); x.raw.p = nullptr; }
    }
    rc = dac_conv(h, st, m.enc_final, x.elu, 3, 1, 1, 1, Out{z, nullptr}, SnakeP{}, B, &y);
    ws.give(x);
    if (rc) return rc;
    capture(h, st, y.raw, B);
    return AC_OK;
}

// z_q [B][N][H] -> sig [B][Lout]
int dac_decoder_fwd(ac_handle* h, hipStream_t st, const float* zq, int B, int N, float* sig, WsPtrs& ws) {
    const ac_dac_config& c = h->dcfg;
    const DacPlan& m = h->dac;
    const bool dbg = h->dbg != nullptr;
    const int nb = c.num_ratios;
    Act zin{zq, (long long)N * m.H, m.H, N, m.H};
    Act2 x, y;
    int rc = dac_conv(h, st, m.dec_first, zin, 7, 1, 1, 3, Out{dbg ? ws.take() : nullptr, ws.take()},
                      dac_snake(h, m.dec[0].a, m.dec[0].ai, c.decoder_hidden_size), B, &x);
    if (rc) return rc;
    ws.give(zq);
    if (dbg) { capture(h, st, x.raw, B); ws.give(x.raw.p); x.raw.p = nullptr; }
    for (int i = 0; i < nb; ++i) {
        const DacBlockPlan& blk = m.dec[i];
        const int s = blk.stride;
        rc = dac_convtr(h, st, blk.conv, x.elu, s, dac_ceil_half(s), Out{ws.take(), ws.take()}, dac_snake(h, blk.ru[0].a1, blk.ru[0].a1i, blk.C), B, &y);
        if (rc) return rc;
        ws.give(x);
        x = y;
        capture(h, st, x.raw, B);
        const int nu = (int)blk.ru.size();
        for (int u = 0; u < nu; ++u) {
            const bool last_u = u == nu - 1;
            SnakeP next;
            if (!last_u) next = dac_snake(h, blk.ru[u + 1].a1, blk.ru[u + 1].a1i, blk.C);
            else if (i + 1 < nb) next = dac_snake(h, m.dec[i + 1].a, m.dec[i + 1].ai, blk.C);
            else next = dac_snake(h, m.dec_a, m.dec_ai, blk.C);
            rc = dac_res_unit(h, st, blk.ru[u], blk.C, x, ws, next, !last_u || dbg, B, &y);
            if (rc) return rc;
            ws.give(x);
            x = y;
            if (dbg) capture(h, st, x.raw, B);
            if (last_u && dbg) { ws.give(x.raw.p); x.raw.p = nullptr; }
        }
    }
    const int F = m.dec_head.Ktot / 7;
    if (F % 4 == 0 && F <= 128 && x.elu.ts == F)
        rc = thin_head(h, st, m.dec_head, F, 7, PAD_ZERO, x.elu, B, sig, 3, 1);
    else
        rc = dac_conv(h, st, m.dec_head, x.elu, 7, 1, 1, 3, Out{sig, nullptr}, SnakeP{}, B, nullptr, Epi{}, 1);
    ws.give(x);
    return rc;
}

int dac_vq_encode(ac_handle* h, hipStream_t st, const float* z, int F, int K, long long* toks, float* qsum) {
    const DacPlan& m = h->dac;
    DacVqParams p{};
    p.z = z;
    p.win = h->blob + m.win;
    p.bin = h->blob + m.bin;
    p.wout = h->blob + m.wout;
    p.bout = h->blob + m.bout;
    p.cb = h->blob + m.cb;
    p.cbn = h->blob + m.cbn;
    p.c2 = h->blob + m.c2;
    p.toks = toks;
    p.qsum = qsum;
    p.F = F;
    p.H = m.H;
    p.C = h->dcfg.codebook_size;
    p.K = K;
    const size_t lds = dac_vq_lds_bytes(m.H);
    if (int rc = ensure_lds(h, reinterpret_cast<const void*>(dac_vq_encode_kernel), lds)) return rc;
    ProfScope ps(h, st, "dac_vq_encode_kernel", 2.0 * F * K * (2.0 * DAC_D * m.H + (double)p.C * DAC_D),
                 (double)F * m.H * 4 * (qsum ? 2 : 1) + (double)F * K * 8);
    hipLaunchKernelGGL(dac_vq_encode_kernel, dim3(cdiv(F, DAC_FR)), dim3(256), lds, st, p);
    HIPCHK(h, hipGetLastError());
    return AC_OK;
}

// quantizer.from_codes (dac.py:126-128): z_q = sum_k (out_proj_k(codebook_k[tok]) + bias), a gather over the projected table
int dac_from_codes(ac_handle* h, hipStream_t st, const long long* toks, int F, int K, float* zq) {
    const DacPlan& m = h->dac;
    RvqDecParams p{};
    p.toks = toks;
    p.e = h->blob + m.proj;
    p.out = zq;
    p.F = F;
    p.H = m.H;
    p.C = h->dcfg.codebook_size;
    p.K = K;
    p.tK = K;
    p.tk0 = 0;
    p.os = m.H;
    const long long cnt = (long long)F * (m.H / 4);
    p.bad = h->sticky_dev ? h->sticky_dev + ST_BAD_TOKEN : nullptr;
    ProfScope ps(h, st, "rvq_decode_kernel", (double)F * m.H * K, (double)F * K * 8 + (double)F * m.H * 4 * (K + 1));
    rvq_decode_launch(st, p, (unsigned)((cnt + 255) / 256));
    HIPCHK(h, hipGetLastError());
    return AC_OK;
}

// ---------------------------------------------------------------------------------------------
// weight packing: HF DacModel state-dict keys (checkpoint.dac_conv_specs is the Python twin)
// ---------------------------------------------------------------------------------------------
int dac_finalize(ac_handle* h, Packer& pk) {
    const ac_dac_config& c = h->dcfg;
    DacPlan& m = h->dac;
    const int nb = c.num_ratios;
    auto snake = [&](const std::string& name, int C, size_t& a, size_t& ai) {
        const std::vector<float>* v = pk.get(name + ".alpha", (size_t)C);
        if (!v) return false;
        a = pk.reserve(C);
        ai = pk.reserve(C);
        for (int i = 0; i < C; ++i) {
            pk.blob[a + i] = (*v)[i];
            pk.blob[ai + i] = 1.0f / ((*v)[i] + 1e-9f);       // (alpha + 1e-9).reciprocal() in fp32
        }
        return true;
    };
    auto units = [&](const std::string& pre, int C, DacBlockPlan& blk) {
        blk.C = C;
        blk.ru.resize(c.num_dilations);
        for (int u = 0; u < c.num_dilations; ++u) {
            DacResUnitPlan& ru = blk.ru[u];
            ru.dil = c.dilations[u];
            const std::string p = pre + ".res_unit" + std::to_string(u + 1);
            if (!snake(p + ".snake1", C, ru.a1, ru.a1i) || !pk.conv(ConvSpec{p + ".conv1", 0, C, C, 7, 1}, ru.c7) ||
                !snake(p + ".snake2", C, ru.a2, ru.a2i) || !pk.conv(ConvSpec{p + ".conv2", 0, C, C, 1, 1}, ru.c1))
                return false;
        }
        return true;
    };
    int ch = c.encoder_hidden_size;
    if (!pk.conv(ConvSpec{"encoder.conv1", 0, 1, ch, 7, 1}, m.enc_stem)) return pk.rc;
    m.enc.resize(nb);
    for (int i = 0; i < nb; ++i) {
        const std::string pre = "encoder.block." + std::to_string(i);
        const int s = c.downsampling_ratios[i];
        m.enc[i].stride = s;
        if (!units(pre, ch, m.enc[i]) || !snake(pre + ".snake1", ch, m.enc[i].a, m.enc[i].ai) ||
            !pk.conv(ConvSpec{pre + ".conv1", 0, ch, 2 * ch, 2 * s, s}, m.enc[i].conv))
            return pk.rc;
        ch *= 2;
    }
    m.H = ch;
    if (!snake("encoder.snake1", ch, m.enc_a, m.enc_ai) || !pk.conv(ConvSpec{"encoder.conv2", 0, ch, m.H, 3, 1}, m.enc_final)) return pk.rc;
    ch = c.decoder_hidden_size;
    if (!pk.conv(ConvSpec{"decoder.conv1", 0, m.H, ch, 7, 1}, m.dec_first)) return pk.rc;
    m.dec.resize(nb);
    for (int i = 0; i < nb; ++i) {
        const std::string pre = "decoder.block." + std::to_string(i);
        const int s = c.upsampling_ratios[i];
        m.dec[i].stride = s;
        if (!snake(pre + ".snake1", ch, m.dec[i].a, m.dec[i].ai) || !pk.convtr(ConvSpec{pre + ".conv_t1", 1, ch, ch / 2, 2 * s, s}, m.dec[i].conv) ||
            !units(pre, ch / 2, m.dec[i]))
            return pk.rc;
        ch /= 2;
    }
    if (!snake("decoder.snake1", ch, m.dec_a, m.dec_ai) || !pk.conv(ConvSpec{"decoder.conv2", 0, ch, 1, 7, 1}, m.dec_head)) return pk.rc;
    // quantiser
    const int K = c.n_codebooks, C = c.codebook_size, H = m.H, D = DAC_D;
    m.win = pk.reserve((size_t)K * D * H);
    m.bin = pk.reserve((size_t)K * D);
    m.wout = pk.reserve((size_t)K * H * D);
    m.bout = pk.reserve((size_t)K * H);
    m.cb = pk.reserve((size_t)K * C * D);
    m.cbn = pk.reserve((size_t)K * C * D);
    m.c2 = pk.reserve((size_t)K * C);
    m.proj = pk.reserve((size_t)K * C * H);
    for (int k = 0; k < K; ++k) {
        const std::string q = "quantizer.quantizers." + std::to_string(k);
        const std::vector<float>* wi = pk.get(q + ".in_proj.weight", (size_t)D * H);
        const std::vector<float>* bi = pk.get(q + ".in_proj.bias", (size_t)D);
        const std::vector<float>* wo = pk.get(q + ".out_proj.weight", (size_t)H * D);
        const std::vector<float>* bo = pk.get(q + ".out_proj.bias", (size_t)H);
        const std::vector<float>* cb = pk.get(q + ".codebook.weight", (size_t)C * D);
        if (!wi || !bi || !wo || !bo || !cb) return pk.rc;
        std::copy(wi->begin(), wi->end(), pk.blob.begin() + m.win + (size_t)k * D * H);
        std::copy(bi->begin(), bi->end(), pk.blob.begin() + m.bin + (size_t)k * D);
        std::copy(wo->begin(), wo->end(), pk.blob.begin() + m.wout + (size_t)k * H * D);
        std::copy(bo->begin(), bo->end(), pk.blob.begin() + m.bout + (size_t)k * H);
        std::copy(cb->begin(), cb->end(), pk.blob.begin() + m.cb + (size_t)k * C * D);
        std::vector<float> cn((size_t)C * D);
        for (int code = 0; code < C; ++code) {          // F.normalize(codebook): x / max(|x|_2, 1e-12), fp32
            float ss = 0.f;
            for (int d = 0; d < D; ++d) ss += (*cb)[(size_t)code * D + d] * (*cb)[(size_t)code * D + d];
            const float dn = std::max(std::sqrt(ss), 1e-12f);
            float s2 = 0.f;
            for (int d = 0; d < D; ++d) {
                const float v = (*cb)[(size_t)code * D + d] / dn;
                cn[(size_t)code * D + d] = v;
                s2 += v * v;
            }
            pk.blob[m.c2 + (size_t)k * C + code] = s2;
        }
        for (int t = 0; t < C / 16; ++t)
            for (int hh = 0; hh < 2; ++hh)
                for (int l = 0; l < 64; ++l)
                    pk.blob[m.cbn + (size_t)k * C * D + ((size_t)t * 2 + hh) * 64 + l] = cn[(size_t)(t * 16 + (l & 15)) * D + 4 * hh + (l >> 4)];
        for (int code = 0; code < C; ++code)             // projected table: out_proj(codebook[code]) + bias
            for (int d = 0; d < H; ++d) {
                const float* w = &(*wo)[(size_t)d * D];
                const float* qv = &(*cb)[(size_t)code * D];
                float acc = w[0] * qv[0];
                for (int n = 1; n < D; ++n) acc = std::fmaf(w[n], qv[n], acc);
                pk.blob[m.proj + ((size_t)k * C + code) * H + d] = acc + (*bo)[d];
            }
    }
    // quantizers[0].in_proj as a 1-tap conv (DAC._sig_to_feats with latent=True, dac.py:104-108)
    m.in_proj0.N = D;
    m.in_proj0.Ktot = H;
    m.in_proj0.has_bias = true;
    m.in_proj0.w_off = m.win;
    m.in_proj0.b_off = m.bin;
    return AC_OK;
}

// in_proj of the first codebook alone (dac.py:103-112, _sig_to_feats with latent=True)
int dac_latent_proj(ac_handle* h, hipStream_t st, const float* z, int N, int nb, float* zlat) {
    const int H = h->dac.H;
    Act za{z, (long long)N * H, H, N, H};
    return dac_conv(h, st, h->dac.in_proj0, za, 1, 1, 1, 0, Out{zlat, nullptr}, SnakeP{}, nb, nullptr);
}

}  // namespace acimpl

extern "C" int ac_dac_create(const ac_dac_config* cfg, ac_handle** out) {
    if (!cfg || !out) return AC_EINVAL;
    *out = nullptr;
    if (cfg->struct_size != (int32_t)sizeof(ac_dac_config)) return AC_EINVAL;
    const ac_dac_config& c = *cfg;
    if (c.num_ratios < 1 || c.num_ratios > AC_MAX_RATIOS || c.encoder_hidden_size < 1 || c.decoder_hidden_size < (1 << c.num_ratios) ||
        c.decoder_hidden_size % (1 << c.num_ratios) || c.n_codebooks < 1 || c.codebook_size % 64 || c.codebook_size < 64 ||
        c.codebook_dim != DAC_D || c.num_dilations < 1 || c.num_dilations > AC_MAX_DILATIONS)
        return AC_EINVAL;
    const int H = c.encoder_hidden_size << c.num_ratios;
    if (H % 64 || H > 1024) return AC_EINVAL;
    for (int i = 0; i < c.num_dilations; ++i)
        if (c.dilations[i] < 1 || 6 * c.dilations[i] > GEN_EXTRA) return AC_EINVAL;
    for (int i = 0; i < c.num_ratios; ++i)
        if (c.downsampling_ratios[i] < 1 || c.upsampling_ratios[i] < 1) return AC_EINVAL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= c.device || c.device < 0) return AC_ENODEV;
    ac_handle* h = new (std::nothrow) ac_handle();
    if (!h) return AC_ENOMEM;
    h->arch = ARCH_DAC;
    h->dcfg = c;
    h->hop = 1;
    for (int i = 0; i < c.num_ratios; ++i) h->hop *= c.downsampling_ratios[i];
    h->D = H;
    h->dac.H = H;
    *out = h;
    return AC_OK;
}
