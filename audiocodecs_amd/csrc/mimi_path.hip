// Mimi: model plan, weight packing and the launch sequences of encode / decode.
// One translation unit of the library (core.h has the map): owns the Mimi kernels of mimi.h and ac_mimi_create.
//
//   encode ([HF] mimi :1237-1259): SEANet encoder (:444-492) -> 8-layer causal transformer (:782-928) ->
//           stride-2 replicate-padded conv (:1195-1208) -> split RVQ (:1084-1127)
//   decode ([HF] :1399-1414): split RVQ decode (:1129-1138) -> depthwise stride-2 transposed conv (:1209-1217)
//           -> transformer -> SEANet decoder (:931-961)
// Data layout: channels-last fp32 as for EnCodec; the transformer works on the [B*T][hidden] token matrix
// (the same memory), so encoder conv -> transformer -> down-sampler need no transposes (the reference
// transposes twice per transformer, :1248-1257).
#include "core.h"
#include "mimi.h"

namespace acimpl {

constexpr int MIMI_ROPE_T = 8192;   // positions tabulated (327 s at 25 Hz; HF max_position_embeddings = 8000)


// conv (stride 1 or k = 2*stride) with the given padding rule, output contiguous [B][M][N]
int mimi_conv(ac_handle* h, hipStream_t st, const PackedGemm& g, const Act& x, int k, int s, int pad, Out out, int B, Act2* y,
              const Epi& epi = Epi{}) {
    const int M = cdiv(x.L, s);
    const int extra = M * s - x.L;
    if (s != 1 && k != 2 * s) return fail(h, AC_EINVAL, "strided conv needs kernel == 2*stride (got k=%d, s=%d)", k, s);
    TapGemmParams p{};
    p.nseg = 1;
    p.seg[0] = make_seg(x, s, s == 1 ? k : 2, pad, extra, 0, nullptr);
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
    p.scale = epi.scale;
    p.res = epi.res;
    p.res_bs = epi.res_bs;
    p.res_rs = epi.res_rs;
    p.gelu = epi.gelu;
    const int rc = run_tap(h, st, p);
    if (y) {
        y->raw = Act{out.raw, p.y_bs, p.y_rs, M, g.N, p.amax_out, p.B};
        y->elu = Act{out.elu, p.y_bs, p.y_rs, M, g.N, p.amax_out, p.B};
    }
    return rc;
}


// ResBlock with identity shortcut: y = x + conv_k1(ELU(conv_k3(ELU(x))))
int mimi_resblock(ac_handle* h, hipStream_t st, const ResBlockPlan& rb, const Act2& x, float* hbuf, Out out, int B, Act2* y) {
    const ac_mimi_config& c = h->mcfg;
    const long long bs = (long long)x.raw.L * rb.C;
    if (rb128_ok(h, rb) && c.residual_kernel_size == 3 && c.compress == 2 && x.raw.ts == 128 && x.raw.bs == bs && aligned16(x.raw.p) &&
        bs * 4 < 0x70000000LL) {
        const unsigned* am = nullptr;
        int rc = rb128_identity_fwd(h, st, rb, x, out, B, &am);
        if (rc) return rc;
        HIPCHK(h, hipGetLastError());
        y->raw = Act{out.raw, bs, rb.C, x.raw.L, rb.C, am, B};
        y->elu = Act{out.elu, bs, rb.C, x.raw.L, rb.C, am, B};
        return AC_OK;
    }
    if (rb.C == 64 && c.residual_kernel_size == 3 && c.compress == 2 && x.raw.ts == rb.C && x.raw.bs == bs && aligned16(x.raw.p) &&
        (!x.elu.p || (x.elu.ts == rb.C && x.elu.bs == bs && aligned16(x.elu.p)))) {
        const unsigned* am = nullptr;
        int rc = rb64_identity_fwd(h, st, rb, x, out, B, &am);
        if (rc) return rc;
        HIPCHK(h, hipGetLastError());
        y->raw = Act{out.raw, bs, rb.C, x.raw.L, rb.C, am, B};
        y->elu = Act{out.elu, bs, rb.C, x.raw.L, rb.C, am, B};
        return AC_OK;
    }
    Act2 hv;
    int rc = mimi_conv(h, st, rb.c3, x.elu, c.residual_kernel_size, 1, PAD_ZERO, Out{nullptr, hbuf}, B, &hv);
    if (rc) return rc;
    Epi e;
    e.res = x.raw.p;
    e.res_bs = x.raw.bs;
    e.res_rs = x.raw.ts;
    return mimi_conv(h, st, rb.fused, hv.elu, 1, 1, PAD_ZERO, out, B, y, e);
}

// blocks that activate their raw input themselves: the producer writes one flavour (128 channels: the fused split-operand
// kernel only, and not while the test hook wants every module's raw output)
bool mimi_rb_self_elu(const ac_handle* h, const ac_mimi_config& c, int C) {
    return (C == 64 || (C == 128 && !h->gemm_fp32 && !h->dbg)) && c.residual_kernel_size == 3 && c.compress == 2;
}

int layernorm_fwd(ac_handle* h, hipStream_t st, const float* x, size_t w_off, size_t b_off, float* y, long long rows, int H, float eps,
                  const unsigned** rows_out) {
    if (H > 64 * LN_MAXV) return fail(h, AC_EINVAL, "hidden size %d exceeds the LayerNorm kernel limit", H);
    LayerNormParams p{x, h->blob + w_off, h->blob + b_off, y, rows, H, eps, nullptr};
    if (rows_out) {                       // row words for the linear layer that reads y (split16.h row mode); every row is written
        *rows_out = nullptr;
        if (!h->gemm_fp32 && rows <= 0x7fffffffLL) p.rowmax = rowmax_new(h, st, rows, false);
        *rows_out = p.rowmax;
    }
    ProfScope ps(h, st, "layernorm_kernel", 8.0 * rows * H, 8.0 * rows * H);
    hipLaunchKernelGGL(layernorm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, p);
    HIPCHK(h, hipGetLastError());
    return AC_OK;
}

int attention_fwd(ac_handle* h, hipStream_t st, const float* qkv, float* out, int B, int T) {
    const ac_mimi_config& c = h->mcfg;
    if (T > h->mimi.rope_T) return fail(h, AC_EINVAL, "%d transformer positions exceed the tabulated RoPE range (%d)", T, h->mimi.rope_T);
    AttnParams p{};
    p.qkv = qkv;
    p.out = out;
    p.cos = h->blob + h->mimi.rope_cos;
    p.sin = h->blob + h->mimi.rope_sin;
    p.B = B;
    p.T = T;
    p.A = c.num_attention_heads * c.head_dim;
    p.window = c.sliding_window;
    p.scaling = 1.0f / std::sqrt((float)c.head_dim);
    const dim3 grid(cdiv(T, 64), c.num_attention_heads, B);
    const double keys = std::min<double>(T, c.sliding_window);
    const bool a16 = c.head_dim == 64 && !h->gemm_fp32 && !h->dev.attn_exact && (p.A % 4) == 0;
    ProfScope ps(h, st, a16 ? "attention16_kernel" : "attention_kernel", 4.0 * B * c.num_attention_heads * (double)T * keys * c.head_dim * 0.5,
                 16.0 * B * (double)T * p.A);
    if (a16) {      // split16 products on the fp16 pipe (mimi.h)
        hipLaunchKernelGGL(attention16_kernel, grid, dim3(256), Attn16Cfg::lds_bytes, st, p);
    } else if (c.head_dim == 64) {
        if (int rc = ensure_lds(h, reinterpret_cast<const void*>(attention_kernel<64>), AttnCfg<64>::lds_bytes)) return rc;
        hipLaunchKernelGGL(attention_kernel<64>, grid, dim3(256), AttnCfg<64>::lds_bytes, st, p);
    } else if (c.head_dim == 32) {
        hipLaunchKernelGGL(attention_kernel<32>, grid, dim3(256), AttnCfg<32>::lds_bytes, st, p);
    } else if (c.head_dim == 16) {
        hipLaunchKernelGGL(attention_kernel<16>, grid, dim3(256), AttnCfg<16>::lds_bytes, st, p);
    } else {
        return fail(h, AC_EINVAL, "head_dim %d unsupported (16, 32 or 64)", c.head_dim);
    }
    HIPCHK(h, hipGetLastError());
    return AC_OK;
}

struct TfScratch {
    float *ln, *qkv, *att, *hid;
};

// x [B*T][H] updated in place: x += s_a * o_proj(attn(LN1 x));  x += s_m * fc2(gelu(fc1(LN2 x)))
int transformer_fwd(ac_handle* h, hipStream_t st, const std::vector<MimiTfLayer>& layers, float* x, int B, int T, const TfScratch& s) {
    const ac_mimi_config& c = h->mcfg;
    const int H = c.hidden_size, A = c.num_attention_heads * c.head_dim, I = c.intermediate_size;
    const long long rows = (long long)B * T;
    for (const MimiTfLayer& L : layers) {
        const unsigned* ln_rows = nullptr;       // split16.h row mode: the LayerNorm leaves the row amax of its output for the projection
        int rc = layernorm_fwd(h, st, x, L.ln1_w, L.ln1_b, s.ln, rows, H, c.norm_eps, &ln_rows);
        if (rc) return rc;
        Epi eq;
        eq.rowmax_in = ln_rows;
        if ((rc = mimi_linear(h, st, L.qkv, s.ln, rows, H, H, 0, s.qkv, 3 * A, eq))) return rc;
        if ((rc = attention_fwd(h, st, s.qkv, s.att, B, T))) return rc;
        Epi ea;
        ea.scale = h->blob + L.sc_a;
        ea.res = x;
        ea.res_rs = H;
        if ((rc = mimi_linear(h, st, L.o, s.att, rows, A, A, 0, x, H, ea))) return rc;
        if ((rc = layernorm_fwd(h, st, x, L.ln2_w, L.ln2_b, s.ln, rows, H, c.norm_eps, &ln_rows))) return rc;
        Epi eg;
        eg.gelu = 1;
        eg.rowmax_in = ln_rows;
        const unsigned* hid_rows = nullptr;      // split16.h row mode: fc1's epilogue leaves the row amax of its output for fc2
        eg.rowmax_out = &hid_rows;
        if ((rc = mimi_linear(h, st, L.fc1, s.ln, rows, H, H, 0, s.hid, I, eg))) return rc;
        Epi em;
        em.rowmax_in = hid_rows;
        em.scale = h->blob + L.sc_m;
        em.res = x;
        em.res_rs = H;
        if ((rc = mimi_linear(h, st, L.fc2, s.hid, rows, I, I, 0, x, H, em))) return rc;
        capture(h, st, Act{x, (long long)T * H, H, T, H}, B);
    }
    return AC_OK;
}

int mimi_rvq_encode(ac_handle* h, hipStream_t st, const float* proj /*[F][2*Dq]*/, int F, int K, long long* toks) {
    const ac_mimi_config& c = h->mcfg;
    const int Dq = c.codebook_dim, C = c.codebook_size, nsem = c.num_semantic_quantizers;
    for (int part = 0; part < 2; ++part) {
        const int k0 = part == 0 ? 0 : nsem;
        const int n = part == 0 ? std::min(K, nsem) : K - nsem;
        if (n <= 0) continue;
        RvqEncParams p{};
        p.x = proj + (size_t)part * Dq;
        p.xs = 2 * Dq;
        p.epk = h->blob + h->mimi.cb_packed + (size_t)k0 * C * Dq;
        p.e = h->blob + h->mimi.cb_plain + (size_t)k0 * C * Dq;
        p.ee = h->blob + h->mimi.cb_ee + (size_t)k0 * C;
        p.toks = toks;
        p.F = F;
        p.H = Dq;
        p.C = C;
        p.K = n;
        p.tK = K;
        p.tk0 = k0;
        const bool s16 = h->mimi.cb16 && Dq == 256 && !h->dev.rvq_exact;
        ProfScope ps(h, st, s16 ? "rvq_encode16_kernel" : "rvq_encode_kernel", 2.0 * F * (double)C * Dq * n, (double)F * Dq * 4 + (double)F * n * 8 + (double)n * C * Dq * 4);
        if (int rc = rvq_encode_cdist_launch(h, st, p, (unsigned)cdiv(F, 16), s16 ? reinterpret_cast<const _Float16*>(h->blob + h->mimi.cb16) + (size_t)k0 * C * Dq * 2 : nullptr,
                                             s16 ? h->blob + h->mimi.cb16_inv + k0 : nullptr)) return rc;
        HIPCHK(h, hipGetLastError());
    }
    return AC_OK;
}

// toks [F][K] -> qfeats [F][H] = out_proj_s(sum semantic codes) + out_proj_a(sum acoustic codes); qsum is [F][2*Dq] scratch
int mimi_rvq_decode(ac_handle* h, hipStream_t st, const long long* toks, int F, int K, float* qsum, float* qfeats) {
    const ac_mimi_config& c = h->mcfg;
    const int Dq = c.codebook_dim, C = c.codebook_size, nsem = c.num_semantic_quantizers, H = c.hidden_size;
    const int nparts = K > nsem ? 2 : 1;
    for (int part = 0; part < nparts; ++part) {
        const int k0 = part == 0 ? 0 : nsem;
        const int n = part == 0 ? std::min(K, nsem) : K - nsem;
        RvqDecParams p{};
        p.toks = toks;
        p.e = h->blob + h->mimi.cb_plain + (size_t)k0 * C * Dq;
        p.out = qsum + (size_t)part * Dq;
        p.F = F;
        p.H = Dq;
        p.C = C;
        p.K = n;
        p.tK = K;
        p.tk0 = k0;
        p.os = 2 * Dq;
        const long long cnt = (long long)F * (Dq / 4);
        p.bad = h->sticky_dev ? h->sticky_dev + ST_BAD_TOKEN : nullptr;
        ProfScope ps(h, st, "rvq_decode_kernel", (double)F * Dq * n, (double)F * n * 8 + (double)F * Dq * 4 * (n + 1));
        rvq_decode_launch(st, p, (unsigned)((cnt + 255) / 256));
        HIPCHK(h, hipGetLastError());
    }
    // one GEMM over the K-concatenation [q_s | q_a] (only the semantic columns when K <= num_semantic_quantizers)
    TapGemmParams p{};
    p.nseg = 1;
    Act qa{qsum, 0, 2 * Dq, F, nparts * Dq};
    p.seg[0] = make_seg(qa, 1, 1, PAD_ZERO, 0, 0, nullptr);
    p.w = h->blob + h->mimi.out_proj.w_off;
    p.bias = nullptr;
    p.y = qfeats;
    p.y_bs = 0;
    p.y_rs = H;
    p.B = 1;
    p.M = F;
    p.N = H;
    p.Ktot = 2 * Dq;
    p.amax_rows = 1;                               // split16.h row mode: a merged row matrix, every row scales by its own amax
    return run_tap(h, st, p);
}

// ---------------------------------------------------------------------------------------------
// workspace
// ---------------------------------------------------------------------------------------------
int mimi_num_frames25(const ac_mimi_config& c, long long T) {   // frames at the SEANet rate (before the stride-2 conv)
    long long L = T;
    for (int r = c.num_ratios - 1; r >= 0; --r) L = (L + c.upsampling_ratios[r] - 1) / c.upsampling_ratios[r];
    return (int)L;
}

Workspace mimi_plan_ws(const ac_handle* h, int B, int T_in, int N_frames, bool enc) {
    const ac_mimi_config& c = h->mcfg;
    Workspace w;
    size_t mx = 0;
    int T25;
    if (enc) {
        long long L = T_in;
        int ch = c.num_filters;
        mx = std::max(mx, (size_t)L * ch);
        for (int r = c.num_ratios - 1; r >= 0; --r) {
            L = (L + c.upsampling_ratios[r] - 1) / c.upsampling_ratios[r];
            ch *= 2;
            mx = std::max(mx, (size_t)L * ch);
        }
        T25 = (int)L;
    } else {
        T25 = N_frames * c.resample_stride;
        long long L = T25;
        int ch = h->mimi.D;
        mx = std::max(mx, (size_t)L * ch);
        for (int r = 0; r < c.num_ratios; ++r) {
            L *= c.upsampling_ratios[r];
            ch /= 2;
            mx = std::max(mx, (size_t)L * ch);
        }
    }
    const int A = c.num_attention_heads * c.head_dim;
    const int widest = std::max(std::max(3 * A, c.intermediate_size), std::max(c.hidden_size, std::max(h->mimi.D, 2 * c.codebook_dim)));
    mx = std::max(mx, (size_t)T25 * widest);
    w.act_floats = align_up(mx * B, 64);
    w.total_bytes = NACT * w.act_floats * sizeof(float) + 256;
    add_pool(w, B, (size_t)B * T25);      // row mode of the transformers' linear layers: one word per token row
    return w;
}

// ---------------------------------------------------------------------------------------------
// forward passes
// ---------------------------------------------------------------------------------------------
// sig [B][T] -> feats [B][N][H] (the embeddings the quantiser sees = what Mimi._sig_to_feats returns, mimi.py:112-121)
int mimi_encoder_fwd(ac_handle* h, hipStream_t st, const float* sig, int B, int T, float* feats, WsPtrs& ws) {
    const ac_mimi_config& c = h->mcfg;
    const MimiPlan& m = h->mimi;
    const bool dbg = h->dbg != nullptr;
    const int F = c.num_filters;
    Act xin{sig, (long long)T, 1, T, 1};
    Act2 x, y;
    int rc;
    // round 6: the stem folded into the first residual block (rb_stream6m.h STEM): the 64-channel tensor at the sample rate is never written.
    // (The capture hook wants the stem's output as a tensor: the separate kernels then.)
    int first_block = 0;
    const bool fold_stem = !dbg && h->dev.rb_stream && !h->gemm_fp32 && m.sm.stem_ok && F == 64 && c.kernel_size == 7 && c.num_ratios >= 1 &&
                           c.residual_kernel_size == 3 && c.compress == 2 && (long long)T * 256 < 0x70000000LL;
    if (fold_stem) {
        const bool v4 = T % 4 == 0 && aligned16(sig);
        const unsigned* am_sig = amax_of(h, st, sig, T, v4 ? 4 : 1, v4 ? T / 4 : T, v4 ? 4 : 1, B, nullptr);
        if (!am_sig) return fail(h, AC_ESTATE, "out of amax slots (split16.h)");
        unsigned* am_out = amax_new(h);
        Out o{nullptr, ws.take()};                         // the down-sampler reads ELU(y) only
        rc = launch_rb_stream6m(h, st, m.enc_rb[0], nullptr, sig, B, T, o, nullptr, 0, am_sig, am_out);
        if (rc) return rc;
        HIPCHK(h, hipGetLastError());
        x.raw = Act{nullptr, (long long)T * 64, 64, T, 64, am_out, B};
        x.elu = Act{o.elu, (long long)T * 64, 64, T, 64, am_out, B};
        first_block = 1;
    } else
    if (F % 4 == 0 && F <= 64 && c.kernel_size <= THIN_MAXK)
        rc = thin_stem(h, st, m.enc_stem, F, c.kernel_size, PAD_ZERO, sig, nullptr, B, T,
                       Out{ws.take(), mimi_rb_self_elu(h, c, F) ? nullptr : ws.take()}, &x);
    else
        rc = mimi_conv(h, st, m.enc_stem, xin, c.kernel_size, 1, PAD_ZERO, Out{ws.take(), ws.take()}, B, &x);
    if (rc) return rc;
    if (!first_block) capture(h, st, x.raw, B);
    for (int i = 0; i < c.num_ratios; ++i) {
        const int ratio = c.upsampling_ratios[c.num_ratios - 1 - i];
        if (i >= first_block) {
            float* hb = ws.take();
            rc = mimi_resblock(h, st, m.enc_rb[i], x, hb, Out{dbg ? ws.take() : nullptr, ws.take()}, B, &y);
            if (rc) return rc;
            ws.give(hb);
            ws.give(x);
            if (dbg) { capture(h, st, y.raw, B); ws.give(y.raw.p); y.raw.p = nullptr; }
            x = y;
        }
        const bool last = i == c.num_ratios - 1;     // the last down-sampler feeds ELU -> final conv only
        const int cout = m.enc_down[i].N;
        Out o = last ? Out{dbg ? ws.take() : nullptr, ws.take()} : Out{ws.take(), mimi_rb_self_elu(h, c, cout) ? nullptr : ws.take()};
        rc = mimi_conv(h, st, m.enc_down[i], x.elu, 2 * ratio, ratio, PAD_ZERO, o, B, &y);
        if (rc) return rc;
        ws.give(x);
        x = y;
        if (!last || dbg) capture(h, st, x.raw, B);
        if (last && dbg) { ws.give(x.raw.p); x.raw.p = nullptr; }
    }
    float* stream = ws.take();
    rc = mimi_conv(h, st, m.enc_final, x.elu, c.last_kernel_size, 1, PAD_ZERO, Out{stream, nullptr}, B, &y);
    if (rc) return rc;
    ws.give(x);
    capture(h, st, y.raw, B);
    const int T25 = y.raw.L;
    TfScratch s{ws.take(), ws.take(), ws.take(), ws.take()};
    rc = transformer_fwd(h, st, m.enc_tf, stream, B, T25, s);
    if (rc) return rc;
    ws.give(s.ln); ws.give(s.qkv); ws.give(s.att); ws.give(s.hid);
    y.raw.amax = nullptr;   // the transformer rewrote `stream` in place: the conv's amax (split16.h) no longer describes it
    y.raw.amax_n = 0;
    rc = mimi_conv(h, st, m.down, y.raw, 2 * c.resample_stride, c.resample_stride, PAD_REPLICATE, Out{feats, nullptr}, B, &x);
    ws.give(stream);
    if (rc) return rc;
    capture(h, st, x.raw, B);
    return AC_OK;
}

// qfeats [B][N][H] (quantiser output) -> sig [B][N*hop]
int mimi_decoder_fwd(ac_handle* h, hipStream_t st, const float* qfeats, int B, int N, float* sig, WsPtrs& ws) {
    const ac_mimi_config& c = h->mcfg;
    const MimiPlan& m = h->mimi;
    const bool dbg = h->dbg != nullptr;
    const int H = c.hidden_size, T25 = N * c.resample_stride;
    float* stream = ws.take();
    {
        UpsampleParams p{qfeats, h->blob + m.up_w, stream, B, N, H, c.resample_stride};
        const long long total = (long long)B * T25 * (H / 4);
        ProfScope ps(h, st, "upsample_dw_kernel", 4.0 * B * (double)T25 * H, 4.0 * B * (double)H * (N + T25));
        hipLaunchKernelGGL(upsample_dw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, p);
        HIPCHK(h, hipGetLastError());
    }
    ws.give(qfeats);   // (when it is one of the rotating buffers) consumed
    Act xs{stream, (long long)T25 * H, H, T25, H};
    capture(h, st, xs, B);
    TfScratch s{ws.take(), ws.take(), ws.take(), ws.take()};
    int rc = transformer_fwd(h, st, m.dec_tf, stream, B, T25, s);
    if (rc) return rc;
    ws.give(s.ln); ws.give(s.qkv); ws.give(s.att); ws.give(s.hid);
    Act2 x, y;
    rc = mimi_conv(h, st, m.dec_first, xs, c.kernel_size, 1, PAD_ZERO, Out{dbg ? ws.take() : nullptr, ws.take()}, B, &x);
    if (rc) return rc;
    ws.give(stream);
    if (dbg) { capture(h, st, x.raw, B); ws.give(x.raw.p); x.raw.p = nullptr; }
    for (int i = 0; i < c.num_ratios; ++i) {
        const int cup = m.dec_up[i].N / c.upsampling_ratios[i];
        rc = convtr_fwd(h, st, m.dec_up[i], x.elu, c.upsampling_ratios[i], Out{ws.take(), mimi_rb_self_elu(h, c, cup) ? nullptr : ws.take()}, B, &y);
        if (rc) return rc;
        ws.give(x);
        x = y;
        capture(h, st, x.raw, B);
        // last block: the final conv runs on the block's output tile in LDS (rb_fused6.h HEAD) -- the widest tensor of the path stays on chip
        if (i == c.num_ratios - 1 && m.dec_rb[i].C == 64 && c.residual_kernel_size == 3 && c.compress == 2 && c.num_filters == 64 && x.raw.ts == 64 &&
            x.raw.bs == (long long)x.raw.L * 64 && aligned16(x.raw.p)) {
            int rc2 = AC_OK;
            if (!dbg && h->dev.rb_stream && h->dev.mimi_tail && !h->gemm_fp32 && m.sm.head_ok && (long long)x.raw.L * 256 < 0x70000000LL) {   // rb_stream6m.h HEAD
                const unsigned* am = amax_of(h, st, x.raw.p, x.raw.bs, x.raw.ts, x.raw.L, x.raw.C, B, x.raw.amax_n == B ? x.raw.amax : nullptr);
                if (!am) return fail(h, AC_ESTATE, "out of amax slots (split16.h)");
                rc2 = launch_rb_stream6m(h, st, m.dec_rb[i], x.raw.p, nullptr, B, x.raw.L, Out{}, sig, c.last_kernel_size, am, nullptr);
                if (!rc2) HIPCHK(h, hipGetLastError());
                ws.give(x);
                return rc2;
            }
            if (rb64_identity_head_fwd(h, st, m.dec_rb[i], x, m.dec_head, c.last_kernel_size, sig, B, &rc2)) {
                ws.give(x);
                return rc2;
            }
        }
        float* hb = ws.take();
        rc = mimi_resblock(h, st, m.dec_rb[i], x, hb, Out{dbg ? ws.take() : nullptr, ws.take()}, B, &y);
        if (rc) return rc;
        ws.give(hb);
        ws.give(x);
        if (dbg) { capture(h, st, y.raw, B); ws.give(y.raw.p); y.raw.p = nullptr; }
        x = y;
    }
    const int F = c.num_filters;
    if (F % 4 == 0 && F <= 64 && c.last_kernel_size <= THIN_MAXK && x.elu.ts == F)
        rc = thin_head(h, st, m.dec_head, F, c.last_kernel_size, PAD_ZERO, x.elu, B, sig);
    else
        rc = mimi_conv(h, st, m.dec_head, x.elu, c.last_kernel_size, 1, PAD_ZERO, Out{sig, nullptr}, B, nullptr);
    ws.give(x);
    return rc;
}

// ---------------------------------------------------------------------------------------------
// weight packing (HF MimiModel state-dict keys; checkpoint.mimi_conv_specs is the Python twin)
// ---------------------------------------------------------------------------------------------
int mimi_finalize(ac_handle* h, Packer& pk) {
    const ac_mimi_config& c = h->mcfg;
    MimiPlan& m = h->mimi;
    const int F = c.num_filters, H = c.hidden_size, n = c.num_ratios;
    auto P = [](const char* part, int i, const char* rest) { return std::string(part) + ".layers." + std::to_string(i) + rest; };
    bool ok = pk.conv(ConvSpec{P("encoder", 0, ".conv"), 0, 1, F, c.kernel_size, 1}, m.enc_stem);
    m.enc_rb.resize(n);
    m.enc_down.resize(n);
    m.dec_up.resize(n);
    m.dec_rb.resize(n);
    auto resblock = [&](const std::string& pre, int ch, ResBlockPlan& rb) {
        rb.C = ch;
        const int hid = ch / c.compress;
        if (!pk.conv(ConvSpec{pre + ".block.1.conv", 0, ch, hid, c.residual_kernel_size, 1}, rb.c3) ||
            !pk.conv(ConvSpec{pre + ".block.3.conv", 0, hid, ch, 1, 1}, rb.fused))
            return false;
        if (c.residual_kernel_size == 3 && c.compress == 2) pk.rb6(rb, false);
        return true;
    };
    int i = 1, ch = F;
    for (int r = n - 1, j = 0; ok && r >= 0; --r, ++j) {
        const int ratio = c.upsampling_ratios[r];
        ok = ok && resblock(P("encoder", i, ""), ch, m.enc_rb[j]);
        ok = ok && pk.conv(ConvSpec{P("encoder", i + 2, ".conv"), 0, ch, 2 * ch, 2 * ratio, ratio}, m.enc_down[j]);
        i += 3;
        ch *= 2;
    }
    m.D = ch;
    ok = ok && pk.conv(ConvSpec{P("encoder", i + 1, ".conv"), 0, ch, H, c.last_kernel_size, 1}, m.enc_final);
    ok = ok && pk.conv(ConvSpec{"downsample.conv", 0, H, H, 2 * c.resample_stride, c.resample_stride}, m.down, false);
    ok = ok && pk.conv(ConvSpec{P("decoder", 0, ".conv"), 0, H, ch, c.kernel_size, 1}, m.dec_first);
    i = 1;
    for (int r = 0; ok && r < n; ++r) {
        const int ratio = c.upsampling_ratios[r];
        ok = ok && pk.convtr(ConvSpec{P("decoder", i + 1, ".conv"), 1, ch, ch / 2, 2 * ratio, ratio}, m.dec_up[r]);
        ok = ok && resblock(P("decoder", i + 2, ""), ch / 2, m.dec_rb[r]);
        i += 3;
        ch /= 2;
    }
    ok = ok && pk.conv(ConvSpec{P("decoder", i + 1, ".conv"), 0, ch, 1, c.last_kernel_size, 1}, m.dec_head);
    if (!ok) return pk.rc;
    if (n >= 1 && c.residual_kernel_size == 3 && c.compress == 2) pk.stream_mimi(m.enc_stem, m.enc_rb[0], m.dec_rb[n - 1], m.dec_head, c.last_kernel_size, m.sm);
    {   // depthwise up-sampler [H][1][2s]
        const std::vector<float>* w = pk.get("upsample.conv.weight", (size_t)H * 2 * c.resample_stride);
        if (!w) return pk.rc;
        m.up_w = pk.reserve(w->size());
        std::copy(w->begin(), w->end(), pk.blob.begin() + m.up_w);
    }
    // transformers
    const int A = c.num_attention_heads * c.head_dim, I = c.intermediate_size;
    auto vec = [&](const std::string& name, size_t nel, size_t& off) {
        const std::vector<float>* v = pk.get(name, nel);
        if (!v) return false;
        off = pk.reserve(nel);
        std::copy(v->begin(), v->end(), pk.blob.begin() + off);
        return true;
    };
    auto linear = [&](const std::vector<std::string>& names, int out_each, int in, PackedGemm& g) {
        g.N = out_each * (int)names.size();
        g.Ktot = in;
        g.has_bias = false;
        g.w_off = pk.reserve((size_t)g.N * in);
        for (size_t j = 0; j < names.size(); ++j) {
            const std::vector<float>* w = pk.get(names[j], (size_t)out_each * in);
            if (!w) return false;
            std::copy(w->begin(), w->end(), pk.blob.begin() + g.w_off + j * (size_t)out_each * in);
        }
        pk.pack6(g);
        return true;
    };
    for (int part = 0; part < 2; ++part) {
        std::vector<MimiTfLayer>& tf = part == 0 ? m.enc_tf : m.dec_tf;
        tf.resize(c.num_hidden_layers);
        for (int l = 0; l < c.num_hidden_layers; ++l) {
            const std::string p = std::string(part == 0 ? "encoder" : "decoder") + "_transformer.layers." + std::to_string(l);
            MimiTfLayer& L = tf[l];
            ok = linear({p + ".self_attn.q_proj.weight", p + ".self_attn.k_proj.weight", p + ".self_attn.v_proj.weight"}, A, H, L.qkv) &&
                 linear({p + ".self_attn.o_proj.weight"}, H, A, L.o) && linear({p + ".mlp.fc1.weight"}, I, H, L.fc1) &&
                 linear({p + ".mlp.fc2.weight"}, H, I, L.fc2) && vec(p + ".input_layernorm.weight", H, L.ln1_w) &&
                 vec(p + ".input_layernorm.bias", H, L.ln1_b) && vec(p + ".post_attention_layernorm.weight", H, L.ln2_w) &&
                 vec(p + ".post_attention_layernorm.bias", H, L.ln2_b) && vec(p + ".self_attn_layer_scale.scale", H, L.sc_a) &&
                 vec(p + ".mlp_layer_scale.scale", H, L.sc_m);
            if (!ok) return pk.rc;
        }
    }
    // RoPE tables: inv_freq (fp32) * position (fp32 product, as the reference's [d/2,1]@[1,T] matmul gives), cos/sin
    // rounded once from double.  The host may hand over the reference's own inv_freq buffer (non-persistent in HF).
    {
        const int hd = c.head_dim;
        std::vector<float> inv(hd / 2);
        auto it = h->host.find("encoder_transformer.rotary_emb.inv_freq");
        if (it != h->host.end() && it->second.size() == (size_t)hd / 2) inv = it->second;
        else
            for (int j = 0; j < hd / 2; ++j) inv[j] = 1.0f / std::pow(c.rope_theta, (float)(2 * j) / (float)hd);
        m.rope_T = MIMI_ROPE_T;
        m.rope_cos = pk.reserve((size_t)m.rope_T * hd);
        m.rope_sin = pk.reserve((size_t)m.rope_T * hd);
        for (int t = 0; t < m.rope_T; ++t)
            for (int j = 0; j < hd / 2; ++j) {
                const float ang = inv[j] * (float)t;
                const float cv = (float)std::cos((double)ang), sv = (float)std::sin((double)ang);
                pk.blob[m.rope_cos + (size_t)t * hd + j] = pk.blob[m.rope_cos + (size_t)t * hd + j + hd / 2] = cv;
                pk.blob[m.rope_sin + (size_t)t * hd + j] = pk.blob[m.rope_sin + (size_t)t * hd + j + hd / 2] = sv;
            }
    }
    // quantiser: projections and codebooks (embed = embed_sum / clamp(cluster_usage, 1e-5), [HF]:980-983)
    const int Dq = c.codebook_dim, C = c.codebook_size, Q = c.num_quantizers, nsem = c.num_semantic_quantizers;
    const std::string qs = "quantizer.semantic_residual_vector_quantizer", qa = "quantizer.acoustic_residual_vector_quantizer";
    if (!linear({qs + ".input_proj.weight", qa + ".input_proj.weight"}, Dq, H, m.in_proj)) return pk.rc;
    {
        const std::vector<float>* ws = pk.get(qs + ".output_proj.weight", (size_t)H * Dq);
        const std::vector<float>* wa = pk.get(qa + ".output_proj.weight", (size_t)H * Dq);
        if (!ws || !wa) return pk.rc;
        m.out_proj.N = H;
        m.out_proj.Ktot = 2 * Dq;
        m.out_proj.has_bias = false;
        m.out_proj.w_off = pk.reserve((size_t)H * 2 * Dq);
        for (int r = 0; r < H; ++r)
            for (int d = 0; d < Dq; ++d) {
                pk.blob[m.out_proj.w_off + (size_t)r * 2 * Dq + d] = (*ws)[(size_t)r * Dq + d];
                pk.blob[m.out_proj.w_off + (size_t)r * 2 * Dq + Dq + d] = (*wa)[(size_t)r * Dq + d];
            }
    }
    m.cb_plain = pk.reserve((size_t)Q * C * Dq);
    m.cb_packed = pk.reserve((size_t)Q * C * Dq);
    m.cb_ee = pk.reserve((size_t)Q * C);
    for (int q = 0; q < Q; ++q) {
        const std::string cb = (q < nsem ? qs + ".layers." + std::to_string(q) : qa + ".layers." + std::to_string(q - nsem)) + ".codebook";
        std::vector<float> e((size_t)C * Dq);
        auto it = h->host.find(cb + ".embed");
        if (it != h->host.end() && it->second.size() == e.size()) {
            e = it->second;
        } else {
            const std::vector<float>* es = pk.get(cb + ".embed_sum", (size_t)C * Dq);
            const std::vector<float>* cu = pk.get(cb + ".cluster_usage", (size_t)C);
            if (!es || !cu) return pk.rc;
            for (int code = 0; code < C; ++code) {
                const float u = std::max((*cu)[code], 1e-5f);
                for (int d = 0; d < Dq; ++d) e[(size_t)code * Dq + d] = (*es)[(size_t)code * Dq + d] / u;
            }
        }
        std::copy(e.begin(), e.end(), pk.blob.begin() + m.cb_plain + (size_t)q * C * Dq);
        for (int code = 0; code < C; ++code) {
            double ss = 0.0;
            for (int d = 0; d < Dq; ++d) ss += (double)e[(size_t)code * Dq + d] * e[(size_t)code * Dq + d];
            pk.blob[m.cb_ee + (size_t)q * C + code] = (float)ss;
        }
        const int HV = Dq / 16;
        for (int ct = 0; ct < C / 16; ++ct)
            for (int v = 0; v < HV; ++v)
                for (int lane = 0; lane < 64; ++lane)
                    for (int u = 0; u < 4; ++u)
                        pk.blob[m.cb_packed + (size_t)q * C * Dq + (((size_t)ct * HV + v) * 64 + lane) * 4 + u] =
                            e[(size_t)(ct * 16 + (lane & 15)) * Dq + v * 16 + 4 * (lane >> 4) + u];
    }
    if (pk.use16() && Dq == 256) pk.pack_cb16(m.cb_plain, Q, C, Dq, &m.cb16, &m.cb16_inv);     // rvq16.h
    return AC_OK;
}

}  // namespace acimpl

extern "C" int ac_mimi_create(const ac_mimi_config* cfg, ac_handle** out) {
    if (!cfg || !out) return AC_EINVAL;
    *out = nullptr;
    if (cfg->struct_size != (int32_t)sizeof(ac_mimi_config)) return AC_EINVAL;
    const ac_mimi_config& c = *cfg;
    if (c.num_ratios < 1 || c.num_ratios > AC_MAX_RATIOS || c.num_filters < 1 || c.hidden_size < 16 || c.hidden_size % 16 ||
        c.hidden_size > 64 * LN_MAXV || c.compress < 1 || c.codebook_size % 32 || c.codebook_size < 32 || c.codebook_dim < 16 ||
        c.codebook_dim % 16 || c.codebook_dim > 256 || c.num_quantizers < 1 || c.num_semantic_quantizers < 1 ||
        c.num_semantic_quantizers > c.num_quantizers || c.kernel_size < 1 || c.kernel_size > 8 || c.last_kernel_size < 1 ||
        c.last_kernel_size > 8 || c.residual_kernel_size < 1 || c.residual_kernel_size > 8 || c.num_hidden_layers < 0 ||
        c.num_attention_heads < 1 || (c.head_dim != 16 && c.head_dim != 32 && c.head_dim != 64) || c.intermediate_size < 16 ||
        c.intermediate_size % 4 || c.sliding_window < 1 || c.resample_stride < 1 || c.resample_stride > 4 || !(c.norm_eps > 0.f) ||
        !(c.rope_theta > 0.f))
        return AC_EINVAL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= c.device || c.device < 0) return AC_ENODEV;
    ac_handle* h = new (std::nothrow) ac_handle();
    if (!h) return AC_ENOMEM;
    h->arch = ARCH_MIMI;
    h->mcfg = c;
    h->hop = c.resample_stride;
    for (int i = 0; i < c.num_ratios; ++i) h->hop *= c.upsampling_ratios[i];
    h->D = c.num_filters << c.num_ratios;
    h->mimi.D = h->D;
    *out = h;
    return AC_OK;
}
