// C ABI of the MI355X codec paths (include/audiocodecs_amd.h): the extern "C" entry points.  The machinery behind them is in
// core.hip (shared + EnCodec) and mimi_path.hip / dac_path.hip / wavtok_path.hip (one codec each, with its ac_*_create); core.h
// has the map.  gfx950 only.
#include "core.h"

// ---------------------------------------------------------------------------------------------
// exported entry points
// ---------------------------------------------------------------------------------------------
extern "C" {

int ac_version(void) { return 400; }   // 400: split16 and exact-fp32 arithmetic only (the bf16x3 / bf16 modes are gone)

int ac_create(const ac_config* cfg, ac_handle** out) {
    if (!cfg || !out) return AC_EINVAL;
    *out = nullptr;
    if (cfg->struct_size != (int32_t)sizeof(ac_config)) return AC_EINVAL;
    if (cfg->num_ratios < 1 || cfg->num_ratios > AC_MAX_RATIOS || cfg->num_filters < 1 || cfg->hidden_size < 16 ||
        cfg->hidden_size % 16 || cfg->compress < 1 || cfg->num_lstm_layers < 1 || cfg->num_lstm_layers > 2 || cfg->codebook_size % 32 ||
        cfg->codebook_size < 16 || cfg->num_quantizers < 1 || cfg->kernel_size < 1 || cfg->kernel_size > 8 ||
        cfg->last_kernel_size < 1 || cfg->last_kernel_size > 8 || cfg->residual_kernel_size < 1 || cfg->residual_kernel_size > 8)
        return AC_EINVAL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= cfg->device || cfg->device < 0) return AC_ENODEV;
    ac_handle* h = new (std::nothrow) ac_handle();
    if (!h) return AC_ENOMEM;
    h->cfg = *cfg;
    h->hop = 1;
    for (int i = 0; i < cfg->num_ratios; ++i) h->hop *= cfg->upsampling_ratios[i];
    h->D = cfg->num_filters << cfg->num_ratios;
    *out = h;
    return AC_OK;
}







// WavTokenizer checkpoints spell the encoder's modules the encodec-library way; the packer knows the HF spelling
static std::string wavtok_key(const std::string& name) {
    static const std::string pre = "feature_extractor.encodec.encoder.model.";
    if (name.compare(0, pre.size(), pre) != 0) return name;
    std::string k = "encoder.layers." + name.substr(pre.size());
    const size_t cc = k.find(".conv.conv.");
    if (cc != std::string::npos) k = k.substr(0, cc) + ".conv." + k.substr(cc + 11);
    auto ends = [&](const char* sfx) { const size_t n = std::strlen(sfx); return k.size() >= n && k.compare(k.size() - n, n, sfx) == 0; };
    if (ends(".weight_g")) k = k.substr(0, k.size() - 9) + ".parametrizations.weight.original0";
    else if (ends(".weight_v")) k = k.substr(0, k.size() - 9) + ".parametrizations.weight.original1";
    return k;
}

int ac_load_weights(ac_handle* h, const char* name, const void* host_ptr, size_t bytes) {
    if (!h || !name || !host_ptr) return h ? fail(h, AC_EINVAL, "null argument") : AC_EINVAL;
    if (h->finalized) return fail(h, AC_ESTATE, "handle already finalized");
    if (bytes % 4) return fail(h, AC_EINVAL, "tensor '%s': byte size %zu is not a multiple of 4", name, bytes);
    const float* f = static_cast<const float*>(host_ptr);
    h->host[h->arch == ARCH_WAVTOK ? wavtok_key(name) : std::string(name)].assign(f, f + bytes / 4);
    return AC_OK;
}



// the developer switches of ac_handle::dev, read from the environment exactly once per handle (here, from ac_finalize)
static void latch_dev_switches(ac_handle* h) {
    auto num = [](const char* name, int dflt) { const char* v = std::getenv(name); return v && *v ? std::atoi(v) : dflt; };
    const char* e = std::getenv("AC_TAP_EPI");
    h->dev.tap_epi_staged = e && std::strcmp(e, "staged") == 0;
    h->dev.tap_dil = num("AC_TAP_DIL", 1);
    h->dev.tap_stagger = num("AC_TAP_STAGGER", 0);
    h->dev.tap_pick = num("AC_TAP_PICK", -1);
    h->dev.tap8 = num("AC_TAP8", -1);
    h->dev.tap8_form = num("AC_TAP8_FORM", 0);
    h->dev.tap8_spread = num("AC_TAP8_SPREAD", 1);
#ifdef AC_DEVELOPER     // (the product library neither reads these nor acts on the words: split16.h AC_DEV_MODE)
    h->dev.rb6_dbg = num("AC_RB6_DBG", 0);
    h->dev.lstm_dbg = num("AC_LSTM_DBG", 0);
#endif
    h->dev.rb_stream = num("AC_RB_STREAM", 1);
    h->dev.chain_stream = num("AC_CHAIN_STREAM", 1);
    h->dev.rb128_stream = num("AC_RB128_STREAM", 1);
    h->dev.front_seg = std::max(0, num("AC_FRONT_SEG", 0));
    h->dev.tail_seg = std::max(0, num("AC_TAIL_SEG", 0));
    h->dev.front_ldspad = std::max(0, num("AC_FRONT_LDSPAD", 0));
    h->dev.lstm_fuse_in = num("AC_LSTM_FUSE_IN", 1);
    const char* r = std::getenv("AC_RVQ");
    h->dev.rvq_exact = r && std::strcmp(r, "fp32") == 0;
    h->dev.prof_detail = num("AC_PROF_DETAIL", 0);
    h->dev.head_seq = num("AC_HEAD_SEQ", 0);
    h->dev.attn_exact = num("AC_ATTN_EXACT", 0);
    h->dev.dac_unit = num("AC_DAC_UNIT", 1);
    h->dev.mimi_tail = num("AC_MIMI_TAIL", 1);
}

int ac_debug_set(ac_handle* h, const char* key, int value) {
    if (!h || !key) return AC_EINVAL;
    struct { const char* k; int* v; } tab[] = {
        {"tap_epi_staged", &h->dev.tap_epi_staged}, {"tap_dil", &h->dev.tap_dil}, {"tap_stagger", &h->dev.tap_stagger}, {"tap_pick", &h->dev.tap_pick}, {"tap8", &h->dev.tap8}, {"tap8_form", &h->dev.tap8_form}, {"tap8_spread", &h->dev.tap8_spread},
        {"rb6_dbg", &h->dev.rb6_dbg}, {"rb_stream", &h->dev.rb_stream}, {"chain_stream", &h->dev.chain_stream}, {"rb128_stream", &h->dev.rb128_stream}, {"front_seg", &h->dev.front_seg}, {"tail_seg", &h->dev.tail_seg}, {"front_ldspad", &h->dev.front_ldspad},
        {"lstm_dbg", &h->dev.lstm_dbg}, {"lstm_fuse_in", &h->dev.lstm_fuse_in}, {"rvq_exact", &h->dev.rvq_exact}, {"prof_detail", &h->dev.prof_detail}, {"head_seq", &h->dev.head_seq}, {"attn_exact", &h->dev.attn_exact}, {"dac_unit", &h->dev.dac_unit}, {"mimi_tail", &h->dev.mimi_tail},
    };
#ifndef AC_DEVELOPER
    // timing modes with wrong results and fault injection exist in the developer library only (libaudiocodecs_amd_dev.so)
    if (std::strcmp(key, "rb6_dbg") == 0 || std::strcmp(key, "lstm_dbg") == 0)
        return fail(h, AC_EINVAL, "ac_debug_set('%s'): a developer-build switch (timing modes with wrong results / fault injection); this is the product library", key);
#endif
    // (before ac_finalize a value would be overwritten when finalize latches the environment; under stream capture a flipped kernel
    //  path would be baked into part of a graph)
    if (!h->finalized) return fail(h, AC_ESTATE, "ac_debug_set('%s'): the switches are latched by ac_finalize; set them afterwards", key);
    for (auto& t : tab)
        if (std::strcmp(t.k, key) == 0) {
            const bool nonneg = t.v == &h->dev.front_seg || t.v == &h->dev.tail_seg || t.v == &h->dev.front_ldspad || t.v == &h->dev.tap_stagger;
            if (nonneg && value < 0) return fail(h, AC_EINVAL, "ac_debug_set('%s', %d): the value must be >= 0", key, value);
            if (t.v == &h->dev.front_ldspad && value > 16384) return fail(h, AC_EINVAL, "ac_debug_set('front_ldspad', %d): at most 16384 bytes", value);
            if (t.v == &h->dev.tap8_form && (value < 0 || value > 5)) return fail(h, AC_EINVAL, "ac_debug_set('tap8_form', %d): 0 (cost model) .. 5", value);
            *t.v = value;
            return AC_OK;
        }
    return fail(h, AC_EINVAL, "ac_debug_set: unknown switch '%s'", key);
}

int ac_set_precision(ac_handle* h, int precision) {
    if (!h) return AC_EINVAL;
    if (h->finalized) return fail(h, AC_ESTATE, "ac_set_precision must precede ac_finalize (weights are packed for one arithmetic)");
    if (precision != AC_PRECISION_FP32 && precision != AC_PRECISION_FP32_EXACT) return fail(h, AC_EINVAL, "unknown precision %d", precision);
    h->precision = precision;
    return AC_OK;
}

int ac_finalize(ac_handle* h) {
    if (!h) return AC_EINVAL;
    if (h->finalized) return fail(h, AC_ESTATE, "handle already finalized");
    {   // arithmetic of the GEMM-shaped kernels: ac_set_precision, else the environment variable AC_GEMM (fp32)
        int pr = h->precision;
        if (pr < 0) {
            const char* gm = std::getenv("AC_GEMM");
            if (gm && *gm && std::strcmp(gm, "fp32") != 0 && std::strcmp(gm, "split16") != 0)     // (bf16 / bf16x3 named modes that no longer exist)
                return fail(h, AC_EINVAL, "AC_GEMM=%s: unknown arithmetic (fp32 = exact fp32 products, split16 or unset = the default)", gm);
            pr = gm && std::strcmp(gm, "fp32") == 0 ? AC_PRECISION_FP32_EXACT : AC_PRECISION_FP32;
        }
        const char* fz = std::getenv("AC_FUSE");
        h->fuse_chains = !(fz && fz[0] == '0');
        latch_dev_switches(h);
        h->gemm_fp32 = pr == AC_PRECISION_FP32_EXACT;
    }
    if (h->arch == ARCH_MIMI) {
        Packer pk{h};
        if (int rc = mimi_finalize(h, pk)) return rc;
        return upload_blob(h, pk, h->mcfg.device);
    }
    if (h->arch == ARCH_DAC) {
        Packer pk{h};
        if (int rc = dac_finalize(h, pk)) return rc;
        return upload_blob(h, pk, h->dcfg.device);
    }
    if (h->arch == ARCH_WAVTOK) {
        Packer pk{h};
        if (int rc = wavtok_finalize(h, pk)) return rc;
        return upload_blob(h, pk, h->wcfg.device);
    }
    const ac_config& c = h->cfg;
    Arch a = make_arch(c);
    Packer pk{h};
    // a caller in encode-only / decode-only mode hands over just that half (encodec.py:67-71 deletes the other one)
    h->has_enc = h->has_dec = false;
    for (const auto& kv : h->host) {
        if (kv.first.compare(0, 8, "encoder.") == 0) h->has_enc = true;
        if (kv.first.compare(0, 8, "decoder.") == 0) h->has_dec = true;
    }
    if (!h->has_enc && !h->has_dec) return fail(h, AC_ESTATE, "no encoder.* or decoder.* tensor was loaded");
    bool ok = true;
    h->enc_rb.resize(c.num_ratios);
    h->enc_down.resize(c.num_ratios);
    h->dec_up.resize(c.num_ratios);
    h->dec_rb.resize(c.num_ratios);
    if (h->has_enc) {
        ok = pk.conv(a.enc_stem, h->enc_stem);
        for (int i = 0; ok && i < c.num_ratios; ++i) {
            ok = ok && pk.resblock(a.enc_rb3[i], a.enc_rb1[i], a.enc_rbs[i], h->enc_rb[i]);
            ok = ok && pk.conv(a.enc_down[i], h->enc_down[i]);
        }
        ok = ok && pk.lstm(a.enc_lstm, a.D, c.num_lstm_layers, h->enc_lstm);
        ok = ok && pk.conv(a.enc_final, h->enc_final);
        if (ok) pk.chain_bounds(h->enc_stem, h->enc_rb[0], h->enc_down[0], h->enc_front);
        if (ok) pk.stream_enc(h->enc_stem, h->enc_rb[0], h->enc_down[0], h->simg);
    }
    if (h->has_dec) {
        ok = ok && pk.conv(a.dec_first, h->dec_first);
        ok = ok && pk.lstm(a.dec_lstm, a.D, c.num_lstm_layers, h->dec_lstm);
        for (int i = 0; ok && i < c.num_ratios; ++i) {
            ok = ok && pk.convtr(a.dec_up[i], h->dec_up[i]);
            ok = ok && pk.resblock(a.dec_rb3[i], a.dec_rb1[i], a.dec_rbs[i], h->dec_rb[i]);
        }
        ok = ok && pk.conv(a.dec_head, h->dec_head);
        if (ok) pk.tail_bounds(h->dec_up[c.num_ratios - 1], h->dec_rb[c.num_ratios - 1], h->dec_head, h->dec_tail);
        if (ok) pk.stream_dec(h->dec_up[c.num_ratios - 1], h->dec_rb[c.num_ratios - 1], h->dec_head, h->simg);
    }
    if (!ok) return pk.rc;
    // codebooks: plain [K][C][H], MFMA B-fragment order, squared norms
    const int C = c.codebook_size, H = c.hidden_size, Q = c.num_quantizers;
    h->cb_plain = pk.reserve((size_t)Q * C * H);
    h->cb_packed = pk.reserve((size_t)Q * C * H);
    h->cb_ee = pk.reserve((size_t)Q * C);
    for (int q = 0; q < Q; ++q) {
        const std::vector<float>* e = pk.get("quantizer.layers." + std::to_string(q) + ".codebook.embed", (size_t)C * H);
        if (!e) return pk.rc;
        std::copy(e->begin(), e->end(), pk.blob.begin() + h->cb_plain + (size_t)q * C * H);
        for (int code = 0; code < C; ++code) {
            double ss = 0.0;
            for (int d = 0; d < H; ++d) ss += (double)(*e)[(size_t)code * H + d] * (*e)[(size_t)code * H + d];
            pk.blob[h->cb_ee + (size_t)q * C + code] = (float)ss;
        }
        const int HV = H / 16;
        for (int ct = 0; ct < C / 16; ++ct)
            for (int v = 0; v < HV; ++v)
                for (int lane = 0; lane < 64; ++lane)
                    for (int u = 0; u < 4; ++u)
                        pk.blob[h->cb_packed + (size_t)q * C * H + (((size_t)ct * HV + v) * 64 + lane) * 4 + u] =
                            (*e)[(size_t)(ct * 16 + (lane & 15)) * H + v * 16 + 4 * (lane >> 4) + u];
    }
    if (pk.use16() && H % 32 == 0 && H / 16 == 8) pk.pack_cb16(h->cb_plain, Q, C, H, &h->cb16, &h->cb16_inv);     // rvq16.h
    return upload_blob(h, pk, c.device);
}

int ac_num_frames(const ac_handle* h, int T) {
    if (!h || T < 1) return AC_EINVAL;
    if (h->arch == ARCH_MIMI) return cdiv(mimi_num_frames25(h->mcfg, T), h->mcfg.resample_stride);
    if (h->arch == ARCH_DAC) return dac_num_frames(h->dcfg, T);   // 0: too short for the strided convs
    long long L = T;
    for (int r = h->cfg.num_ratios - 1; r >= 0; --r) L = (L + h->cfg.upsampling_ratios[r] - 1) / h->cfg.upsampling_ratios[r];
    return (int)L;
}
long long ac_num_samples(const ac_handle* h, int N) {
    if (!h || N < 1) return AC_EINVAL;
    return h->arch == ARCH_DAC ? dac_num_samples(h->dcfg, N) : (long long)N * h->hop;
}
int ac_hop_length(const ac_handle* h) { return h ? h->hop : AC_EINVAL; }
int ac_hidden_size(const ac_handle* h) {
    if (!h) return AC_EINVAL;
    return h->arch == ARCH_MIMI ? h->mcfg.hidden_size : h->arch == ARCH_DAC ? h->dac.H : h->cfg.hidden_size;
}
int ac_codebook_dim(const ac_handle* h) {
    if (!h) return AC_EINVAL;
    return h->arch == ARCH_MIMI ? h->mcfg.codebook_dim : h->arch == ARCH_DAC ? DAC_CODE_DIM : h->cfg.hidden_size;
}

static int num_q(const ac_handle* h) {
    return h->arch == ARCH_MIMI ? h->mcfg.num_quantizers : h->arch == ARCH_DAC ? h->dcfg.n_codebooks : h->cfg.num_quantizers;
}

static Workspace any_plan_ws(const ac_handle* h, int B, int T, int N, bool enc) {
    if (h->arch == ARCH_DAC) return dac_plan_ws(h, B, T, N, enc);
    if (h->arch == ARCH_WAVTOK) return wavtok_plan_ws(h, B, T, N, enc);
    return h->arch == ARCH_MIMI ? mimi_plan_ws(h, B, T, N, enc) : plan_ws(h, B, T, N, enc);
}

// DAC: clips go through the stack in chunks that keep the workspace bounded (dac_path.h)
static int dac_encode_impl(ac_handle* h, const float* sig, int B, int T, int K, float* z_out, float* zlat_out, long long* toks, float* qfeats,
                           void* ws, size_t ws_bytes, hipStream_t st) {
    const int N = dac_num_frames(h->dcfg, T);
    if (N < 1) return fail(h, AC_EINVAL, "input too short: %d samples give no frame (hop %d)", T, h->hop);
    const int Bc = dac_chunk_clips(h, B, T, 0, true), H = h->dac.H;
    for (int b0 = 0; b0 < B; b0 += Bc) {
        const int nb = std::min(Bc, B - b0);
        WsPtrs p;
        int rc = carve(h, dac_plan_ws(h, B, T, 0, true), ws, ws_bytes, &p);
        if (rc) return rc;
        float* z = z_out ? z_out + (size_t)b0 * N * H : p.act[NACT - 1];
        if (!z_out) p.used[NACT - 1] = true;
        if ((rc = amax_begin(h, st, nb))) return rc;      // every chunk starts with fresh amax slots (no activation view crosses chunks)
        rc = dac_encoder_fwd(h, st, sig + (size_t)b0 * T, nb, T, z, p);
        if (rc) return rc;
        if (zlat_out) {
            rc = dac_latent_proj(h, st, z, N, nb, zlat_out + (size_t)b0 * N * DAC_CODE_DIM);
            if (rc) return rc;
        }
        if (toks) {
            rc = dac_vq_encode(h, st, z, nb * N, K, toks + (size_t)b0 * N * K, qfeats ? qfeats + (size_t)b0 * N * H : nullptr);
            if (rc) return rc;
        }
    }
    return AC_OK;
}

size_t ac_encode_workspace_bytes(const ac_handle* h, int B, int T) {
    if (!h || B < 1 || T < 1) return 0;
    return any_plan_ws(h, B, T, 0, true).total_bytes;
}
size_t ac_decode_workspace_bytes(const ac_handle* h, int B, int N) {
    if (!h || B < 1 || N < 1) return 0;
    return any_plan_ws(h, B, 0, N, false).total_bytes;
}
size_t ac_quantizer_workspace_bytes(const ac_handle* h, int B, int N) {
    if (!h || B < 1 || N < 1 || h->arch != ARCH_MIMI) return 0;
    return align_up((size_t)B * N * 2 * h->mcfg.codebook_dim * sizeof(float) + 256, 256) + pool_bytes(1, (size_t)B * N) + 256;
}

// Mimi's quantizer entry points: scratch = [projection | split16 pool (row ring for the B*N frames)]
static float* quantizer_scratch(ac_handle* h, void* ws, int B, int N) {
    char* base = reinterpret_cast<char*>(align_up(reinterpret_cast<uintptr_t>(ws), 256));
    const size_t proj = align_up((size_t)B * N * 2 * h->mcfg.codebook_dim * sizeof(float), 256);
    pool_bind(h, base + proj, 1, (size_t)B * N);
    return reinterpret_cast<float*>(base);
}

int ac_encode_feats(ac_handle* h, const float* sig, const float* rel_len, int B, int T, float* feats, void* ws, size_t ws_bytes, void* stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (!sig || !feats || B < 1 || T < 1) return fail(h, AC_EINVAL, "ac_encode_feats: bad argument (B=%d, T=%d)", B, T);
    if ((rc = check_len(h, T))) return rc;
    if (!h->has_enc) return fail(h, AC_ESTATE, "ac_encode_feats: the handle was loaded without encoder weights (mode=\"decode\")");
    if (h->arch == ARCH_DAC) return dac_encode_impl(h, sig, B, T, 0, feats, nullptr, nullptr, nullptr, ws, ws_bytes, (hipStream_t)stream);
    WsPtrs p;
    rc = carve(h, any_plan_ws(h, B, T, 0, true), ws, ws_bytes, &p);
    if (rc) return rc;
    if ((rc = amax_begin(h, (hipStream_t)stream, B))) return rc;
    if (h->arch == ARCH_MIMI) return mimi_encoder_fwd(h, (hipStream_t)stream, sig, B, T, feats, p);   // no sample mask in Mimi ([HF] mimi :1245-1247)
    return encoder_fwd(h, (hipStream_t)stream, sig, rel_len, B, T, feats, p);
}

int ac_encode(ac_handle* h, const float* sig, const float* rel_len, int B, int T, int K, int64_t* toks, void* ws, size_t ws_bytes, void* stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (!sig || !toks || B < 1 || T < 1) return fail(h, AC_EINVAL, "ac_encode: bad argument (B=%d, T=%d)", B, T);
    if (K < 1 || K > num_q(h)) return fail(h, AC_EINVAL, "ac_encode: K=%d outside [1, %d]", K, num_q(h));
    if ((rc = check_len(h, T))) return rc;
    if (!h->has_enc) return fail(h, AC_ESTATE, "ac_encode: the handle was loaded without encoder weights (mode=\"decode\")");
    if (h->arch == ARCH_DAC)
        return dac_encode_impl(h, sig, B, T, K, nullptr, nullptr, reinterpret_cast<long long*>(toks), nullptr, ws, ws_bytes, (hipStream_t)stream);
    WsPtrs p;
    rc = carve(h, any_plan_ws(h, B, T, 0, true), ws, ws_bytes, &p);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    if ((rc = amax_begin(h, st, B))) return rc;
    const int N = ac_num_frames(h, T);
    if (h->arch == ARCH_MIMI) {
        const ac_mimi_config& c = h->mcfg;
        float* feats = p.act[NACT - 1];   // the encoder never has more than 5 buffers live
        p.used[NACT - 1] = true;
        rc = mimi_encoder_fwd(h, st, sig, B, T, feats, p);
        if (rc) return rc;
        float* proj = p.take();
        rc = mimi_linear(h, st, h->mimi.in_proj, feats, (long long)B * N, c.hidden_size, c.hidden_size, 0, proj, 2 * c.codebook_dim);
        if (rc) return rc;
        return mimi_rvq_encode(h, st, proj, B * N, K, reinterpret_cast<long long*>(toks));
    }
    // feats land in the activation buffer the encoder's last conv does not read from
    float* feats = p.lstm.gin;  // free again once the LSTM is done
    rc = encoder_fwd(h, st, sig, rel_len, B, T, feats, p);
    if (rc) return rc;
    return rvq_encode_fwd(h, st, feats, B * N, K, reinterpret_cast<long long*>(toks));
}

int ac_quantize_ws(ac_handle* h, const float* feats, int B, int N, int K, int64_t* toks, void* ws, size_t ws_bytes, void* stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (!feats || !toks || B < 1 || N < 1 || K < 1 || K > num_q(h)) return fail(h, AC_EINVAL, "ac_quantize: bad argument");
    pool_bind(h, nullptr, 0, 0);      // (EnCodec / DAC: the codebook search runs outside split-operand arithmetic)
    if (h->arch == ARCH_DAC) return dac_vq_encode(h, (hipStream_t)stream, feats, B * N, K, reinterpret_cast<long long*>(toks), nullptr);
    if (h->arch == ARCH_MIMI) {
        const ac_mimi_config& c = h->mcfg;
        if (!ws || ws_bytes < ac_quantizer_workspace_bytes(h, B, N)) return fail(h, AC_ENOMEM, "ac_quantize: workspace missing or too small");
        float* proj = quantizer_scratch(h, ws, B, N);
        if ((rc = amax_begin(h, (hipStream_t)stream, 1))) return rc;
        rc = mimi_linear(h, (hipStream_t)stream, h->mimi.in_proj, feats, (long long)B * N, c.hidden_size, c.hidden_size, 0, proj, 2 * c.codebook_dim);
        if (rc) return rc;
        return mimi_rvq_encode(h, (hipStream_t)stream, proj, B * N, K, reinterpret_cast<long long*>(toks));
    }
    return rvq_encode_fwd(h, (hipStream_t)stream, feats, B * N, K, reinterpret_cast<long long*>(toks));
}

int ac_quantize(ac_handle* h, const float* feats, int B, int N, int K, int64_t* toks, void* stream) {
    if (h && h->arch == ARCH_MIMI) return fail(h, AC_EINVAL, "ac_quantize: Mimi handles need scratch, call ac_quantize_ws");
    return ac_quantize_ws(h, feats, B, N, K, toks, nullptr, 0, stream);
}

int ac_dequantize_ws(ac_handle* h, const int64_t* toks, int B, int N, int K, float* qfeats, void* ws, size_t ws_bytes, void* stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (!qfeats || !toks || B < 1 || N < 1 || K < 1 || K > num_q(h)) return fail(h, AC_EINVAL, "ac_dequantize: bad argument");
    pool_bind(h, nullptr, 0, 0);
    if (h->arch == ARCH_DAC) return dac_from_codes(h, (hipStream_t)stream, reinterpret_cast<const long long*>(toks), B * N, K, qfeats);
    if (h->arch == ARCH_MIMI) {
        if (!ws || ws_bytes < ac_quantizer_workspace_bytes(h, B, N)) return fail(h, AC_ENOMEM, "ac_dequantize: workspace missing or too small");
        float* qsum = quantizer_scratch(h, ws, B, N);
        if ((rc = amax_begin(h, (hipStream_t)stream, 1))) return rc;
        return mimi_rvq_decode(h, (hipStream_t)stream, reinterpret_cast<const long long*>(toks), B * N, K, qsum, qfeats);
    }
    return rvq_decode_fwd(h, (hipStream_t)stream, reinterpret_cast<const long long*>(toks), B * N, K, qfeats);
}

int ac_dequantize(ac_handle* h, const int64_t* toks, int B, int N, int K, float* qfeats, void* stream) {
    if (h && h->arch == ARCH_MIMI) return fail(h, AC_EINVAL, "ac_dequantize: Mimi handles need scratch, call ac_dequantize_ws");
    return ac_dequantize_ws(h, toks, B, N, K, qfeats, nullptr, 0, stream);
}

int ac_decode(ac_handle* h, const int64_t* toks, int B, int N, int K, float* sig, void* ws, size_t ws_bytes, void* stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (!sig || !toks || B < 1 || N < 1) return fail(h, AC_EINVAL, "ac_decode: bad argument (B=%d, N=%d)", B, N);
    if (K < 1 || K > num_q(h)) return fail(h, AC_EINVAL, "ac_decode: K=%d outside [1, %d]", K, num_q(h));
    if ((rc = check_len(h, ac_num_samples(h, N)))) return rc;
    if (!h->has_dec) return fail(h, AC_ESTATE, "ac_decode: the handle was loaded without decoder weights (mode=\"encode\")");
    if (h->arch == ARCH_DAC) {
        hipStream_t st = (hipStream_t)stream;
        const int Bc = dac_chunk_clips(h, B, 0, N, false);
        const long long Lout = dac_num_samples(h->dcfg, N);
        if (Lout < 1) return fail(h, AC_EINVAL, "ac_decode: %d frames decode to no samples", N);
        for (int b0 = 0; b0 < B; b0 += Bc) {
            const int nb = std::min(Bc, B - b0);
            WsPtrs p;
            rc = carve(h, dac_plan_ws(h, B, 0, N, false), ws, ws_bytes, &p);
            if (rc) return rc;
            float* zq = p.take();
            if ((rc = amax_begin(h, st, nb))) return rc;  // every chunk starts with fresh amax slots
            rc = dac_from_codes(h, st, reinterpret_cast<const long long*>(toks) + (size_t)b0 * N * K, nb * N, K, zq);
            if (rc) return rc;
            capture(h, st, Act{zq, (long long)N * h->dac.H, h->dac.H, N, h->dac.H}, nb);
            rc = dac_decoder_fwd(h, st, zq, nb, N, sig + (size_t)b0 * Lout, p);
            if (rc) return rc;
        }
        return AC_OK;
    }
    WsPtrs p;
    rc = carve(h, any_plan_ws(h, B, 0, N, false), ws, ws_bytes, &p);
    if (rc) return rc;
    if ((rc = amax_begin(h, (hipStream_t)stream, B))) return rc;
    if (h->arch == ARCH_WAVTOK) {
        hipStream_t st = (hipStream_t)stream;
        float* zq = p.take();   // codes_to_features: the single codebook's vectors [B*N][dimension]
        rc = rvq_decode_fwd(h, st, reinterpret_cast<const long long*>(toks), B * N, K, zq);
        if (rc) return rc;
        return wavtok_decoder_fwd(h, st, zq, B, N, sig, p);
    }
    if (h->arch == ARCH_MIMI) {
        hipStream_t st = (hipStream_t)stream;
        float* qsum = p.take();
        float* qf = p.take();
        rc = mimi_rvq_decode(h, st, reinterpret_cast<const long long*>(toks), B * N, K, qsum, qf);
        if (rc) return rc;
        p.give(qsum);
        capture(h, st, Act{qf, (long long)N * h->mcfg.hidden_size, h->mcfg.hidden_size, N, h->mcfg.hidden_size}, B);
        // qf stays taken while the up-sampler reads it; the decoder needs at most 5 more buffers
        return mimi_decoder_fwd(h, st, qf, B, N, sig, p);
    }
    return decoder_fwd(h, (hipStream_t)stream, reinterpret_cast<const long long*>(toks), B, N, K, sig, p);
}

int ac_decode_feats(ac_handle* h, const float* feats, int B, int N, float* sig, void* ws, size_t ws_bytes, void* stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (h->arch != ARCH_WAVTOK) return fail(h, AC_EINVAL, "ac_decode_feats: WavTokenizer handles only (the other wrappers do not implement _feats_to_sig)");
    if (!feats || !sig || B < 1 || N < 1) return fail(h, AC_EINVAL, "ac_decode_feats: bad argument (B=%d, N=%d)", B, N);
    if ((rc = check_len(h, ac_num_samples(h, N)))) return rc;
    if (!h->has_dec) return fail(h, AC_ESTATE, "ac_decode_feats: the handle was loaded without decoder weights (mode=\"encode\")");
    WsPtrs p;
    rc = carve(h, wavtok_plan_ws(h, B, 0, N, false), ws, ws_bytes, &p);
    if (rc) return rc;
    if ((rc = amax_begin(h, (hipStream_t)stream, B))) return rc;
    return wavtok_decoder_fwd(h, (hipStream_t)stream, feats, B, N, sig, p);
}

int ac_embs(ac_handle* h, int K, float* embs, void* stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (!embs || K < 1 || K > num_q(h)) return fail(h, AC_EINVAL, "ac_embs: bad argument");
    size_t n, off;
    if (h->arch == ARCH_MIMI) { n = (size_t)K * h->mcfg.codebook_size * h->mcfg.codebook_dim; off = h->mimi.cb_plain; }
    else if (h->arch == ARCH_DAC) { n = (size_t)K * h->dcfg.codebook_size * DAC_CODE_DIM; off = h->dac.cb; }
    else { n = (size_t)K * h->cfg.codebook_size * h->cfg.hidden_size; off = h->cb_plain; }
    HIPCHK(h, hipMemcpyAsync(embs, h->blob + off, n * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return AC_OK;
}

int ac_embs_projected(ac_handle* h, int K, float* embs, void* stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (h->arch == ARCH_DAC) {   // out_proj_k(codebook_k) + bias, tabulated at ac_finalize (dac.py:68-90)
        if (!embs || K < 1 || K > h->dcfg.n_codebooks) return fail(h, AC_EINVAL, "ac_embs_projected: bad argument");
        const size_t n = (size_t)K * h->dcfg.codebook_size * h->dac.H;
        HIPCHK(h, hipMemcpyAsync(embs, h->blob + h->dac.proj, n * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
        return AC_OK;
    }
    if (h->arch != ARCH_MIMI) return fail(h, AC_EINVAL, "ac_embs_projected: EnCodec has no output projection");
    const ac_mimi_config& c = h->mcfg;
    if (!embs || K < 1 || K > c.num_quantizers) return fail(h, AC_EINVAL, "ac_embs_projected: bad argument");
    // the one launching entry point without a workspace argument: its few KB of pool were allocated at ac_finalize
    pool_bind(h, h->own_pool, 1, h->own_pool_rows);
    if ((rc = amax_begin(h, (hipStream_t)stream, 1))) return rc;
    for (int q = 0; q < K; ++q) {
        const int part = q < c.num_semantic_quantizers ? 0 : 1;
        rc = mimi_linear(h, (hipStream_t)stream, h->mimi.out_proj, h->blob + h->mimi.cb_plain + (size_t)q * c.codebook_size * c.codebook_dim,
                         c.codebook_size, c.codebook_dim, c.codebook_dim, part * c.codebook_dim,
                         embs + (size_t)q * c.codebook_size * c.hidden_size, c.hidden_size);
        if (rc) return rc;
    }
    return AC_OK;
}

int ac_encode_quantized(ac_handle* h, const float* sig, int B, int T, int K, int64_t* toks, float* qfeats, void* ws, size_t ws_bytes, void* stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (h->arch != ARCH_DAC) return fail(h, AC_EINVAL, "ac_encode_quantized: DAC handles only (others: ac_encode + ac_dequantize_ws)");
    if (!sig || !toks || !qfeats || B < 1 || T < 1 || K < 1 || K > num_q(h)) return fail(h, AC_EINVAL, "ac_encode_quantized: bad argument");
    if ((rc = check_len(h, T))) return rc;
    return dac_encode_impl(h, sig, B, T, K, nullptr, nullptr, reinterpret_cast<long long*>(toks), qfeats, ws, ws_bytes, (hipStream_t)stream);
}

int ac_encode_feats_latent(ac_handle* h, const float* sig, int B, int T, float* feats_latent, void* ws, size_t ws_bytes, void* stream) {
    int rc = check_ready(h);
    if (rc) return rc;
    if (h->arch != ARCH_DAC) return fail(h, AC_EINVAL, "ac_encode_feats_latent: DAC handles only");
    if (!sig || !feats_latent || B < 1 || T < 1) return fail(h, AC_EINVAL, "ac_encode_feats_latent: bad argument");
    if ((rc = check_len(h, T))) return rc;
    return dac_encode_impl(h, sig, B, T, 0, nullptr, feats_latent, nullptr, nullptr, ws, ws_bytes, (hipStream_t)stream);
}

int ac_poll_status(ac_handle* h, void* stream) {
    if (!h) return AC_EINVAL;
    HIPCHK(h, hipStreamSynchronize((hipStream_t)stream));
    return check_ready(h);
}

int ac_lstm_status(ac_handle* h) {
    if (!h) return AC_EINVAL;
    if (!h->lp_ctl) return 0;
    if (hipDeviceSynchronize() != hipSuccess) return AC_EHIP;
    const bool usable = (h->arch == ARCH_ENCODEC || h->arch == ARCH_WAVTOK) && (h->enc_lstm.has_persist || h->dec_lstm.has_persist) && h->num_cus == 256 && !h->lstm_step_only;
    volatile unsigned* s = h->sticky;
    if (s && (s[ST_LSTM_TIMEOUT] || s[ST_LSTM_PLACEMENT]))
        return fail(h, AC_EHIP, "persistent LSTM: %u bounded waits expired, %u misplaced launches (outputs of those calls are NaN)", s[ST_LSTM_TIMEOUT], s[ST_LSTM_PLACEMENT]);
    return usable ? 1 : 0;
}

int ac_debug_split_row(const float* w, int n, uint16_t* hi, uint16_t* lo) {
    if (!w || n < 1) return AC_EINVAL;
    const int s = Packer::row_scale(w, (size_t)n);
    for (int k = 0; k < n; ++k) {
        uint16_t t[3];
        Packer::split16h(w[k], s, t);
        if (hi) hi[k] = t[0];
        if (lo) lo[k] = t[1];
    }
    return s;
}

int ac_debug_bounds(const ac_handle* h, float* out, int cap) {
    // Layout (EnCodec handles; 0 where a half was not loaded):
    //   [0..6]   enc_front.h: sb0 sb1 | hb0 hb1 | fb0 fb1h fb1x        |x0| <= sb0 + sb1 amax(sig); |h| <= hb0 + hb1 X0; |y1| <= fb0 + fb1h H + fb1x X0
    //   [7..10]  dec_tail.h:  ub0 ub1 | hb0 hb1                        |u| <= ub0 + ub1 amax(xe);   |h| <= hb0 + hb1 U
    //   [11 + 2i], [12 + 2i]  hb0, hb1 of encoder residual block i, then of decoder residual block i (AC_MAX_RATIOS each):
    //            |hidden| <= hb0 + hb1 amax(block input)  (rb_fused6.h / rb_fused6_128.h)
    if (!h || !out || !h->finalized || h->arch != ARCH_ENCODEC) return AC_EINVAL;
    const int n = 11 + 4 * AC_MAX_RATIOS;
    if (cap < n) return AC_EINVAL;
    for (int i = 0; i < n; ++i) out[i] = 0.f;
    const int last = h->cfg.num_ratios - 1;
    if (h->enc_front.ok) {
        const float v[7] = {h->enc_front.sb0, h->enc_front.sb1, h->enc_rb[0].hb0, h->enc_rb[0].hb1, h->enc_front.fb0, h->enc_front.fb1h, h->enc_front.fb1x};
        for (int i = 0; i < 7; ++i) out[i] = v[i];
    }
    if (h->dec_tail.ok) {
        const float v[4] = {h->dec_tail.sb0, h->dec_tail.sb1, h->dec_rb[last].hb0, h->dec_rb[last].hb1};
        for (int i = 0; i < 4; ++i) out[7 + i] = v[i];
    }
    for (int i = 0; i < h->cfg.num_ratios; ++i) {
        if (h->has_enc && h->enc_rb[i].has6) { out[11 + 2 * i] = h->enc_rb[i].hb0; out[12 + 2 * i] = h->enc_rb[i].hb1; }
        if (h->has_dec && h->dec_rb[i].has6) { out[11 + 2 * AC_MAX_RATIOS + 2 * i] = h->dec_rb[i].hb0; out[12 + 2 * AC_MAX_RATIOS + 2 * i] = h->dec_rb[i].hb1; }
    }
    return n;
}

int ac_debug_clock(ac_handle* h, int enable, double* shader_mhz) {
    if (!h || !h->finalized) return AC_EINVAL;
    if (shader_mhz) *shader_mhz = 0.0;
    if (h->clk_dev) {
        unsigned long long v[2] = {0, 0};
        HIPCHK(h, hipDeviceSynchronize());
        HIPCHK(h, hipMemcpy(v, h->clk_dev, sizeof v, hipMemcpyDeviceToHost));
        if (shader_mhz && v[1]) *shader_mhz = 100.0 * (double)v[0] / (double)v[1];
        HIPCHK(h, hipMemset(h->clk_dev, 0, sizeof v));
    }
    if (enable && !h->clk_dev) {
        HIPCHK(h, hipMalloc(reinterpret_cast<void**>(&h->clk_dev), 16 + 8 * (16 + 8 * 16 * 8)));   // (+ the stage stamps of a T6_TRACE build)
        HIPCHK(h, hipMemset(h->clk_dev, 0, 16 + 8 * (16 + 8 * 16 * 8)));
    } else if (!enable && h->clk_dev) {
        (void)hipFree(h->clk_dev);
        h->clk_dev = nullptr;
    }
    return AC_OK;
}

int ac_debug_trace(ac_handle* h, unsigned long long* out, int words) {   // developer (T6_TRACE builds): the stage stamps behind the clock words
    if (!h || !h->clk_dev || !out || words < 1) return AC_EINVAL;
    HIPCHK(h, hipDeviceSynchronize());
    const int n = std::min(words, 16 + 8 * 16 * 8);
    HIPCHK(h, hipMemcpy(out, h->clk_dev, (size_t)n * 8, hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemset(h->clk_dev + 4, 0, (size_t)(12 + 8 * 16 * 8) * 8));
    return n;
}

int ac_debug_capture(ac_handle* h, float* buf_dev, size_t cap_floats) {
    if (!h) return AC_EINVAL;
    h->dbg = buf_dev;
    h->dbg_cap = buf_dev ? cap_floats : 0;
    h->dbg_used = 0;
    return AC_OK;
}

size_t ac_debug_captured(const ac_handle* h) { return h ? h->dbg_used : 0; }

int ac_profile_begin(ac_handle* h) {
    if (!h) return AC_EINVAL;
    h->prof = true;
    h->prof_detail = h->dev.prof_detail != 0;
    h->recs.clear();
    h->ev_used = 0;
    return AC_OK;
}

int ac_profile_end(ac_handle* h, ac_kernel_stat* out, int cap) {
    if (!h) return AC_EINVAL;
    h->prof = false;
    std::vector<ac_kernel_stat> st(h->prof_names.size());
    for (size_t i = 0; i < st.size(); ++i) {
        std::memset(&st[i], 0, sizeof st[i]);
        std::snprintf(st[i].name, sizeof st[i].name, "%s", h->prof_names[i].c_str());
    }
    for (const ProfRec& r : h->recs) {
        if (hipEventSynchronize(r.e1) != hipSuccess) return fail(h, AC_EHIP, "hipEventSynchronize failed");
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) return fail(h, AC_EHIP, "hipEventElapsedTime failed");
        st[r.name_id].launches += r.count;
        st[r.name_id].total_ms += ms;
        st[r.name_id].flops += r.flops;
        st[r.name_id].bytes += r.bytes;
    }
    h->recs.clear();
    h->ev_used = 0;
    int n = 0;
    for (size_t i = 0; i < st.size() && n < cap; ++i)
        if (st[i].launches) out[n++] = st[i];
    return n;
}

int ac_resample(const float* x, int B, int L, const float* kern, int n, int o, int taps, int width, float* y, int L_out, void* stream) {
    if (!x || !kern || !y || B < 1 || L < 1 || n < 1 || o < 1 || taps < 1 || width < 0 || L_out < 1) return AC_EINVAL;
    return resample_launch(x, B, L, kern, n, o, taps, width, y, L_out, (hipStream_t)stream);
}

const char* ac_last_error(const ac_handle* h) { return h ? h->err.c_str() : "null handle"; }

void ac_destroy(ac_handle* h) {
    if (!h) return;
    if (h->blob) (void)hipFree(h->blob);
    if (h->lp_ctl) (void)hipFree(h->lp_ctl);
    if (h->own_pool) (void)hipFree(h->own_pool);
    if (h->sticky) (void)hipHostFree(h->sticky);
    if (h->clk_dev) (void)hipFree(h->clk_dev);
    for (hipEvent_t e : h->ev_pool) (void)hipEventDestroy(e);
    delete h;
}

}  // extern "C"
