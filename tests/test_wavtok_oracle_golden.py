"""WavTokenizer oracle (oracle/wavtokenizer_oracle.py) -- PARITY UNPINNED w.r.t. the reference (backend package not on
disk).  What CAN be checked on CPU:
  * the oracle reproduces the committed fixtures (pins it across rounds; tools/make_golden_wavtok.py wrote them);
  * its SEANet encoder + codebook search agree with an INDEPENDENT third-party implementation of the same published
    modules: transformers' EncodecModel with use_causal_conv=False (both are ports of facebook's encodec library, whose
    SEANetEncoder WavTokenizer embeds) holding the same weights;
  * its ISTFT(padding="same") inverts the matching STFT exactly (the property the published module is built on).
"""
import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR
from oracle import wavtokenizer_oracle as O
from test_oracle_golden import TAU, tokens_match_up_to_ties
from wavtok_cases import CASES, REC_STRIDE, make_input


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_oracle_matches_its_fixture(case, wavtok_golden, wavtok_checkpoints):
    z, meta = wavtok_golden
    name = case["name"]
    cfg, sd = wavtok_checkpoints(case["cfg"], case["weights_seed"])
    W = O.cast_weights(sd)
    inp = make_input(case, GOLDEN_DIR)
    info = meta["cases"][name]
    torch.set_num_threads(min(8, torch.get_num_threads()))
    with torch.no_grad():
        if case["kind"] == "decode":
            toks = inp["toks"]
        else:
            toks = O.sig_to_toks(cfg, W, inp["sig"])
            gold = z[f"{name}.toks"].astype(np.int64)
            assert list(toks.shape) == info["toks_shape"] and toks.dtype == torch.int64
            n, bad, excused = tokens_match_up_to_ties(toks.numpy(), gold, z[f"{name}.margin64"])
            assert bad == 0
            feats = O.sig_to_feats(cfg, W, inp["sig"])
            np.testing.assert_allclose(feats.numpy().reshape(-1)[::REC_STRIDE], z[f"{name}.feats_strided"], rtol=0, atol=2e-5)
            toks = torch.from_numpy(gold)
        rec = O.toks_to_sig(cfg, W, toks)
    assert list(rec.shape) == info["rec_shape"]
    err = rec.numpy().reshape(-1)[::REC_STRIDE] - z[f"{name}.rec_strided"]
    assert np.sqrt(np.mean(err.astype(np.float64) ** 2)) < 1e-5


def _hf_noncausal_encodec(cfg, sd):
    transformers = pytest.importorskip("transformers")
    from transformers import EncodecConfig, EncodecModel

    hc = EncodecConfig(
        sampling_rate=cfg.sampling_rate, audio_channels=1, num_filters=cfg.num_filters, hidden_size=cfg.dimension,
        codebook_dim=cfg.dimension, upsampling_ratios=list(cfg.ratios), kernel_size=cfg.kernel_size,
        last_kernel_size=cfg.last_kernel_size, residual_kernel_size=cfg.residual_kernel_size, compress=cfg.compress,
        num_lstm_layers=cfg.num_lstm_layers, codebook_size=cfg.codebook_size, use_causal_conv=False, pad_mode="reflect",
        norm_type="weight_norm", use_conv_shortcut=True, normalize=False,
    )
    model = EncodecModel(hc).eval()
    pre = "feature_extractor.encodec.encoder.model."
    mapped = {}
    for k, v in sd.items():
        if k.startswith(pre):
            k2 = "encoder.layers." + k[len(pre):].replace(".conv.conv.", ".conv.")
            k2 = k2.replace(".weight_g", ".parametrizations.weight.original0").replace(".weight_v", ".parametrizations.weight.original1")
            mapped[k2] = v
    mapped["quantizer.layers.0.codebook.embed"] = sd["feature_extractor.encodec.quantizer.vq.layers.0._codebook.embed"]
    missing, unexpected = model.load_state_dict(mapped, strict=False)
    assert not unexpected
    assert all(m.startswith("decoder.") or m.startswith("quantizer.layers.") for m in missing), missing
    return model


@pytest.mark.parametrize("cfg_name,T", [("tiny", 1111), ("tiny", 5), ("full", 4801), ("f75", 2000)])
def test_encoder_and_codebook_search_agree_with_hf_noncausal_seanet(cfg_name, T, wavtok_checkpoints):
    cfg, sd = wavtok_checkpoints(cfg_name, 0)
    model = _hf_noncausal_encodec(cfg, sd)
    W = O.cast_weights(sd)
    sig = torch.randn(2, T, generator=torch.Generator().manual_seed(T)) * 0.1
    with torch.no_grad():
        ref = model.encoder(sig[:, None])                       # [B, dimension, N]
        got = O.encoder(cfg, W, sig[:, None])
        assert ref.shape == got.shape and got.shape[-1] == cfg.num_frames(T)
        np.testing.assert_allclose(got.numpy(), ref.numpy(), atol=3e-6, rtol=1e-5)
        ref_idx = model.quantizer.layers[0].codebook.encode(ref.permute(0, 2, 1))     # EuclideanCodebook.quantize on [B,N,D]
        idx = O.vq_encode(O.codebook(W), ref)
        assert torch.equal(idx, ref_idx.view_as(idx))


@pytest.mark.parametrize("n_fft,hop", [(2400, 600), (1280, 320), (192, 48)])
def test_istft_same_inverts_the_matching_stft(n_fft, hop):
    g = torch.Generator().manual_seed(n_fft)
    N = 23
    x = torch.randn(2, N * hop, generator=g, dtype=torch.float64)
    w = torch.hann_window(n_fft, dtype=torch.float64)
    pad = (n_fft - hop) // 2
    xp = torch.nn.functional.pad(x, (pad, pad))
    spec = torch.stft(xp, n_fft, hop, n_fft, w, center=False, return_complex=True)    # [B, bins, N]
    assert spec.shape[-1] == N
    y = O.istft_same(spec, n_fft, hop, w)
    assert y.shape == x.shape
    np.testing.assert_allclose(y.numpy(), x.numpy(), atol=1e-10)


# ---------------------------------------------------------------------------------------------------------------------
# Round 3: three more modules of the oracle against INDEPENDENT third-party implementations that are on disk
# (transformers 5.15.0).  Still "parity unpinned" w.r.t. the reference's own backend -- but the Vocos side of the oracle no
# longer rests on recollection alone.
# ---------------------------------------------------------------------------------------------------------------------
def _seeded(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def test_head_matches_xcodec2_istft_head():
    """oracle head() (Vocos ISTFTHead, padding="same") vs transformers' Xcodec2ISTFTHead
    (models/xcodec2/modeling_xcodec2.py:746-796: a port of Vocos' head -- Linear -> (log-magnitude, phase) halves ->
    exp, clamp 100 -> polar -> irfft x hann -> fold -> trim (n_fft - hop)/2 -> divide by the folded squared window)."""
    xc = pytest.importorskip("transformers.models.xcodec2.modeling_xcodec2")
    from types import SimpleNamespace

    for n_fft, hop, dim, N in ((16, 4, 12, 9), (2400, 600, 32, 7), (1280, 320, 24, 5)):
        hf = xc.Xcodec2ISTFTHead(SimpleNamespace(hidden_size=dim, n_fft=n_fft, hop_length=hop)).eval()
        w, b = _seeded((n_fft + 2, dim), 1, 0.3 / dim ** 0.5), _seeded((n_fft + 2,), 2, 0.3)
        with torch.no_grad():
            hf.linear.weight.copy_(w)
            hf.linear.bias.copy_(b)
            x = _seeded((2, N, dim), 3)
            want = hf(x)[:, 0]
            cfg = SimpleNamespace(n_fft=n_fft, hop_length=hop)
            got = O.head(cfg, {"head.out.weight": w, "head.out.bias": b}, x)
        assert got.shape == want.shape == (2, N * hop)
        np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=0, atol=2e-6 * float(want.abs().max()))


def test_resnet_block_matches_xcodec2_resnet_block():
    """oracle resnet_block() (pos_net ResnetBlock: x + conv2(swish(GN(conv1(swish(GN(x)))))), GroupNorm(32, eps 1e-6), k3 pad 1)
    vs transformers' Xcodec2ResNetBlock (models/xcodec2/modeling_xcodec2.py:639-661) with the same weights."""
    xc = pytest.importorskip("transformers.models.xcodec2.modeling_xcodec2")
    from types import SimpleNamespace

    C, N = 64, 11
    hf = xc.Xcodec2ResNetBlock(SimpleNamespace(hidden_size=C, activation_dropout=0.1)).eval()
    W = {}
    with torch.no_grad():
        for i, (mod, nm) in enumerate(((hf.norm1, "norm1"), (hf.norm2, "norm2"))):
            mod.weight.copy_(1.0 + _seeded((C,), 10 + i, 0.2))
            mod.bias.copy_(_seeded((C,), 20 + i, 0.2))
            W[f"p.{nm}.weight"], W[f"p.{nm}.bias"] = mod.weight.clone(), mod.bias.clone()
        for i, (mod, nm) in enumerate(((hf.conv1, "conv1"), (hf.conv2, "conv2"))):
            mod.weight.copy_(_seeded((C, C, 3), 30 + i, (3 * C) ** -0.5))
            mod.bias.copy_(_seeded((C,), 40 + i, 0.1))
            W[f"p.{nm}.weight"], W[f"p.{nm}.bias"] = mod.weight.clone(), mod.bias.clone()
        x = _seeded((2, C, N), 50)
        want = hf(x.transpose(1, 2)).transpose(1, 2)          # the HF module takes [B, N, C]
        got = O.resnet_block(x, W, "p", 32)
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=0, atol=3e-6)


def test_convnext_block_matches_transformers_convnext_layer():
    """oracle convnext() (dwconv k7 -> norm -> Linear -> GELU -> Linear -> gamma -> residual) vs transformers' ConvNextLayer
    (models/convnext/modeling_convnext.py:114-156) on a [B, C, 1, N] image: the two differ only in the norm -- LayerNorm with
    affine (weight, bias) there, AdaLayerNorm = layer_norm * scale[cond] + shift[cond] in Vocos -- so the embedding row of the
    oracle is fed the HF layer's LayerNorm weight and bias."""
    cn = pytest.importorskip("transformers.models.convnext.modeling_convnext")
    from transformers import ConvNextConfig

    C, N = 48, 13
    hf = cn.ConvNextLayer(ConvNextConfig(hidden_act="gelu", layer_scale_init_value=0.5), dim=C).eval()
    with torch.no_grad():
        # 1-D depthwise conv = the middle row of a 7x7 depthwise kernel on a height-1 image (rows 0-2, 4-6 only see padding)
        dw1 = _seeded((C, 1, 7), 60, 7 ** -0.5)
        k2 = torch.zeros(C, 1, 7, 7)
        k2[:, :, 3, :] = dw1
        hf.dwconv.weight.copy_(k2)
        hf.dwconv.bias.copy_(_seeded((C,), 61, 0.1))
        hf.layernorm.weight.copy_(1.0 + _seeded((C,), 62, 0.2))
        hf.layernorm.bias.copy_(_seeded((C,), 63, 0.2))
        hf.pwconv1.weight.copy_(_seeded((4 * C, C), 64, C ** -0.5))
        hf.pwconv1.bias.copy_(_seeded((4 * C,), 65, 0.1))
        hf.pwconv2.weight.copy_(_seeded((C, 4 * C), 66, (4 * C) ** -0.5))
        hf.pwconv2.bias.copy_(_seeded((C,), 67, 0.1))
        hf.layer_scale_parameter.copy_(_seeded((C,), 68, 0.3))
        W = {"p.dwconv.weight": dw1, "p.dwconv.bias": hf.dwconv.bias.clone(),
             "p.norm.scale.weight": torch.stack([torch.zeros(C), hf.layernorm.weight.clone()]),      # cond = 1 selects the HF affine
             "p.norm.shift.weight": torch.stack([torch.zeros(C), hf.layernorm.bias.clone()]),
             "p.pwconv1.weight": hf.pwconv1.weight.clone(), "p.pwconv1.bias": hf.pwconv1.bias.clone(),
             "p.pwconv2.weight": hf.pwconv2.weight.clone(), "p.pwconv2.bias": hf.pwconv2.bias.clone(),
             "p.gamma": hf.layer_scale_parameter.clone()}
        x = _seeded((2, C, N), 69)
        want = hf(x[:, :, None, :])[:, :, 0, :]
        got = O.convnext(x, W, "p", 1)
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=0, atol=3e-6)
