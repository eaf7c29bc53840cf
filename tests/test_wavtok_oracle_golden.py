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
