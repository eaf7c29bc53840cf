"""Pins the oracle (oracle/encodec_oracle.py) to the reference wrapper's own outputs.

The fixtures were produced by tools/make_golden.py running /root/reference's
audiocodecs.encodec.Encodec (sig_to_toks / toks_to_sig / sig_to_feats, plus forward hooks on every
module for the tiny config).  CPU-only; runs in the `-m "not gpu"` suite.
"""
import numpy as np
import pytest
import torch

from golden_cases import CASES, REC_STRIDE, make_input
from conftest import GOLDEN_DIR
from oracle import encodec_oracle as O

TAU = 1e-4  # near-tie threshold on the fp64 relative margin (see DESIGN.md "near-tie policy")


def tokens_match_up_to_ties(toks, gold, margin, tau=TAU):
    """Exact equality required for every token whose frame has had no near-tie (margin<=tau) at this
    or an earlier stage.  Returns (n_checked, n_bad, n_excused)."""
    safe = np.cumprod(margin > tau, axis=-1).astype(bool)  # [B,N,K], stage order along K
    bad = (toks != gold) & safe
    return int(safe.sum()), int(bad.sum()), int((~safe).sum())


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_oracle_matches_reference_fixture(case, golden, checkpoints):
    z, meta = golden
    name = case["name"]
    cfg, sd = checkpoints(case["cfg"], case["weights_seed"])
    W = O.fold_weight_norm(sd)
    inp = make_input(case, GOLDEN_DIR)
    info = meta["cases"][name]
    torch.set_num_threads(min(8, torch.get_num_threads()))
    with torch.no_grad():
        if case["kind"] == "decode":
            toks = inp["toks"]
        else:
            toks = O.sig_to_toks(cfg, W, inp["sig"], inp.get("length"), info["K"])
            gold = z[f"{name}.toks"].astype(np.int64)
            assert list(toks.shape) == info["toks_shape"] and toks.dtype == torch.int64
            n, bad, excused = tokens_match_up_to_ties(toks.numpy(), gold, z[f"{name}.margin64"])
            assert bad == 0, f"{bad}/{n} tokens differ outside near-ties"
            feats = O.sig_to_feats(cfg, W, inp["sig"], inp.get("length"))
            np.testing.assert_allclose(
                feats.numpy().reshape(-1)[::REC_STRIDE], z[f"{name}.feats_strided"], rtol=0, atol=2e-5
            )
            toks = torch.from_numpy(gold)  # decode the reference's tokens, so decode is pinned on its own
        rec = O.toks_to_sig(cfg, W, toks)
    assert list(rec.shape) == info["rec_shape"]
    r = rec.numpy()
    err = r.reshape(-1)[::REC_STRIDE] - z[f"{name}.rec_strided"]
    assert np.sqrt(np.mean(err.astype(np.float64) ** 2)) < 1e-5
    assert abs(np.sqrt(np.mean(r.astype(np.float64) ** 2)) - info["rec_rms"]) < 1e-5


@pytest.mark.parametrize("name", ["tiny_taps", "tiny_ragged"])
def test_oracle_intermediates_match_reference_hooks(name, golden, checkpoints):
    z, meta = golden
    case = next(c for c in CASES if c["name"] == name)
    cfg, sd = checkpoints("tiny", 0)
    W = O.fold_weight_norm(sd)
    inp = make_input(case, GOLDEN_DIR)
    taps = {}
    with torch.no_grad():
        feats = O.sig_to_feats(cfg, W, inp["sig"], inp.get("length"))
        O.masked_embeddings(cfg, W, inp["sig"], inp.get("length"), taps=taps)  # hooks saw the masked pass
        dtaps = {}
        rec = O.toks_to_sig(cfg, W, torch.from_numpy(z[f"{name}.toks"].astype(np.int64)), taps=dtaps)
    taps.update(dtaps)
    np.testing.assert_allclose(feats.numpy(), z[f"{name}.feats"], atol=2e-6)
    np.testing.assert_allclose(rec.numpy(), z[f"{name}.rec_full"], atol=2e-6)
    checked = 0
    for k, v in taps.items():
        key = f"{name}.act.{k}"
        assert key in z.files, key
        np.testing.assert_allclose(v.numpy(), z[key], atol=2e-6, err_msg=k)
        checked += 1
    assert checked >= 18


def test_explicit_lstm_equals_aten_lstm(checkpoints):
    cfg, sd = checkpoints("tiny", 0)
    W = O.fold_weight_norm(sd, torch.float64)
    x = torch.randn(2, cfg.lstm_dim, 9, dtype=torch.float64, generator=torch.Generator().manual_seed(0))
    p = "encoder.layers.13.lstm"
    a = O.lstm_skip(x, W, p, 2, explicit=False)
    b = O.lstm_skip(x, W, p, 2, explicit=True)
    np.testing.assert_allclose(a.numpy(), b.numpy(), atol=1e-12)


def test_pad1d_small_input_rule():
    # [HF]:148-155: length <= max_pad -> zero-extend, reflect, drop
    x = torch.tensor([[[1.0, 2.0]]])
    y = O.pad1d_reflect(x, 6, 0)
    assert y.shape[-1] == 8
    # zero-extended to length 7: [1,2,0,0,0,0,0]; left reflect of 6 -> [0,0,0,0,0,2]
    assert y.flatten().tolist() == [0, 0, 0, 0, 0, 2, 1, 2]


def test_bad_bandwidth_raises(checkpoints):
    cfg, sd = checkpoints("tiny", 0)
    with pytest.raises(ValueError):
        O.num_quantizers_for(cfg, 3)  # 2.25 kbps is not a target bandwidth ([HF]:564-567)
