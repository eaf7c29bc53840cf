"""Pins the Mimi oracle (oracle/mimi_oracle.py) to the reference wrapper's own outputs.

Fixtures: tools/make_golden_mimi.py running /root/reference's audiocodecs.mimi.Mimi (sig_to_toks /
toks_to_sig / sig_to_feats / toks_to_qfeats / embs, plus forward hooks on every module for the tiny
config) on seeded synthetic weights.  CPU-only; runs in the `-m "not gpu"` suite.
"""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR
from mimi_cases import CASES, REC_STRIDE, make_input
from oracle import mimi_oracle as O
from test_oracle_golden import TAU, tokens_match_up_to_ties

CPU_CASES = [c for c in CASES if c["name"] != "full_example"]  # the 10.6 s clip runs in the gpu suite's oracle leg


def tap_key(k: str) -> str:
    """oracle tap name -> fixture key: encoder.layers.3 -> enc3, decoder_transformer.layers.1 -> dectr1."""
    for long, short in (("encoder_transformer.layers.", "enctr"), ("decoder_transformer.layers.", "dectr"),
                        ("encoder.layers.", "enc"), ("decoder.layers.", "dec")):
        if k.startswith(long):
            return short + k[len(long):]
    return k


def strided(a: np.ndarray, meta) -> np.ndarray:
    a = a.reshape(-1)
    return a[:: (1 if a.size <= meta["act_full_max"] else meta["act_stride"])]


@pytest.mark.parametrize("case", CPU_CASES, ids=[c["name"] for c in CPU_CASES])
def test_oracle_matches_reference_fixture(case, mimi_golden, mimi_checkpoints):
    z, meta = mimi_golden
    name = case["name"]
    cfg, sd = mimi_checkpoints(case["cfg"], case["weights_seed"])
    W = O.cast_weights(sd)
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
            np.testing.assert_allclose(feats.numpy().reshape(-1)[::REC_STRIDE], z[f"{name}.feats_strided"], rtol=0, atol=3e-5)
            toks = torch.from_numpy(gold)  # decode the reference's tokens: decode is pinned on its own
        qf = O.toks_to_qfeats(cfg, W, toks)
        np.testing.assert_allclose(qf.numpy().reshape(-1)[::REC_STRIDE], z[f"{name}.qfeats_strided"], rtol=0, atol=1e-5)
        rec = O.toks_to_sig(cfg, W, toks)
    assert list(rec.shape) == info["rec_shape"]
    r = rec.numpy()
    err = r.reshape(-1)[::REC_STRIDE] - z[f"{name}.rec_strided"]
    assert np.sqrt(np.mean(err.astype(np.float64) ** 2)) < 2e-5
    assert abs(np.sqrt(np.mean(r.astype(np.float64) ** 2)) - info["rec_rms"]) < 2e-5
    if f"{name}.embs_latent_strided" in z.files:
        es = meta["embs_stride"]
        for latent, key in ((True, "embs_latent_strided"), (False, "embs_proj_strided")):
            e = O.embs(cfg, W, info["K"], latent)
            assert list(e.shape) == info["embs_shapes"][0 if latent else 1]
            np.testing.assert_allclose(e.numpy().reshape(-1)[::es], z[f"{name}.{key}"], rtol=0, atol=2e-6)


@pytest.mark.parametrize("name", ["tiny_taps", "tiny_odd"])
def test_oracle_intermediates_match_reference_hooks(name, mimi_golden, mimi_checkpoints):
    z, meta = mimi_golden
    case = next(c for c in CASES if c["name"] == name)
    cfg, sd = mimi_checkpoints("tiny", 0)
    W = O.cast_weights(sd)
    inp = make_input(case, GOLDEN_DIR)
    taps = {}
    with torch.no_grad():
        feats = O.sig_to_feats(cfg, W, inp["sig"], taps=taps)
        dtaps = {}
        rec = O.toks_to_sig(cfg, W, torch.from_numpy(z[f"{name}.toks"].astype(np.int64)), taps=dtaps)
    taps.update(dtaps)
    np.testing.assert_allclose(feats.numpy(), z[f"{name}.feats"], atol=5e-6)
    gold_rec = z[f"{name}.rec_full"]
    step = 1 if rec.numel() <= meta["act_full_max"] else meta["act_stride"]
    np.testing.assert_allclose(rec.numpy()[:, ::step], gold_rec, atol=5e-6)
    checked = 0
    shapes = meta["cases"][name]["act_shapes"]
    for k, v in taps.items():
        fk = tap_key(k)
        if fk == "quantizer.decode":
            continue
        key = f"{name}.act.{fk}"
        assert key in z.files, key
        if fk.startswith(("enctr", "dectr")):
            assert list(v.shape) == shapes[fk]
        np.testing.assert_allclose(strided(v.numpy(), meta), z[key], atol=5e-6, err_msg=k)
        checked += 1
    assert checked == 26  # 10 encoder + 10 decoder modules, 4 transformer layers, down/up-sample


def test_sliding_window_is_exercised(mimi_checkpoints):
    """MIMI_TINY.sliding_window (6) is shorter than the 10 frames of tiny_taps: the mask matters."""
    cfg, sd = mimi_checkpoints("tiny", 0)
    W = O.cast_weights(sd, torch.float64)
    x = torch.randn(1, 10, cfg.hidden_size, dtype=torch.float64, generator=torch.Generator().manual_seed(0))
    a = O.transformer(cfg, W, x, "encoder_transformer")
    wide = {**cfg.__dict__, "sliding_window": 250}
    b = O.transformer(wide, W, x, "encoder_transformer")
    assert torch.allclose(a[:, :6], b[:, :6], atol=1e-12) and not torch.allclose(a[:, 6:], b[:, 6:], atol=1e-6)


def test_num_codebooks_bounds(mimi_checkpoints):
    cfg, sd = mimi_checkpoints("tiny", 0)
    W = O.cast_weights(sd)
    z = torch.zeros(1, cfg.hidden_size, 2)
    with pytest.raises(ValueError):
        O.rvq_encode(cfg, W, z, 33)
    with pytest.raises(ValueError):
        O.rvq_encode(cfg, W, z, 0)
