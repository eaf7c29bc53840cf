"""Pins the DAC oracle (oracle/dac_oracle.py) to the STAND-IN for the reference's backend.

`descript-audio-codec` is not installed here, so /root/reference's audiocodecs.dac.DAC cannot run; the
fixtures (tools/make_golden_dac.py) come from the same-architecture transformers.DacModel called the way the
wrapper calls dac.DAC.  Parity with the reference itself stays UNPINNED (oracle header).  CPU-only."""
import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR
from dac_cases import CASES, REC_STRIDE, make_input
from oracle import dac_oracle as O
from test_oracle_golden import TAU, tokens_match_up_to_ties


def strided(a, meta):
    a = np.asarray(a).reshape(-1)
    return a[:: (1 if a.size <= meta["act_full_max"] else meta["act_stride"])]


@pytest.mark.parametrize("variant", ["descript", "hf"])
@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_oracle_matches_standin_fixture(case, variant, dac_golden, dac_checkpoints):
    z, meta = dac_golden
    name = case["name"]
    cfg, sd = dac_checkpoints(case["cfg"], case["weights_seed"])
    W = O.cast_weights(sd)
    inp = make_input(case, GOLDEN_DIR)
    info = meta["cases"][name]
    K = info["K"]
    torch.set_num_threads(min(8, torch.get_num_threads()))
    with torch.no_grad():
        if case["kind"] == "decode":
            if variant == "hf":
                pytest.skip("decode does not depend on the search variant")
            toks = inp["toks"]
        else:
            toks = O.sig_to_toks(cfg, W, inp["sig"], None, K, variant)
            gold = z[f"{name}.toks"].astype(np.int64)
            assert list(toks.shape) == info["toks_shape"] and toks.dtype == torch.int64
            n, bad, excused = tokens_match_up_to_ties(toks.numpy(), gold, z[f"{name}.margin64"])
            assert bad == 0, f"{bad}/{n} tokens differ outside near-ties"
            feats = O.sig_to_feats(cfg, W, inp["sig"])
            np.testing.assert_allclose(feats.numpy().reshape(-1)[::REC_STRIDE], z[f"{name}.feats_strided"], rtol=0, atol=3e-5)
            lat = O.sig_to_feats(cfg, W, inp["sig"], latent=True)
            np.testing.assert_allclose(lat.numpy().reshape(-1)[::7], z[f"{name}.feats_latent"], rtol=0, atol=3e-5)
            if np.array_equal(toks.numpy(), gold):
                qf = O.sig_to_qfeats(cfg, W, inp["sig"], None, K, variant)
                np.testing.assert_allclose(qf.numpy().reshape(-1)[::REC_STRIDE], z[f"{name}.qfeats_fwd_strided"], rtol=0, atol=3e-5)
            toks = torch.from_numpy(gold)
        zq, _ = O.from_codes(cfg, W, toks.movedim(-1, -2))
        np.testing.assert_allclose(zq.movedim(-1, -2).numpy().reshape(-1)[::REC_STRIDE], z[f"{name}.qfeats_codes_strided"], rtol=0, atol=1e-5)
        rec = O.toks_to_sig(cfg, W, toks)
    assert list(rec.shape) == info["rec_shape"]
    r = rec.numpy()
    err = r.reshape(-1)[::REC_STRIDE] - z[f"{name}.rec_strided"]
    assert np.sqrt(np.mean(err.astype(np.float64) ** 2)) < 2e-5
    assert abs(np.sqrt(np.mean(r.astype(np.float64) ** 2)) - info["rec_rms"]) < 2e-5
    if f"{name}.embs_latent_strided" in z.files:
        es = meta["embs_stride"]
        for latent, key in ((True, "embs_latent_strided"), (False, "embs_proj_strided")):
            e = O.embs(cfg, W, K, latent)
            assert list(e.shape) == info["embs_shapes"][0 if latent else 1]
            np.testing.assert_allclose(e.numpy().reshape(-1)[::es], z[f"{name}.{key}"], rtol=0, atol=2e-6)


@pytest.mark.parametrize("name", ["tiny_taps", "tiny_odd"])
def test_oracle_intermediates_match_standin_hooks(name, dac_golden, dac_checkpoints):
    z, meta = dac_golden
    case = next(c for c in CASES if c["name"] == name)
    cfg, sd = dac_checkpoints("tiny", 0)
    W = O.cast_weights(sd)
    inp = make_input(case, GOLDEN_DIR)
    taps = {}
    with torch.no_grad():
        O.sig_to_feats(cfg, W, inp["sig"], taps=taps)
        O.toks_to_sig(cfg, W, torch.from_numpy(z[f"{name}.toks"].astype(np.int64)), taps=taps)
    shapes = meta["cases"][name]["act_shapes"]
    checked = 0
    for k, v in taps.items():
        if k == "quantizer.from_codes":
            continue
        assert list(v.shape) == shapes[k], k
        np.testing.assert_allclose(strided(v.numpy(), meta), z[f"{name}.act.{k}"], atol=1e-5, err_msg=k)
        checked += 1
    assert checked == len(shapes) == 36   # 2 + 4*4 encoder, 2 + 4*4 decoder modules


def test_lengths_and_short_input(dac_checkpoints):
    cfg, sd = dac_checkpoints("tiny", 0)
    W = O.cast_weights(sd)
    for T in (cfg.hop_length, 1000, 3333):
        with torch.no_grad():
            z = O.encoder(cfg, W, torch.zeros(1, 1, T))
            y = O.decoder(cfg, W, z)
        assert z.shape[-1] == cfg.num_frames(T) and y.shape[-1] == cfg.num_samples(z.shape[-1])
    with pytest.raises(RuntimeError):   # too short for the last strided conv: upstream's conv1d raises
        O.encoder(cfg, W, torch.zeros(1, 1, 100))
