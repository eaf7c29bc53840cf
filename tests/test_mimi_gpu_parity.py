"""GPU parity for Mimi: the HIP path through the C ABI (audiocodecs_amd.Mimi) against
(a) the reference-generated golden fixtures and (b) the CPU oracle on the same seeded inputs.
Bars as for EnCodec: tokens equal wherever the fp64 margin exceeds TAU at this and earlier stages of the
frame (remainder counted and bounded), waveform within 1e-4 RMS.
"""
import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR
from golden_cases import noise
from mimi_cases import CASES, REC_STRIDE, make_input
from test_gpu_parity import capture, rms
from test_oracle_golden import TAU, tokens_match_up_to_ties
import parity_record

pytestmark = pytest.mark.gpu

# capture order of the HIP path == HF module order (ELU modules have no output of their own here)
ENC_TAPS = ["enc0", "enc1", "enc3", "enc4", "enc6", "enc7", "enc9", "enc10", "enc12", "enc14", "enctr0", "enctr1", "downsample"]
DEC_TAPS = ["qdecode", "upsample", "dectr0", "dectr1", "dec0", "dec2", "dec3", "dec5", "dec6", "dec8", "dec9", "dec11", "dec12"]


@pytest.fixture(scope="module")
def codecs(mimi_checkpoints):
    from audiocodecs_amd import Mimi

    cache = {}

    def get(cfg_name, seed, K=8, latent=True):
        key = (cfg_name, seed, K, latent)
        if key not in cache:
            cfg, sd = mimi_checkpoints(cfg_name, seed)
            cache[key] = Mimi(24000, num_codebooks=K, latent=latent, state_dict=sd, config=cfg).eval()
        return cache[key]

    return get


def gold_act(z, meta, name, tap):
    """Fixture activation -> (flat strided values, channels-last selector) for comparison."""
    shape = meta["cases"][name]["act_shapes"][tap]
    numel = int(np.prod(shape))
    step = 1 if numel <= meta["act_full_max"] else meta["act_stride"]
    return shape, step, z[f"{name}.act.{tap}"]


def compare_taps(z, meta, name, taps, flat, atol):
    off = 0
    for tap in taps:
        if tap == "qdecode":   # quantizer.decode output: not hooked in the fixture, [B][N][hidden] here
            K = meta["cases"][name]["K"]
            B, N = meta["cases"][name]["toks_shape"][:2]
            shape = meta["cases"][name]["act_shapes"]["downsample"]  # [B, hidden, N]
            off += shape[0] * shape[1] * shape[2]
            continue
        shape, step, g = gold_act(z, meta, name, tap)
        n = int(np.prod(shape))
        got = flat[off : off + n]
        if tap.startswith(("enctr", "dectr")):       # hooks saw [B,T,H]: same layout as ours
            got = got.reshape(shape)
        else:                                        # hooks saw [B,C,L]; ours is [B,L,C]
            got = got.reshape(shape[0], shape[2], shape[1]).transpose(0, 2, 1)
        np.testing.assert_allclose(got.reshape(-1)[::step], g, atol=atol, rtol=1e-5, err_msg=tap)
        off += n
    return off


@pytest.mark.parametrize("name", ["tiny_taps", "tiny_odd"])
def test_every_module_output_matches_reference_hooks(name, mimi_golden, codecs):
    z, meta = mimi_golden
    case = next(c for c in CASES if c["name"] == name)
    K = meta["cases"][name]["K"]
    codec = codecs("tiny", 0, K)
    sig = make_input(case, GOLDEN_DIR)["sig"].cuda()
    codec.sig_to_toks(sig[:, :64])  # creates the native handle
    toks, flat = capture(codec, lambda: codec.sig_to_toks(sig))
    off = compare_taps(z, meta, name, ENC_TAPS, flat, 5e-6)
    assert off == flat.size
    gold = z[f"{name}.toks"].astype(np.int64)
    n, bad, excused = tokens_match_up_to_ties(toks.cpu().numpy(), gold, z[f"{name}.margin64"])
    assert bad == 0
    gt = torch.from_numpy(gold).cuda()
    rec, flat = capture(codec, lambda: codec.toks_to_sig(gt))
    off = compare_taps(z, meta, name, DEC_TAPS, flat, 1e-5)
    assert off == flat.size
    gold_rec = z[f"{name}.rec_full"]
    step = 1 if rec.numel() <= meta["act_full_max"] else meta["act_stride"]
    np.testing.assert_allclose(rec.cpu().numpy()[:, ::step], gold_rec, atol=1e-5)
    feats = codec.sig_to_feats(sig)
    np.testing.assert_allclose(feats.cpu().numpy(), z[f"{name}.feats"], atol=1e-5)


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_golden_fixture(case, mimi_golden, codecs):
    z, meta = mimi_golden
    name = case["name"]
    info = meta["cases"][name]
    codec = codecs(case["cfg"], case["weights_seed"], info["K"])
    inp = make_input(case, GOLDEN_DIR)
    if case["kind"] == "decode":
        toks = inp["toks"].cuda()
    else:
        sig = inp["sig"].cuda()
        length = inp["length"].cuda() if "length" in inp else None
        toks = codec.sig_to_toks(sig, length)
        assert toks.dtype == torch.int64 and list(toks.shape) == info["toks_shape"]
        gold = z[f"{name}.toks"].astype(np.int64)
        margin = z[f"{name}.margin64"]
        mism, bad, excused = parity_record.tokens("mimi", name, toks.cpu().numpy(), gold, margin, TAU)
        assert bad == 0, f"{bad} tokens differ outside near-ties"
        if margin.min() > TAU:       # no near-tie anywhere in the fixture: bit-exact
            assert np.array_equal(toks.cpu().numpy(), gold)
        assert mism <= excused
        feats = codec.sig_to_feats(sig, length).cpu().numpy()
        err = feats.reshape(-1)[::REC_STRIDE] - z[f"{name}.feats_strided"]
        assert rms(err) < 2e-5 and np.abs(err).max() < 3e-4, (rms(err), np.abs(err).max())
        toks = torch.from_numpy(gold).cuda()  # decode the REFERENCE's tokens
    qf = codec.toks_to_qfeats(toks).cpu().numpy()
    np.testing.assert_allclose(qf.reshape(-1)[::REC_STRIDE], z[f"{name}.qfeats_strided"], atol=2e-5)
    rec = codec.toks_to_sig(toks).cpu().numpy()
    assert list(rec.shape) == info["rec_shape"]
    err = rec.reshape(-1)[::REC_STRIDE] - z[f"{name}.rec_strided"]
    parity_record.record("mimi", name, waveform_rms_err=rms(err))
    assert rms(err) < 1e-4, rms(err)          # the north-star bar
    assert rms(err) < 2e-5, rms(err)          # what fp32 parity mode delivers
    assert abs(rms(rec) - info["rec_rms"]) < 1e-4
    if f"{name}.embs_latent_strided" in z.files:
        es = meta["embs_stride"]
        for latent, key in ((True, "embs_latent_strided"), (False, "embs_proj_strided")):
            e = codecs(case["cfg"], case["weights_seed"], info["K"], latent).embs()
            assert list(e.shape) == info["embs_shapes"][0 if latent else 1]
            np.testing.assert_allclose(e.cpu().numpy().reshape(-1)[::es], z[f"{name}.{key}"], rtol=0, atol=3e-6)


def test_against_oracle_on_fresh_inputs(codecs, mimi_checkpoints):
    """Seeded inputs that are in no fixture, odd batch and length: HIP vs the CPU oracle (fp32), fp64 margins."""
    from oracle import mimi_oracle as O

    cfg, sd = mimi_checkpoints("full", 0)
    codec = codecs("full", 0)
    W, W64 = O.cast_weights(sd), O.cast_weights(sd, torch.float64)
    sig = noise(1977, 3, 30011)
    with torch.no_grad():
        otoks = O.sig_to_toks(cfg, W, sig)
        _, m64 = O.sig_to_toks(cfg, W64, sig.double(), None, 8, True)
        orec = O.toks_to_sig(cfg, W, otoks)
        ofeats = O.sig_to_feats(cfg, W, sig)
    toks = codec.sig_to_toks(sig.cuda())
    n, bad, excused = tokens_match_up_to_ties(toks.cpu().numpy(), otoks.numpy(), m64.numpy())
    assert bad == 0, f"{bad}/{n}"
    feats = codec.sig_to_feats(sig.cuda()).cpu().numpy()
    assert rms(feats - ofeats.numpy()) < 2e-5
    rec = codec.toks_to_sig(otoks.cuda()).cpu().numpy()
    assert rec.shape == tuple(orec.shape)
    assert rms(rec - orec.numpy()) < 2e-5
    # whole-API round trip through the Codec base class
    out = codec(sig.cuda())
    assert out.shape == rec.shape


def test_batch_independence_and_determinism(codecs):
    codec = codecs("full", 0)
    sig = noise(2024, 5, 9000).cuda()
    a = codec.sig_to_toks(sig)
    b = codec.sig_to_toks(sig)
    assert torch.equal(a, b)
    one = codec.sig_to_toks(sig[2:3])
    assert torch.equal(a[2:3], one)           # clips do not interact
    r = codec.toks_to_sig(a)
    assert torch.equal(r[4:5], codec.toks_to_sig(a[4:5]))


def test_errors(codecs, mimi_checkpoints):
    from audiocodecs_amd import Mimi, _native

    cfg, sd = mimi_checkpoints("tiny", 0)
    sig = noise(5, 1, 4000)
    for K in (0, 33):
        with pytest.raises(ValueError):
            Mimi(24000, num_codebooks=K, state_dict=sd, config=cfg).sig_to_toks(sig.cuda())
    with pytest.raises(_native.NativeError):
        codecs("tiny", 0).sig_to_toks(sig)     # CPU tensor: no fallback
    with pytest.raises(ValueError):
        Mimi(24000, mode="bogus", state_dict=sd, config=cfg)
