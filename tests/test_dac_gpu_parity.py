"""GPU parity for DAC: the HIP path through the C ABI (audiocodecs_amd.DAC) against the
stand-in fixtures (transformers.DacModel called as the reference wrapper calls dac.DAC -- the reference's
own backend is not installed, parity with it is unpinned) and against the CPU oracle on seeded inputs."""
import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR
from dac_cases import CASES, REC_STRIDE, make_input
from golden_cases import noise
from test_gpu_parity import capture, rms
from test_oracle_golden import TAU, tokens_match_up_to_ties
import parity_record

pytestmark = pytest.mark.gpu


def enc_taps(nb):
    t = ["encoder.conv1"]
    for i in range(nb):
        t += [f"encoder.block.{i}.res_unit{u}" for u in (1, 2, 3)] + [f"encoder.block.{i}.conv1"]
    return t + ["encoder.conv2"]


def dec_taps(nb):
    t = ["from_codes", "decoder.conv1"]
    for i in range(nb):
        t += [f"decoder.block.{i}.conv_t1"] + [f"decoder.block.{i}.res_unit{u}" for u in (1, 2, 3)]
    return t


@pytest.fixture(scope="module")
def codecs(dac_checkpoints):
    from audiocodecs_amd import DAC

    cache = {}

    def get(cfg_name, seed, K=8, latent=False):
        key = (cfg_name, seed, K, latent)
        if key not in cache:
            cfg, sd = dac_checkpoints(cfg_name, seed)
            cache[key] = DAC(cfg.sampling_rate, cfg.sampling_rate, num_codebooks=K, latent=latent, state_dict=sd, config=cfg).eval()
        return cache[key]

    return get


def compare_taps(z, meta, name, taps, flat, atol, skip_shape=None):
    off = 0
    shapes = meta["cases"][name]["act_shapes"]
    for tap in taps:
        if tap == "from_codes":
            off += int(np.prod(skip_shape))
            continue
        shape = shapes[tap]                       # hooks saw [B,C,L]; ours is [B,L,C]
        n = int(np.prod(shape))
        got = flat[off : off + n].reshape(shape[0], shape[2], shape[1]).transpose(0, 2, 1).reshape(-1)
        step = 1 if n <= meta["act_full_max"] else meta["act_stride"]
        np.testing.assert_allclose(got[::step], z[f"{name}.act.{tap}"], atol=atol, rtol=1e-5, err_msg=tap)
        off += n
    return off


@pytest.mark.parametrize("name", ["tiny_taps", "tiny_odd"])
def test_every_module_output_matches_standin_hooks(name, dac_golden, codecs):
    z, meta = dac_golden
    case = next(c for c in CASES if c["name"] == name)
    info = meta["cases"][name]
    codec = codecs("tiny", 0, info["K"])
    sig = make_input(case, GOLDEN_DIR)["sig"].cuda()
    codec.sig_to_toks(sig[:, :1024])  # creates the native handle
    toks, flat = capture(codec, lambda: codec.sig_to_toks(sig))
    off = compare_taps(z, meta, name, enc_taps(4), flat, 1e-5)
    assert off == flat.size
    gold = z[f"{name}.toks"].astype(np.int64)
    n, bad, excused = tokens_match_up_to_ties(toks.cpu().numpy(), gold, z[f"{name}.margin64"])
    assert bad == 0
    gt = torch.from_numpy(gold).cuda()
    B, N, K = gold.shape
    rec, flat = capture(codec, lambda: codec.toks_to_sig(gt))
    off = compare_taps(z, meta, name, dec_taps(4), flat, 1e-4, skip_shape=(B, N, codec.config.hidden_size))  # 19 layers deep, values O(2)
    assert off == flat.size
    assert list(rec.shape) == info["rec_shape"]
    err = rec.cpu().numpy().reshape(-1)[::REC_STRIDE] - z[f"{name}.rec_strided"]
    assert rms(err) < 2e-5


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_golden_fixture(case, dac_golden, codecs):
    z, meta = dac_golden
    name = case["name"]
    info = meta["cases"][name]
    K = info["K"]
    codec = codecs(case["cfg"], case["weights_seed"], K)
    inp = make_input(case, GOLDEN_DIR)
    if case["kind"] == "decode":
        toks = inp["toks"].cuda()
    else:
        sig = inp["sig"].cuda()
        toks = codec.sig_to_toks(sig)
        assert toks.dtype == torch.int64 and list(toks.shape) == info["toks_shape"]
        gold = z[f"{name}.toks"].astype(np.int64)
        margin = z[f"{name}.margin64"]
        mism, bad, excused = parity_record.tokens("dac", name, toks.cpu().numpy(), gold, margin, TAU)
        assert bad == 0, f"{bad} tokens differ outside near-ties"
        if margin.min() > TAU:       # no near-tie anywhere in the fixture: bit-exact
            assert np.array_equal(toks.cpu().numpy(), gold)
        assert mism <= excused
        feats = codec.sig_to_feats(sig).cpu().numpy()
        err = feats.reshape(-1)[::REC_STRIDE] - z[f"{name}.feats_strided"]
        assert rms(err) < 3e-5 and np.abs(err).max() < 5e-4, (rms(err), np.abs(err).max())
        lat = codecs(case["cfg"], case["weights_seed"], K, True).sig_to_feats(sig).cpu().numpy()
        np.testing.assert_allclose(lat.reshape(-1)[::7], z[f"{name}.feats_latent"], atol=5e-5)
        if np.array_equal(toks.cpu().numpy(), gold):
            qf = codec.sig_to_qfeats(sig).cpu().numpy()
            np.testing.assert_allclose(qf.reshape(-1)[::REC_STRIDE], z[f"{name}.qfeats_fwd_strided"], atol=5e-5)
        toks = torch.from_numpy(gold).cuda()  # decode the stand-in's tokens
    rec = codec.toks_to_sig(toks).cpu().numpy()
    assert list(rec.shape) == info["rec_shape"]
    err = rec.reshape(-1)[::REC_STRIDE] - z[f"{name}.rec_strided"]
    parity_record.record("dac", name, waveform_rms_err=rms(err))
    assert rms(err) < 1e-4, rms(err)
    assert rms(err) < 3e-5, rms(err)
    assert abs(rms(rec) - info["rec_rms"]) < 1e-4
    if f"{name}.embs_latent_strided" in z.files:
        es = meta["embs_stride"]
        for latent, key in ((True, "embs_latent_strided"), (False, "embs_proj_strided")):
            e = codecs(case["cfg"], case["weights_seed"], K, latent).embs()
            assert list(e.shape) == info["embs_shapes"][0 if latent else 1]
            np.testing.assert_allclose(e.cpu().numpy().reshape(-1)[::es], z[f"{name}.{key}"], rtol=0, atol=3e-6)


def test_against_oracle_on_fresh_inputs(codecs, dac_checkpoints):
    from oracle import dac_oracle as O

    cfg, sd = dac_checkpoints("full", 0)
    codec = codecs("full", 0, 9)
    W, W64 = O.cast_weights(sd), O.cast_weights(sd, torch.float64)
    sig = noise(2977, 3, 20011)
    with torch.no_grad():
        otoks = O.sig_to_toks(cfg, W, sig, None, 9)
        _, m64 = O.sig_to_toks(cfg, W64, sig.double(), None, 9, "descript", True)
        orec = O.toks_to_sig(cfg, W, otoks)
        ofeats = O.sig_to_feats(cfg, W, sig)
    toks = codec.sig_to_toks(sig.cuda())
    n, bad, excused = tokens_match_up_to_ties(toks.cpu().numpy(), otoks.numpy(), m64.numpy())
    assert bad == 0, f"{bad}/{n}"
    assert rms(codec.sig_to_feats(sig.cuda()).cpu().numpy() - ofeats.numpy()) < 3e-5
    rec = codec.toks_to_sig(otoks.cuda()).cpu().numpy()
    assert rec.shape == tuple(orec.shape) and rms(rec - orec.numpy()) < 3e-5
    out = codec(sig.cuda())  # Codec.forward, mode "reconstruct"
    assert out.shape == rec.shape


def test_batch_independence_and_determinism(codecs):
    codec = codecs("full", 0, 9)
    sig = noise(3024, 5, 9000).cuda()
    a = codec.sig_to_toks(sig)
    assert torch.equal(a, codec.sig_to_toks(sig))
    assert torch.equal(a[2:3], codec.sig_to_toks(sig[2:3]))
    r = codec.toks_to_sig(a)
    assert torch.equal(r[4:5], codec.toks_to_sig(a[4:5]))


def test_errors_and_codebook_clamp(codecs, dac_checkpoints):
    from audiocodecs_amd import DAC, _native

    cfg, sd = dac_checkpoints("tiny", 0)
    codec = codecs("tiny", 0, 4)
    with pytest.raises(RuntimeError):
        codec.sig_to_toks(noise(5, 1, 100).cuda())          # shorter than the strided convs allow (upstream: conv1d raises)
    with pytest.raises(_native.NativeError):
        codec.sig_to_toks(noise(5, 1, 4000))                # CPU tensor: no fallback
    many = DAC(cfg.sampling_rate, cfg.sampling_rate, num_codebooks=50, state_dict=sd, config=cfg)
    assert many.sig_to_toks(noise(5, 1, 4000).cuda()).shape[-1] == cfg.n_codebooks   # upstream's loop just runs out of quantisers
    with pytest.raises(ImportError):
        DAC(16000)                                          # no bundled weights


def test_dilated_taps_from_one_slab_equal_the_reload_per_tap_path(dac_checkpoints, monkeypatch):
    """csrc/tap_gemm6.h T6_DIL_HALO: the dilated k7 convs of the residual units read their seven taps from one wide A slab per
    chunk (the instantiation run_tap picks where it measured faster) instead of reloading the slab per tap.  Both walk
    (chunk, tap) in the same order, so tokens AND waveform must be bit-equal between the two (ac_debug_set tap_dil = 0: reload path),
    also at lengths that put clip edges inside the halo."""
    from audiocodecs_amd import DAC
    from audiocodecs_amd._native import debug_set

    cfg, sd = dac_checkpoints("full", 0)
    codec = DAC(44100, 44100, num_codebooks=9, state_dict=sd, config=cfg).eval()
    for B, T in ((2, 8192), (3, 5003), (1, 700)):
        sig = noise(5150 + T, B, T).cuda()
        names = {s[0] for s in codec.profile_kernels(lambda: codec.toks_to_sig(codec.sig_to_toks(sig)))}
        assert any(", dil>" in n for n in names), names              # the wide-slab instantiation is what runs by default (decoder: 96 / 192 / 384 / 768 channels)
        toks, rec = codec.sig_to_toks(sig), None
        rec = codec.toks_to_sig(toks)
        debug_set(codec, "tap_dil", 0)
        names0 = {s[0] for s in codec.profile_kernels(lambda: codec.toks_to_sig(codec.sig_to_toks(sig)))}
        assert not any(", dil>" in n for n in names0), names0
        toks0 = codec.sig_to_toks(sig)
        rec0 = codec.toks_to_sig(toks)
        debug_set(codec, "tap_dil", 1)
        assert torch.equal(toks, toks0), (B, T)
        assert torch.equal(rec, rec0), (B, T)


@pytest.mark.parametrize("codec_name", ["dac", "encodec"])
def test_direct_epilogue_equals_the_staged_one(codec_name, dac_checkpoints, checkpoints, monkeypatch):
    """csrc/tap_gemm6.h: conv outputs leave the accumulators either through the direct epilogue (one 4-byte store per value;
    plain / ELU flavours of 128-column layers, and -- with residual / Snake copies -- DAC's layers under 128 channels) or
    through LDS-staged 16-byte rows (ac_debug_set tap_epi_staged = 1 forces it everywhere).  The same operations in the same order per
    element: tokens and waveform must be bit-equal, edge tiles included."""
    from audiocodecs_amd import DAC, Encodec
    from audiocodecs_amd._native import debug_set

    if codec_name == "dac":
        cfg, sd = dac_checkpoints("full", 0)
        codec = DAC(44100, 44100, num_codebooks=9, state_dict=sd, config=cfg).eval()
        shapes = ((2, 6151), (1, 1024))
    else:
        cfg, sd = checkpoints("full", 0)
        codec = Encodec(24000, num_codebooks=8, state_dict=sd, config=cfg).eval()
        shapes = ((3, 9999), (1, 24000))
    for B, T in shapes:
        sig = noise(6160 + T, B, T).cuda()
        toks = codec.sig_to_toks(sig)
        rec = codec.toks_to_sig(toks)
        debug_set(codec, "tap_epi_staged", 1)
        toks0 = codec.sig_to_toks(sig)
        rec0 = codec.toks_to_sig(toks)
        debug_set(codec, "tap_epi_staged", 0)
        assert torch.equal(toks, toks0), (codec_name, B, T)
        assert torch.equal(rec, rec0), (codec_name, B, T)
