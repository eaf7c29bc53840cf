"""GPU parity of the WavTokenizer path (through the C ABI, via audiocodecs_amd.WavTokenizer) against the oracle's
fixtures and the oracle itself on fresh inputs.  PARITY UNPINNED w.r.t. the reference (see wavtok_cases.py): what is
asserted here is HIP path == oracle -- token ids bit-exact outside fp64 near-ties, waveform within 1e-4 RMS (measured
< 2e-5), every module output of the tiny and of the FULL architecture."""
import ctypes as C

import numpy as np
import pytest
import torch

import parity_record
from conftest import GOLDEN_DIR
from golden_cases import noise
from test_oracle_golden import TAU
from wavtok_cases import CASES, REC_STRIDE, make_input

pytestmark = pytest.mark.gpu

ENC_TAPS = ["enc0", "enc1", "enc3", "enc4", "enc6", "enc7", "enc9", "enc10", "enc12", "enc13"]


def dec_taps(cfg):
    return ["embed", "pos0", "pos1", "pos2", "pos3", "pos4", "pos5", "norm"] + [f"cnx{l}" for l in range(cfg.num_layers)] + ["final"]


@pytest.fixture(scope="module")
def codecs(wavtok_checkpoints):
    from audiocodecs_amd import WavTokenizer

    cache = {}

    def get(cfg_name, seed):
        key = (cfg_name, seed)
        if key not in cache:
            cfg, sd = wavtok_checkpoints(cfg_name, seed)
            cache[key] = WavTokenizer(24000, state_dict=sd, arch=cfg).eval()
        return cache[key]

    return get


def rms(a):
    return float(np.sqrt(np.mean(np.asarray(a, dtype=np.float64) ** 2)))


def capture(codec, fn, nfloats=1 << 25):
    nat = next(iter(codec._natives.values()))
    buf = torch.zeros(nfloats, device="cuda")
    nat.lib.ac_debug_capture(nat.h, C.c_void_p(buf.data_ptr()), nfloats)
    try:
        out = fn()
        torch.cuda.synchronize()
        used = nat.lib.ac_debug_captured(nat.h)
    finally:
        nat.lib.ac_debug_capture(nat.h, None, 0)
    assert used <= nfloats
    return out, buf[:used].cpu().numpy()


def check_taps(flat, names, gold_of, atol, rtol=2e-5):
    """flat: captured module outputs, each [B][L][C]; gold_of(name) -> [B,C,L] array."""
    off, worst = 0, {}
    for tap in names:
        g = gold_of(tap)
        n = g.size
        got = flat[off : off + n].reshape(g.shape[0], g.shape[2], g.shape[1]).transpose(0, 2, 1)
        scale = max(1.0, float(np.abs(g).max()))
        worst[tap] = float(np.abs(got - g).max() / scale)
        np.testing.assert_allclose(got, g, atol=atol * scale, rtol=rtol, err_msg=tap)
        off += n
    assert off == flat.size
    return worst


@pytest.mark.parametrize("name", ["tiny_taps", "tiny_odd"])
def test_every_module_output_tiny(name, wavtok_golden, codecs, wavtok_checkpoints):
    """Tiny architecture (generic kernels + the decoder's own kernels): each module output vs the oracle's."""
    from oracle import wavtokenizer_oracle as O

    z, meta = wavtok_golden
    case = next(c for c in CASES if c["name"] == name)
    codec = codecs("tiny", 0)
    cfg, sd = wavtok_checkpoints("tiny", 0)
    W = O.cast_weights(sd)
    sig = make_input(case, GOLDEN_DIR)["sig"]
    codec.sig_to_toks(sig[:, :96].cuda())
    toks, flat = capture(codec, lambda: codec.sig_to_toks(sig.cuda()))
    check_taps(flat, ENC_TAPS, lambda t: z[f"{name}.act.{t}"], 3e-6)
    gold = z[f"{name}.toks"].astype(np.int64)
    assert np.array_equal(toks.cpu().numpy(), gold)          # min margin of these fixtures is > 4e-4
    dt = {}
    with torch.no_grad():
        orec = O.toks_to_sig(cfg, W, torch.from_numpy(gold), dt)
    rec, flat = capture(codec, lambda: codec.toks_to_sig(torch.from_numpy(gold).cuda()))
    check_taps(flat, dec_taps(cfg), lambda t: dt[t].numpy(), 5e-6)
    np.testing.assert_allclose(rec.cpu().numpy(), z[f"{name}.rec_full"], atol=1e-5)
    assert rms(rec.cpu().numpy() - orec.numpy()) < 3e-6


def test_every_module_output_full_config_production_kernels(codecs, wavtok_checkpoints):
    """FULL 40 tok/s architecture through the production kernels (non-causal rb_fused6 / rb128_fused6 / tap_gemm6 /
    lstm_persist6 in the encoder; tap_gemm6, GroupNorm, attention, dwconv+LN, polar, iSTFT GEMM in the decoder)."""
    from oracle import wavtokenizer_oracle as O

    cfg, sd = wavtok_checkpoints("full", 0)
    codec = codecs("full", 0)
    W = O.cast_weights(sd)
    sig = noise(8181, 2, 24000)
    et, dt = {}, {}
    with torch.no_grad():
        O.sig_to_feats(cfg, W, sig, et)
        otoks = O.sig_to_toks(cfg, W, sig)
        orec = O.toks_to_sig(cfg, W, otoks, dt)
    codec.sig_to_toks(sig[:, :1200].cuda())
    _, flat = capture(codec, lambda: codec.sig_to_toks(sig.cuda()), 1 << 26)
    worst = check_taps(flat, ENC_TAPS, lambda t: et[t].numpy(), 5e-6)
    rec, flat = capture(codec, lambda: codec.toks_to_sig(otoks.cuda()), 1 << 26)
    worst.update(check_taps(flat, dec_taps(cfg), lambda t: dt[t].numpy(), 1e-5))
    err = rms((rec.cpu() - orec).numpy())
    parity_record.record("wavtokenizer", "full_config_module_taps", worst_rel_err_per_tap=worst, waveform_rms_err=err)
    assert err < 2e-5


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_golden_fixture(case, wavtok_golden, codecs):
    z, meta = wavtok_golden
    name = case["name"]
    info = meta["cases"][name]
    codec = codecs(case["cfg"], case["weights_seed"])
    inp = make_input(case, GOLDEN_DIR)
    if case["kind"] == "decode":
        toks = inp["toks"].cuda()
    else:
        sig = inp["sig"].cuda()
        toks = codec.sig_to_toks(sig)
        assert toks.dtype == torch.int64 and list(toks.shape) == info["toks_shape"]
        gold = z[f"{name}.toks"].astype(np.int64)
        margin = z[f"{name}.margin64"]
        mism, bad, excused = parity_record.tokens("wavtokenizer", name, toks.cpu().numpy(), gold, margin, TAU)
        assert bad == 0, f"{bad} tokens differ outside near-ties"
        if margin.min() > TAU:
            assert np.array_equal(toks.cpu().numpy(), gold)
        assert mism <= excused
        feats = codec.sig_to_feats(sig).cpu().numpy()
        err = feats.reshape(-1)[::REC_STRIDE] - z[f"{name}.feats_strided"]
        assert rms(err) < 2e-5 and np.abs(err).max() < 3e-4, (rms(err), np.abs(err).max())
        toks = torch.from_numpy(gold).cuda()
    qf = codec.toks_to_qfeats(toks).cpu().numpy()
    np.testing.assert_array_equal(qf.reshape(-1)[::REC_STRIDE], z[f"{name}.qfeats_strided"])    # a gather: bit-exact
    rec = codec.toks_to_sig(toks).cpu().numpy()
    assert list(rec.shape) == info["rec_shape"]
    err = rec.reshape(-1)[::REC_STRIDE] - z[f"{name}.rec_strided"]
    parity_record.record("wavtokenizer", name, waveform_rms_err=rms(err))
    assert rms(err) < 1e-4, rms(err)          # the north-star bar
    assert rms(err) < 2e-5, rms(err)
    assert abs(rms(rec) - info["rec_rms"]) < 1e-4


@pytest.mark.parametrize("B,T", [(3, 36001), (33, 6000), (1, 100000)])
def test_against_oracle_on_fresh_inputs(B, T, codecs, wavtok_checkpoints):
    from oracle import wavtokenizer_oracle as O

    cfg, sd = wavtok_checkpoints("full", 0)
    codec = codecs("full", 0)
    W, W64 = O.cast_weights(sd), O.cast_weights(sd, torch.float64)
    sig = noise(3000 + B, B, T)
    with torch.no_grad():
        otoks = O.sig_to_toks(cfg, W, sig)
        _, m64 = O.sig_to_toks(cfg, W64, sig.double(), True)
        orec = O.toks_to_sig(cfg, W, otoks)
    toks = codec.sig_to_toks(sig.cuda())
    mism, bad, excused = parity_record.tokens("wavtokenizer", f"oracle_B{B}_T{T}", toks.cpu().numpy(), otoks.numpy(), m64.numpy(), TAU)
    assert bad == 0 and mism <= excused
    rec = codec.toks_to_sig(otoks.cuda()).cpu()
    e = rms((rec - orec).numpy())
    parity_record.record("wavtokenizer", f"oracle_B{B}_T{T}", waveform_rms_err=e)
    assert e < 2e-5
    assert torch.equal(codec(sig.cuda()), codec.toks_to_sig(toks))      # reconstruct mode (codec.py:45-55)


def test_rest_of_codec_api(codecs, wavtok_checkpoints):
    from oracle import wavtokenizer_oracle as O

    cfg, sd = wavtok_checkpoints("full", 0)
    codec = codecs("full", 0)
    W = O.cast_weights(sd)
    assert codec.num_codebooks == 1 and codec.vocab_size == 4096
    e = codec.embs().cpu()
    assert e.shape == (1, 4096, 512) and torch.equal(e, O.embs(W))
    sig = noise(55, 2, 6000).cuda()
    toks = codec.sig_to_toks(sig)
    assert toks.shape == (2, 10, 1)
    qf = codec.sig_to_qfeats(sig)
    assert torch.equal(qf, codec.toks_to_qfeats(toks)) and torch.equal(qf.cpu(), O.toks_to_qfeats(cfg, W, toks.cpu()))
    # feats_to_sig (wavtokenizer.py:128-135): decode arbitrary features; on the code vectors it equals toks_to_sig
    assert torch.equal(codec.feats_to_sig(qf), codec.toks_to_sig(toks))
    f = torch.randn(2, 7, 512, generator=torch.Generator().manual_seed(3)) * 0.3
    with torch.no_grad():
        o = O.feats_to_sig(cfg, W, f)
    assert rms((codec.feats_to_sig(f.cuda()).cpu() - o).numpy()) < 2e-5
    lg = codec.logits()
    assert lg.shape == (1, 4096, 4096)


def test_modes_and_errors(wavtok_checkpoints):
    from audiocodecs_amd import WavTokenizer
    from audiocodecs_amd._native import NativeError

    cfg, sd = wavtok_checkpoints("tiny", 0)
    with pytest.raises(ValueError):
        WavTokenizer(24000, mode="bogus", state_dict=sd, arch=cfg)
    both = WavTokenizer(24000, state_dict=sd, arch=cfg).eval()
    enc = WavTokenizer(24000, mode="encode", state_dict=sd, arch=cfg).eval()
    dec = WavTokenizer(24000, mode="decode", state_dict=sd, arch=cfg).eval()
    sig = noise(77, 2, 960).cuda()
    toks = both.sig_to_toks(sig)
    assert torch.equal(enc(sig), toks) and torch.equal(dec(toks), both.toks_to_sig(toks))
    with pytest.raises(NativeError, match="without decoder weights"):
        enc.toks_to_sig(toks)
    with pytest.raises(NativeError, match="without encoder weights"):
        dec.sig_to_toks(sig)
    with pytest.raises(NativeError):
        both.sig_to_toks(torch.zeros(1, 100))              # CPU tensor: no fallback
    with pytest.raises(NativeError):
        both.toks_to_sig(torch.zeros(1, 3, 2, dtype=torch.long, device="cuda"))   # K > 1
    # resampling at the Codec boundary: 16 kHz in, 24 kHz model, 16 kHz out
    c16 = WavTokenizer(16000, state_dict=sd, arch=cfg).eval()
    rec = c16(noise(78, 1, 3200).cuda())
    assert rec.shape == (1, 3200)


def test_batches_beyond_one_lstm_launch_and_nan_clip(codecs):
    """More than 64 clips: the persistent LSTM runs once per 64-clip chunk; clips are independent units, so any split of the
    batch gives the same ids and samples.  A clip with a NaN sample does not disturb the others (lstm_tail_kernel)."""
    codec = codecs("full", 0)
    sig = noise(4400, 70, 9000).cuda()
    toks = codec.sig_to_toks(sig)
    parts = torch.cat([codec.sig_to_toks(sig[i : i + 16]) for i in range(0, 70, 16)])
    assert torch.equal(toks, parts)
    rec = codec.toks_to_sig(toks)
    recp = torch.cat([codec.toks_to_sig(toks[i : i + 16]) for i in range(0, 70, 16)])
    assert torch.equal(rec, recp)
    bad = sig[:4].clone()
    bad[1, 4000] = float("nan")
    tb = codec.sig_to_toks(bad)
    for b in (0, 2, 3):
        assert torch.equal(tb[b], toks[b])
    nat = next(iter(codec._natives.values()))
    assert nat.lib.ac_lstm_status(nat.h) == 1
