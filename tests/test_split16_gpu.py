"""split16 arithmetic (csrc/split16.h: two scaled fp16 planes per operand, 3 partial products -- the default fp32-fidelity
arithmetic of the GEMM-shaped kernels): what fp16 lacks is RANGE, so these cases push the per-clip / per-row / per-channel
power-of-two scales: input gains from 1e-3 to 50, a loud burst inside a quiet clip, all-zero and denormal-small clips.  Checked against the CPU oracle with the usual policy
(tokens exact outside fp64 near-ties, waveform within 1e-5 of the signal's scale) and for independence of batch neighbours."""
import numpy as np
import pytest
import torch

import parity_record
from conftest import GOLDEN_DIR
from golden_cases import noise
from test_oracle_golden import TAU, tokens_match_up_to_ties

pytestmark = pytest.mark.gpu


def rms(a):
    return float(np.sqrt(np.mean(np.asarray(a.detach().cpu().numpy(), dtype=np.float64) ** 2)))


@pytest.fixture(scope="module")
def enc(checkpoints):
    from audiocodecs_amd import Encodec
    from oracle import encodec_oracle as O

    cfg, sd = checkpoints("full", 0)
    codec = Encodec(24000, num_codebooks=8, state_dict=sd).eval()
    return cfg, sd, codec, O.fold_weight_norm(sd), O.fold_weight_norm(sd, torch.float64)


def _against_oracle(enc, sig, name):
    from oracle import encodec_oracle as O

    cfg, sd, codec, W, W64 = enc
    toks = codec.sig_to_toks(sig.cuda())
    rec = codec.toks_to_sig(toks)
    torch.set_num_threads(min(16, max(1, torch.get_num_threads())))
    with torch.no_grad():
        otoks = O.sig_to_toks(cfg, W, sig)
        _, m64 = O.sig_to_toks(cfg, W64, sig.double(), None, 8, True)
        orec = O.toks_to_sig(cfg, W, toks.cpu())
    diff, bad, excused = parity_record.tokens("encodec", f"split16_{name}", toks.cpu().numpy(), otoks.numpy(), m64.numpy(), TAU)
    err = rms(rec.cpu() - orec)
    parity_record.record("encodec", f"split16_{name}", waveform_rms_err=err)
    assert bad == 0, f"{bad} tokens differ outside fp64 near-ties"
    assert diff <= excused
    assert bool(torch.isfinite(rec).all())
    assert err < 1e-5 * max(1.0, rms(orec)), (err, rms(orec))
    return toks, rec


@pytest.mark.parametrize("gain", [1e-3, 1.0, 50.0])
def test_input_gain(enc, gain):
    sig = noise(7100, 2, 24000) * gain
    _against_oracle(enc, sig, f"gain_{gain:g}")


def test_loud_burst_in_a_quiet_clip_and_its_neighbours(enc):
    cfg, sd, codec, W, W64 = enc
    sig = noise(7101, 3, 24000) * 1e-3
    sig[1, 9000:9400] += noise(7102, 1, 400)[0] * 30.0          # 90 dB above the rest of the clip
    toks, rec = _against_oracle(enc, sig, "burst")
    # the quiet neighbours do not see the burst: a clip's scales are its own
    for b in (0, 2):
        tb = codec.sig_to_toks(sig[b : b + 1].cuda())
        assert torch.equal(tb, toks[b : b + 1])
        assert torch.equal(codec.toks_to_sig(tb), rec[b : b + 1])
    assert torch.equal(codec.sig_to_toks(sig[1:2].cuda()), toks[1:2])


def test_zero_and_vanishing_clips(enc):
    sig = torch.zeros(3, 16000)
    sig[1] = noise(7103, 1, 16000)[0] * 1e-30
    sig[2] = noise(7104, 1, 16000)[0]
    _against_oracle(enc, sig, "zero_tiny")


def test_saturated_lstm(checkpoints):
    """Gate biases of +30 drive i, g, o to 1.0 and c up by one per step: h becomes EXACTLY 1.0 after a few steps.  The exchange of
    the persistent LSTM tells "arrived" from "not yet written" by bit 14 of every published fp16 term, so h must travel in a
    form whose bit 14 is clear at 1.0 -- an early version published 2 h and would have timed out here."""
    from audiocodecs_amd import Encodec
    from oracle import encodec_oracle as O

    cfg, sd = checkpoints("full", 0)
    sd = {k: v.clone() for k, v in sd.items()}
    D = 512
    n = 0
    for k in sd:
        if ".lstm.bias_ih_l" in k:
            sd[k][:] = 30.0              # i, f, g, o all saturated: c[t] = c[t-1] + 1, h = 1.0 exactly once tanh(c) rounds to 1
            n += 1
    assert n == 4
    codec = Encodec(24000, num_codebooks=8, state_dict=sd).eval()
    sig = noise(7106, 2, 16000)
    W, W64 = O.fold_weight_norm(sd), O.fold_weight_norm(sd, torch.float64)
    _against_oracle((cfg, sd, codec, W, W64), sig, "saturated_lstm")
    nat = next(iter(codec._natives.values()))
    assert nat.lib.ac_lstm_status(nat.h) >= 0, "persistent LSTM reported a failed launch"


@pytest.mark.parametrize("name", ["dac", "mimi", "wavtokenizer"])
def test_a_clip_alone_equals_the_clip_in_a_batch(name, dac_checkpoints, mimi_checkpoints, wavtok_checkpoints):
    """Scales are per clip / per row, and row mode is never inferred from the batch shape: encoding and decoding one clip
    gives bit-identical results whether it runs alone or between neighbours (a one-clip conv once took the row-mode path
    that the same conv in a batch does not take)."""
    from audiocodecs_amd import DAC, Mimi, WavTokenizer

    if name == "dac":
        cfg, sd = dac_checkpoints("full", 0)
        codec = DAC(44100, 44100, num_codebooks=9, state_dict=sd, config=cfg).eval()
        sig = noise(7110, 3, 30000)
    elif name == "mimi":
        cfg, sd = mimi_checkpoints("full", 0)
        codec = Mimi(24000, num_codebooks=8, state_dict=sd, config=cfg).eval()
        sig = noise(7111, 3, 36000)
    else:
        cfg, sd = wavtok_checkpoints("full", 0)
        codec = WavTokenizer(24000, state_dict=sd, arch=cfg).eval()
        sig = noise(7112, 3, 30000)
    sig[1] *= 40.0                                   # a loud neighbour in the middle
    sig = sig.cuda()
    toks = codec.sig_to_toks(sig)
    rec = codec.toks_to_sig(toks)
    for b in range(3):
        tb = codec.sig_to_toks(sig[b : b + 1])
        assert torch.equal(tb, toks[b : b + 1]), f"{name}: tokens of clip {b} depend on its batch"
        assert torch.equal(codec.toks_to_sig(tb), rec[b : b + 1]), f"{name}: decode of clip {b} depends on its batch"


def _speech_like_batch():
    """Segments of the reference's example.wav (16 kHz speech fed as 24 kHz samples: the spectrum shifts, the dynamics stay) at
    levels from -40 dBFS to full scale, with what real recordings add: a DC offset, isolated full-scale clicks, a hard-clipped
    segment, digital silence in the middle of a clip, and a clip that fades in from nothing."""
    from golden_cases import read_example_wav

    wav = read_example_wav(GOLDEN_DIR)[0]
    T = 36000
    starts = [20000, 60000, 100000, 140000, 180000, 210000]
    clips = [wav[s : s + T].clone() for s in starts]
    peak = [float(c.abs().max()) for c in clips]
    clips[0] = clips[0] / peak[0]                                 # 0 dBFS
    clips[1] = clips[1] / peak[1] * 10 ** (-40 / 20)              # -40 dBFS
    clips[2] = clips[2] / peak[2] * 0.1 + 0.05                    # -20 dBFS riding on a DC offset
    clips[3] = clips[3] / peak[3] * 0.02
    clips[3][5000] = 1.0                                          # clicks 34 dB above the speech around them
    clips[3][5001] = -1.0
    clips[3][23456] = 0.9
    clips[4] = (clips[4] / peak[4] * 4.0).clamp(-1.0, 1.0)        # hard clipping
    clips[4][12000:20000] = 0.0                                   # digital silence inside the clip
    clips[5] = clips[5] / peak[5] * torch.linspace(0.0, 1.0, T) ** 4
    return torch.stack(clips)


def test_speech_like_batch_tokens_and_bound_waste(enc):
    """(a) the usual parity policy on data with the dynamics of real recordings (tokens exact outside fp64 near-ties, waveform
    within 1e-5 of the signal's scale); (b) every split16 scale that comes from a BOUND instead of a measured amax (the tensors
    that exist only inside a fused kernel) wastes less than 2^10 of range: bound / true amax, the true amax taken from the
    oracle's module outputs per clip.  split16 keeps fp32-grade relative precision 16 bits below a tensor's largest element
    (split16.h), so 10 wasted bits still leave every element down to 2^-6 of the largest exact to fp32 rounding and the rest
    with an absolute error of 2^-30 of the largest."""
    import ctypes as C
    import math

    import torch.nn.functional as F
    from oracle import encodec_oracle as O

    cfg, sd, codec, W, W64 = enc
    sig = _speech_like_batch()
    toks, rec = _against_oracle(enc, sig, "speech_like")
    nat = next(iter(codec._natives.values()))
    buf = (C.c_float * 64)()
    n = nat.lib.ac_debug_bounds(nat.h, buf, 64)
    assert n == 11 + 4 * 8
    b = [float(v) for v in buf[:n]]
    assert b[1] > 0 and b[8] > 0, "the fused chains are what runs on this handle"
    taps, dtaps = {}, {}
    with torch.no_grad():
        O.masked_embeddings(cfg, W, sig, None, taps=taps)
        O.toks_to_sig(cfg, W, toks.cpu(), taps=dtaps)

        def amax(t):                     # per clip
            return t.abs().flatten(1).max(dim=1).values.double()

        def hidden(x, p):                # ELU(conv_k3(ELU(x))) of the residual block with prefix p
            return F.elu(O.conv1d_causal(F.elu(x), W[p + ".block.1.conv.weight"], W[p + ".block.1.conv.bias"]))

        waste = {}
        # encoder front: bounds chained from amax(sig)
        a_sig = amax(sig)
        X0 = b[0] + b[1] * a_sig
        H = b[2] + b[3] * X0
        Y1 = b[4] + b[5] * H + b[6] * X0
        waste["enc_front.x0"] = X0 / amax(taps["enc0"])
        waste["enc_front.hidden"] = H / amax(hidden(taps["enc0"], "encoder.layers.1"))
        waste["enc_front.y1"] = Y1 / amax(taps["enc1"])
        # separate residual blocks: hidden bound from the exact amax of the block input
        for i, (tin, layer) in enumerate((("enc3", 4), ("enc6", 7), ("enc9", 10)), start=1):
            if b[12 + 2 * i] > 0:
                waste[f"enc_rb{i}.hidden"] = (b[11 + 2 * i] + b[12 + 2 * i] * amax(taps[tin])) / amax(hidden(taps[tin], f"encoder.layers.{layer}"))
        for i, (tin, layer) in enumerate((("dec3", 4), ("dec6", 7), ("dec9", 10))):
            if b[12 + 16 + 2 * i] > 0:
                waste[f"dec_rb{i}.hidden"] = (b[11 + 16 + 2 * i] + b[12 + 16 + 2 * i] * amax(dtaps[tin])) / amax(hidden(dtaps[tin], f"decoder.layers.{layer}"))
        # decoder tail: bounds chained from the exact amax of ELU(block output)
        xe = F.elu(dtaps["dec10"])
        U = b[7] + b[8] * amax(xe)
        waste["dec_tail.u"] = U / amax(dtaps["dec12"])
        waste["dec_tail.hidden"] = (b[9] + b[10] * U) / amax(hidden(dtaps["dec12"], "decoder.layers.13"))
        # LSTM output: |lstm(x) + x| <= 1 + amax(x)
        waste["enc_lstm.out"] = (1.0 + amax(taps["enc12"])) / amax(taps["enc13"])
        waste["dec_lstm.out"] = (1.0 + amax(dtaps["dec0"])) / amax(dtaps["dec1"])
    worst = {k: float(v.max()) for k, v in waste.items()}
    parity_record.record("encodec", "split16_speech_like", bound_waste_bits={k: round(math.log2(max(v, 1e-30)), 2) for k, v in worst.items()})
    for k, v in waste.items():
        assert bool((v >= 1.0 - 1e-6).all()), (k, v)              # a bound below the true maximum would overflow the fp16 planes
        assert worst[k] < 2.0 ** 10, (k, math.log2(worst[k]))


def test_trained_like_weight_statistics(checkpoints):
    """Round-3 verdict, "missing" #4: every parity figure is on seeded synthetic weights whose rows all look alike, and no trained
    checkpoint exists offline.  What trained weights have that the synthetic ones lack is SPREAD: weight-norm gains that differ by
    orders of magnitude between output channels, rows dominated by a few large taps, recurrent matrices large enough to saturate
    gates.  split16 scales weights per OUTPUT ROW (2^-s per row) and activations per clip, so exactly this spread is what its range
    argument has to survive: per-channel gains 2^N(0, 2) (each layer's RMS gain kept, so that the embeddings stay at the codebooks'
    scale and the fp64 near-tie margin keeps its meaning: 0.75 % of the frames are near-ties, as with the plain weights), 0.5 %
    of the direction weights x 25, recurrent weights x 3.  Usual policy against the CPU oracle on the same perturbed weights."""
    from audiocodecs_amd import Encodec
    from oracle import encodec_oracle as O

    cfg, sd = checkpoints("full", 0)
    sd = {k: v.clone() for k, v in sd.items()}
    g = torch.Generator().manual_seed(20261002)
    spread = []
    for k in sorted(sd):
        if k.endswith("parametrizations.weight.original0"):
            e = torch.randn(sd[k].shape, generator=g) * 2.0
            e = e - 0.5 * torch.log2(torch.exp2(2 * e).mean())     # the layer's RMS gain unchanged: embeddings stay where the codebooks are
            sd[k] = sd[k] * torch.exp2(e)
            spread.append(float(e.max() - e.min()))
        elif k.endswith("parametrizations.weight.original1"):
            m = torch.rand(sd[k].shape, generator=g) < 0.005
            sd[k] = torch.where(m, sd[k] * 25.0, sd[k])
        elif ".lstm.weight_hh_l" in k:
            sd[k] = sd[k] * 3.0
    assert len(spread) > 20 and max(spread) > 8.0                   # some layer's gains span more than 2^8
    codec = Encodec(24000, num_codebooks=8, state_dict=sd).eval()
    W, W64 = O.fold_weight_norm(sd), O.fold_weight_norm(sd, torch.float64)
    sig = noise(7110, 2, 24000)
    sig[1] *= 0.05
    _against_oracle((cfg, sd, codec, W, W64), sig, "trained_like_weights")
    nat = next(iter(codec._natives.values()))
    assert nat.lib.ac_lstm_status(nat.h) >= 0
