"""The fused thin-channel chains of the EnCodec path (csrc/enc_front.h: stem -> ResBlock(32) -> ELU -> Conv1d(32,64,k4,s2);
csrc/dec_tail.h: ConvTranspose1d(64,32,k4,s2) -> ResBlock(32) -> ELU -> Conv1d(32,1,k7)) against
  (a) the same layers as separate kernels (AC_FUSE=0: stem / rb_fused6<32> / thin_conv6 / head, the round-2 path), and
  (b) the CPU oracle's module taps ([HF] modeling_encodec.py:290-301, :330-341 as called from audiocodecs/encodec.py:90,139),
at the lengths where the chains' edge rules matter: odd T (one reflected step on the right of the strided conv), T around the
32-sample chunk and the stream-segment boundaries, ragged `length` masks, and one-clip-vs-batch bit equality (a stream's result
must not depend on how the clip was cut into segments or on its batch neighbours)."""
import ctypes as C

import numpy as np
import pytest
import torch

from golden_cases import noise

pytestmark = pytest.mark.gpu


def rms(a):
    return float(np.sqrt(np.mean(np.asarray(a, dtype=np.float64) ** 2)))


@pytest.fixture(scope="module")
def pair(checkpoints):
    """(fused, separate): two handles on the same weights; the second is finalized under AC_FUSE=0."""
    import os

    from audiocodecs_amd import Encodec

    cfg, sd = checkpoints("full", 0)
    fused = Encodec(24000, num_codebooks=8, state_dict=sd, config=cfg).eval()
    fused.sig_to_feats(noise(1, 1, 640).cuda())
    old = os.environ.get("AC_FUSE")
    os.environ["AC_FUSE"] = "0"
    try:
        sep = Encodec(24000, num_codebooks=8, state_dict=sd, config=cfg).eval()
        sep.sig_to_feats(noise(1, 1, 640).cuda())   # the handle is created (and finalized) on first use
    finally:
        if old is None:
            del os.environ["AC_FUSE"]
        else:
            os.environ["AC_FUSE"] = old
    return fused, sep, cfg, sd


def kernel_names(codec, fn):
    return {s[0] for s in codec.profile_kernels(fn)}


def test_the_fused_kernels_are_what_runs(pair):
    fused, sep, cfg, sd = pair
    sig = noise(31, 2, 4800).cuda()
    names = kernel_names(fused, lambda: fused.toks_to_sig(fused.sig_to_toks(sig)))
    assert ("enc_stream_kernel" in names or "enc_front_kernel" in names) and ("dec_stream_kernel" in names or "dec_tail_kernel" in names), names
    assert not any(n.startswith(("stem_kernel", "head_kernel", "thin_conv6", "rb_fused6_kernel<32")) for n in names), names
    names0 = kernel_names(sep, lambda: sep.toks_to_sig(sep.sig_to_toks(sig)))
    assert not ({"enc_front_kernel", "dec_tail_kernel", "enc_stream_kernel", "dec_stream_kernel"} & names0) and "stem_kernel" in names0


# lengths: below / at / above one chunk (32), odd lengths, the 64-sample threshold of the fused path, a stream-segment boundary
# (8 chunks = 256 samples at small batch), hop multiples and non-multiples
LENGTHS = [64, 65, 95, 96, 97, 127, 255, 256, 257, 289, 319, 320, 321, 641, 1023, 2049, 4800, 24001]


@pytest.mark.parametrize("T", LENGTHS)
def test_encoder_front_matches_separate_kernels(pair, T):
    fused, sep, cfg, sd = pair
    sig = noise(500 + T, 3, T).cuda()
    a, b = fused.sig_to_feats(sig), sep.sig_to_feats(sig)
    scale = float(b.abs().max())
    assert float((a - b).abs().max()) < 2e-5 * max(1.0, scale), (T, float((a - b).abs().max()), scale)
    ta, tb = fused.sig_to_toks(sig), sep.sig_to_toks(sig)
    assert float((ta == tb).float().mean()) > 0.995     # two fp32-faithful evaluations: only fp32-level near-ties may differ


@pytest.mark.parametrize("N", [1, 2, 3, 7, 15, 16, 17, 75])
def test_decoder_tail_matches_separate_kernels(pair, N):
    fused, sep, cfg, sd = pair
    g = torch.Generator().manual_seed(900 + N)
    toks = torch.randint(0, 1024, (3, N, 8), generator=g).cuda()
    a, b = fused.toks_to_sig(toks), sep.toks_to_sig(toks)
    assert a.shape == b.shape == (3, 320 * N)
    assert rms((a - b).cpu().numpy()) < 2e-6 * max(1.0, rms(b.cpu().numpy())), N
    assert float((a - b).abs().max()) < 2e-5 * max(1.0, float(b.abs().max()))


def test_ragged_length_mask_inside_the_fused_front(pair):
    """audiocodecs/encodec.py:84-92: samples at t >= T * length are zeroed before the encoder."""
    fused, sep, cfg, sd = pair
    T = 4803
    sig = noise(77, 4, T).cuda()
    length = torch.tensor([1.0, 0.7, 0.31, 0.003], device="cuda")
    a, b = fused.sig_to_toks(sig, length), sep.sig_to_toks(sig, length)
    assert float((a == b).float().mean()) > 0.995
    masked = sig.clone()
    for i, l in enumerate(length.tolist()):
        masked[i, int(np.ceil(np.float32(T) * np.float32(l))):] = 0     # t >= T * length (fp32 product, as the kernel compares)
    assert torch.equal(fused.sig_to_toks(masked), a)


def test_a_clip_does_not_depend_on_its_batch_or_segmentation(pair, monkeypatch):
    """Streams are cut per clip from the batch size (enc_front_fwd / dec_tail_fwd: seg_chunks ~ 6144 streams): one clip alone,
    the same clip among 40 others, and explicit segment lengths all give the same bits."""
    fused, sep, cfg, sd = pair
    sig = noise(4141, 41, 9600).cuda()
    feats = fused.sig_to_feats(sig)
    solo = fused.sig_to_feats(sig[17:18])
    assert torch.equal(solo[0], feats[17])
    toks = fused.sig_to_toks(sig)
    rec = fused.toks_to_sig(toks)
    assert torch.equal(fused.toks_to_sig(toks[17:18])[0], rec[17])
    from audiocodecs_amd._native import debug_set

    try:
        for seg in (1, 3, 1000):     # seg 1 at 20 clips: 6000 one-chunk streams, every SIMD holds two waves
            debug_set(fused, "front_seg", seg)
            debug_set(fused, "tail_seg", seg)
            for _ in range(2):
                assert torch.equal(fused.sig_to_feats(sig[:20]), feats[:20]), seg
                assert torch.equal(fused.toks_to_sig(toks[:20]), rec[:20]), seg
    finally:
        debug_set(fused, "front_seg", 0)
        debug_set(fused, "tail_seg", 0)


def test_module_taps_inside_the_chains(pair):
    """The fused kernels write the module outputs they never store otherwise (stem, ResBlock(32), transposed conv, ResBlock(32))
    while the capture hook is armed: each against the oracle's tap at an odd length."""
    from oracle import encodec_oracle as O
    from test_gpu_parity import FULL_DEC_TAPS, FULL_ENC_TAPS, capture

    fused, sep, cfg, sd = pair
    W = O.fold_weight_norm(sd)
    sig = noise(8181, 2, 3333)
    taps, dtaps = {}, {}
    with torch.no_grad():
        O.masked_embeddings(cfg, W, sig, None, taps=taps)
        otoks = O.sig_to_toks(cfg, W, sig)
        O.toks_to_sig(cfg, W, otoks, taps=dtaps)
    _, flat = capture(fused, lambda: fused.sig_to_toks(sig.cuda()), 1 << 25)
    off = 0
    for tap in FULL_ENC_TAPS[:3]:          # enc0 stem, enc1 ResBlock(32), enc3 strided conv
        g = taps[tap].numpy()
        got = flat[off : off + g.size].reshape(g.shape[0], g.shape[2], g.shape[1]).transpose(0, 2, 1)
        np.testing.assert_allclose(got, g, atol=5e-6 * max(1.0, float(np.abs(g).max())), rtol=2e-5, err_msg=tap)
        off += g.size
    _, flat = capture(fused, lambda: fused.toks_to_sig(otoks.cuda()), 1 << 25)
    sizes = [dtaps[t].numpy().size for t in FULL_DEC_TAPS]
    off = sum(sizes[:-2])
    for tap in FULL_DEC_TAPS[-2:]:         # dec12 transposed conv, dec13 ResBlock(32)
        g = dtaps[tap].numpy()
        got = flat[off : off + g.size].reshape(g.shape[0], g.shape[2], g.shape[1]).transpose(0, 2, 1)
        np.testing.assert_allclose(got, g, atol=5e-6 * max(1.0, float(np.abs(g).max())), rtol=2e-5, err_msg=tap)
        off += g.size
    assert off == flat.size


def test_nan_sample_stays_inside_its_receptive_field(pair):
    """A NaN sample poisons only the frames whose receptive field holds it (enc_front's bounds come from the FINITE samples'
    amax, split16.h); the other clips of the batch are bit-identical to a clean run."""
    fused, sep, cfg, sd = pair
    sig = noise(9292, 3, 9600).cuda()
    clean = fused.sig_to_feats(sig)
    bad = sig.clone()
    bad[1, 5000] = float("nan")
    f = fused.sig_to_feats(bad)
    assert torch.equal(f[0], clean[0]) and torch.equal(f[2], clean[2])
    frame = 5000 // 320
    assert torch.equal(f[1, : frame - 1], clean[1, : frame - 1])
    assert bool(torch.isnan(f[1, frame + 1 :]).all())


@pytest.mark.parametrize("B,T", [(1, 4000), (5, 9600), (20, 9600), (64, 24000)])
def test_fused_chains_reruns_are_bit_equal_at_small_sizes(pair, B, T):
    """FLOAT outputs of the fused chains repeat bit for bit also at sizes where the stream segmentation differs from the full batch's
    (few streams per CU, one wave per SIMD up to two): the packed-FMA fault of round 3 (profiles/r3_pk_fma_hazard.md) changed rows
    from run to run and no token-level test saw it."""
    fused, sep, cfg, sd = pair
    sig = noise(9300 + B, B, T).cuda()
    feats = fused.sig_to_feats(sig)
    toks = fused.sig_to_toks(sig)
    rec = fused.toks_to_sig(toks)
    for _ in range(5):
        assert torch.equal(fused.sig_to_feats(sig), feats)
        assert torch.equal(fused.toks_to_sig(toks), rec)
