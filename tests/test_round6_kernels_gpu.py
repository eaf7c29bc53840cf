"""Round-6 kernels (csrc/stream_path.hip): the thin stages at sixteen waves per CU.  The EnCodec chains (enc_stream / dec_stream) are
covered where their round-3 forms were (tests/test_fused_chains_gpu.py, the tap tests of tests/test_gpu_parity.py); here
  * rb_stream6 against rb_fused6<64> on the EnCodec path (ac_debug_set "rb_stream"): two fp32-faithful evaluations of one function;
  * Mimi's first encoder block with the stem folded in (rb_stream6m.h STEM) against stem_kernel + the block, and its last decoder block with
    the head folded in (HEAD: also tests/test_round4_kernels_gpu.py) at lengths around the 16-row tile and the stream-segment seams;
  * Mimi's 128-channel identity blocks without a slab (rb_stream128m.h: the k3 conv's row offsets by DPP row shifts + a 1 KB halo);
  * a stream's result must not depend on its batch neighbours (segments are cut by batch size)."""
import pytest
import torch

from golden_cases import noise

pytestmark = pytest.mark.gpu


def _kernels(codec, fn):
    return {s[0].split("<")[0] for s in codec.profile_kernels(fn)}


def _mimi():
    from audiocodecs_amd import Mimi, checkpoint
    from audiocodecs_amd.config import MIMI_24KHZ

    sd = checkpoint.synthetic_mimi_state_dict(MIMI_24KHZ, seed=0)
    return Mimi(24000, num_codebooks=8, state_dict=sd, config=MIMI_24KHZ).eval()


@pytest.mark.parametrize("B,T", [(1, 1920), (2, 1920 * 3 + 13), (3, 1920 * 7 + 1), (5, 1920 * 2 + 959)])
def test_mimi_stem_and_head_folds_match_the_separate_kernels(B, T):
    from audiocodecs_amd._native import debug_set

    codec = _mimi()
    sig = noise(6100 + T % 97, B, T).cuda()
    with torch.no_grad():
        names = _kernels(codec, lambda: codec.toks_to_sig(codec.sig_to_toks(sig)))
        assert "rb_stream6m_kernel" in names and "stem_kernel" not in names and "rb_fused6_head_kernel" not in names, names
        assert "rb_stream128m_kernel" in names and "rb128_fused6_kernel" not in names, names      # the slab-less 128-channel block (rb_stream128m.h)
        feats, toks = codec.sig_to_feats(sig), codec.sig_to_toks(sig)
        wav = codec.toks_to_sig(toks)
        assert torch.equal(codec.sig_to_toks(sig), toks) and torch.equal(codec.toks_to_sig(toks), wav)      # reruns are bit-equal
        one = codec.sig_to_feats(sig[:1].contiguous())                                                     # ... and do not depend on the batch
        assert torch.equal(one, feats[:1])
        debug_set(codec, "rb_stream", 0)
        names = _kernels(codec, lambda: codec.toks_to_sig(codec.sig_to_toks(sig)))
        assert "stem_kernel" in names and not any(n.startswith("rb_stream6") for n in names), names
        f0, t0 = codec.sig_to_feats(sig), codec.sig_to_toks(sig)
        w0 = codec.toks_to_sig(toks)
        debug_set(codec, "rb_stream", 1)
    scale = float(f0.abs().max())
    assert float((feats - f0).abs().max()) < 5e-5 * max(1.0, scale), (float((feats - f0).abs().max()), scale)
    assert float((toks == t0).float().mean()) > 0.99
    assert float((wav - w0).abs().max()) <= 2e-5 * max(float(w0.abs().max()), 1e-3)


@pytest.mark.parametrize("T", [320, 641, 4800, 24001])
def test_rb_stream6_matches_rb_fused6_on_the_encodec_path(checkpoints, T):
    from audiocodecs_amd import Encodec
    from audiocodecs_amd._native import debug_set

    cfg, sd = checkpoints("full", 0)
    codec = Encodec(24000, num_codebooks=8, state_dict=sd, config=cfg).eval()
    sig = noise(6200 + T % 89, 3, T).cuda()
    with torch.no_grad():
        names = _kernels(codec, lambda: codec.toks_to_sig(codec.sig_to_toks(sig)))
        assert "rb_stream6_kernel" in names and "rb_fused6_kernel" not in names, names
        assert "rb_stream128m_kernel" in names and "rb128_fused6_kernel" not in names, names      # the 128-channel block, 1x1-shortcut form
        f1, t1 = codec.sig_to_feats(sig), codec.sig_to_toks(sig)
        w1 = codec.toks_to_sig(t1)
        debug_set(codec, "rb_stream", 0)
        names = _kernels(codec, lambda: codec.toks_to_sig(codec.sig_to_toks(sig)))
        assert "rb_fused6_kernel" in names and "rb_stream6_kernel" not in names and "rb128_fused6_kernel" in names, names
        f0, t0 = codec.sig_to_feats(sig), codec.sig_to_toks(sig)
        w0 = codec.toks_to_sig(t1)
        debug_set(codec, "rb_stream", 1)
    assert float((f1 - f0).abs().max()) < 2e-5 * max(1.0, float(f0.abs().max()))
    assert float((t1 == t0).float().mean()) > 0.995
    assert float((w1 - w0).abs().max()) <= 2e-5 * max(float(w0.abs().max()), 1e-3)
