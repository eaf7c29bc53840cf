"""Round-4 kernels that REPLACE an older kernel with a different summation order -- attention16_kernel (csrc/mimi.h: Mimi attention
in split16 arithmetic) and head4_kernel (csrc/thin.h: four lanes per output sample) -- against the kernels they replace, which
stay selectable (ac_debug_set "attn_exact" / "head_seq").  The oracle parity of the default path is tests/test_mimi_gpu_parity.py /
test_dac_gpu_parity.py; here the two implementations of one layer are compared with each other over the sizes where they differ in
structure: one transformer frame, the 64-query / 64-key tile edges, more frames than the sliding window (250), ragged batches."""
import pytest
import torch

from golden_cases import noise

pytestmark = pytest.mark.gpu


def _codec(name):
    import bench

    return bench.build_codec(name)[0]


def _kernels(codec, fn):
    return {s[0].split("<")[0] for s in codec.profile_kernels(fn)}


# Mimi: 1920 samples per 12.5 Hz frame, the transformers run at 25 Hz: T25 = 2 * frames.  32 / 33 frames = 64 / 66 positions (tile edge),
# 130 frames = 260 positions (> window 250: the first key tile of the last query tile is partly masked), 163 frames = 326 positions.
@pytest.mark.parametrize("B,frames", [(1, 1), (2, 32), (3, 33), (2, 130), (1, 163)])
def test_attention16_matches_the_fp32_mfma_attention(B, frames):
    from audiocodecs_amd._native import debug_set

    codec = _codec("mimi")
    sig = noise(9100 + frames, B, 1920 * frames).cuda()
    with torch.no_grad():
        codec.sig_to_toks(sig[:1])
        debug_set(codec, "attn_exact", 1)
        assert "attention16_kernel" not in _kernels(codec, lambda: codec.sig_to_feats(sig))
        f_ref = codec.sig_to_feats(sig)
        r_ref = codec.toks_to_sig(codec.sig_to_toks(sig))
        debug_set(codec, "attn_exact", 0)
        assert "attention16_kernel" in _kernels(codec, lambda: codec.sig_to_feats(sig))
        f_new = codec.sig_to_feats(sig)
        t_new = codec.sig_to_toks(sig)
        r_new = codec.toks_to_sig(t_new)
        again = codec.sig_to_feats(sig)
    assert torch.equal(f_new, again)                                   # deterministic (no atomics in the data path: the amax words are maxima)
    scale = float(f_ref.abs().max())
    assert float((f_new - f_ref).abs().max()) <= 2e-5 * scale, (float((f_new - f_ref).abs().max()), scale)
    assert torch.isfinite(r_new).all()
    # the waveform goes through a token decision: compare where the tokens agree (they do outside near-ties)
    debug_set(codec, "attn_exact", 1)
    with torch.no_grad():
        t_ref = codec.sig_to_toks(sig)
    debug_set(codec, "attn_exact", 0)
    agree = float((t_ref == t_new).float().mean())
    assert agree >= 0.97, agree
    if agree == 1.0:
        assert float((r_new - r_ref).abs().max()) <= 2e-4 * max(float(r_ref.abs().max()), 1e-3)


@pytest.mark.parametrize("name,B,T", [("mimi", 2, 1920 * 3 + 5), ("mimi", 1, 1920), ("dac", 2, 5000), ("dac", 1, 700)])
def test_head4_matches_the_sequential_head(name, B, T):
    from audiocodecs_amd._native import debug_set

    codec = _codec(name)
    sig = noise(9300 + T, B, T).cuda()
    with torch.no_grad():
        toks = codec.sig_to_toks(sig)
        if name == "mimi": debug_set(codec, "mimi_tail", 0)        # (by default Mimi's final conv is folded into its last block: rb_fused6_head_kernel)
        debug_set(codec, "head_seq", 1)
        assert "head4_kernel" not in _kernels(codec, lambda: codec.toks_to_sig(toks))
        ref = codec.toks_to_sig(toks)
        debug_set(codec, "head_seq", 0)
        assert "head4_kernel" in _kernels(codec, lambda: codec.toks_to_sig(toks))
        new = codec.toks_to_sig(toks)
        assert torch.equal(new, codec.toks_to_sig(toks))
    assert new.shape == ref.shape
    # one fp32 dot product of 7 F (3 F) terms per sample in two summation orders
    assert float((new - ref).abs().max()) <= 4e-6 * max(float(ref.abs().max()), 1.0), float((new - ref).abs().max())


@pytest.mark.parametrize("B,T", [(2, 9000), (1, 700), (3, 4099)])
def test_dac_unit6_matches_the_two_launch_residual_unit(B, T, dac_checkpoints):
    """dac_unit6_kernel (csrc/dac_unit6.h: k7 conv -> Snake -> 1 x 1 conv + residual of the 96-channel units as one kernel, the hidden
    activation scaled per wave tile instead of per clip) against the two tap-GEMM launches it replaces (ac_debug_set "dac_unit" 0): same
    tokens (the units are in the decoder), waveform equal to fp32 rounding; clip edges inside the dilated halo (T = 700)."""
    from audiocodecs_amd import DAC
    from audiocodecs_amd._native import debug_set

    cfg, sd = dac_checkpoints("full", 0)
    codec = DAC(44100, 44100, num_codebooks=9, state_dict=sd, config=cfg).eval()
    sig = noise(9500 + T, B, T).cuda()
    with torch.no_grad():
        toks = codec.sig_to_toks(sig)
        debug_set(codec, "dac_unit", 0)
        assert not any(n.startswith("dac_unit6") for n in _kernels(codec, lambda: codec.toks_to_sig(toks)))
        ref = codec.toks_to_sig(toks)
        debug_set(codec, "dac_unit", 1)
        names = _kernels(codec, lambda: codec.toks_to_sig(toks))
        assert "dac_unit6_kernel" in names, names
        new = codec.toks_to_sig(toks)
        assert torch.equal(new, codec.toks_to_sig(toks))
        assert torch.equal(codec.sig_to_toks(sig), toks)
    assert float((new - ref).abs().max()) <= 2e-5 * max(float(ref.abs().max()), 1e-3), float((new - ref).abs().max())


@pytest.mark.parametrize("B,frames", [(2, 3), (1, 1), (3, 7)])
def test_mimi_tail_with_the_final_conv_folded_in_is_bit_identical(B, frames):
    """rb_fused6_head_kernel (csrc/rb_fused6.h HEAD: Mimi's last residual block with the decoder's final Conv1d(64, 1, 3) applied to the
    block's output tile in LDS, tiles advancing by 62 rows) against rb_fused6<64, false> + head4_kernel (ac_debug_set "mimi_tail" 0): the
    block's arithmetic is unchanged and the conv's is head4_kernel's in the same order -- the waveform must be BIT-identical, tile seams
    (every 62 samples), the clip's first samples (zero left context) and ragged ends included."""
    from audiocodecs_amd._native import debug_set

    codec = _codec("mimi")
    sig = noise(9700 + frames, B, 1920 * frames + 13).cuda()
    close = lambda a, r: float((a - r).abs().max()) <= 2e-5 * max(float(r.abs().max()), 1e-3)
    with torch.no_grad():
        toks = codec.sig_to_toks(sig)
        # the round-4 pair, bit-identical: rb_fused6<64, false> + head4_kernel against rb_fused6_head_kernel
        debug_set(codec, "rb_stream", 0)
        debug_set(codec, "mimi_tail", 0)
        names = _kernels(codec, lambda: codec.toks_to_sig(toks))
        assert "rb_fused6_kernel" in names and "head4_kernel" in names and not any(n.startswith("rb_stream6") for n in names), names
        ref = codec.toks_to_sig(toks)
        debug_set(codec, "mimi_tail", 1)
        names = _kernels(codec, lambda: codec.toks_to_sig(toks))
        assert "rb_fused6_head_kernel" in names and "head4_kernel" not in names, names
        new = codec.toks_to_sig(toks)
        assert new.shape == ref.shape
        assert torch.equal(new, ref), float((new - ref).abs().max())
        # round 6: the stream kernels (the same arithmetic with every MFMA's 32 products in the lanes' load order, the head on the matrix
        # pipe: fp32-faithful, not bit-equal) -- the block alone, and with the head folded in (rb_stream6m.h), tile seams every 16 samples
        debug_set(codec, "rb_stream", 1)
        debug_set(codec, "mimi_tail", 0)
        names = _kernels(codec, lambda: codec.toks_to_sig(toks))
        assert "rb_stream6_kernel" in names and "head4_kernel" in names, names
        assert close(codec.toks_to_sig(toks), ref)
        debug_set(codec, "mimi_tail", 1)
        names = _kernels(codec, lambda: codec.toks_to_sig(toks))
        assert "rb_stream6m_kernel" in names and "head4_kernel" not in names and "rb_fused6_head_kernel" not in names, names
        fold = codec.toks_to_sig(toks)
        assert close(fold, ref), float((fold - ref).abs().max())
        assert torch.equal(fold, codec.toks_to_sig(toks))
