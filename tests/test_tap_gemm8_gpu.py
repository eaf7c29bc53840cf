"""tap_gemm8 (csrc/tap_gemm8.h) is tap_gemm6's arithmetic in the same order behind a different load pipeline (weights through an LDS-DMA
ring, activation chunks requested two stages ahead, 8 waves): a layer's outputs must be BIT-IDENTICAL whichever kernel runs it.  The
cost model picks per layer; ac_debug_set "tap8" = 0 / 1 forces tap_gemm6 / tap_gemm8 wherever the shape allows (tile forms 1, 2, 3 by
"tap8_form").  Sizes cover clip-edge tiles of strided segments, ragged last tiles, one-tap layers in row mode (the transformers'
linear layers), seven-tap convs, and batches whose tiles do not fill the chip."""
import numpy as np
import pytest
import torch

from golden_cases import noise

pytestmark = pytest.mark.gpu


def _build(name, dac_checkpoints=None):
    import bench

    codec, cfg, sd = bench.build_codec(name)
    return codec, cfg


@pytest.mark.parametrize("name,B,T", [("encodec", 3, 36001), ("encodec", 17, 9600), ("mimi", 4, 48000), ("mimi", 1, 1920 * 3 + 7), ("wavtokenizer", 5, 24000),
                                      ("dac", 2, 20000), ("dac", 1, 700)])
def test_tap_gemm8_is_bit_identical_to_tap_gemm6(name, B, T):
    from audiocodecs_amd._native import debug_set

    codec, cfg = _build(name)
    sig = noise(7700 + T, B, T).cuda()
    with torch.no_grad():
        codec.sig_to_toks(sig[:1])               # creates the handle
        ref = None
        # (tap8_spread 2 / 0: a stage's requests dealt between its MFMA units (round 5) everywhere / all at its top; 1: per tile form)
        for tap8, form, pp in ((0, 0, 1), (1, 0, 1), (1, 1, 2), (1, 2, 2), (1, 3, 2), (-1, 0, 1), (1, 1, 0), (1, 2, 0), (1, 3, 0), (-1, 0, 0)):
            debug_set(codec, "tap8", tap8)
            debug_set(codec, "tap8_form", form)
            debug_set(codec, "tap8_spread", pp)
            names = {s[0].split("<")[0] for s in codec.profile_kernels(lambda: codec.toks_to_sig(codec.sig_to_toks(sig)))}
            if tap8 == 0:
                assert "tap_gemm8_kernel" not in names
            if tap8 == 1:
                assert "tap_gemm8_kernel" in names, names
            feats = codec.sig_to_feats(sig)
            toks = codec.sig_to_toks(sig)
            rec = codec.toks_to_sig(toks)
            if ref is None:
                ref = (feats, toks, rec)
                continue
            assert torch.equal(feats, ref[0]), (name, tap8, form, pp)
            assert torch.equal(toks, ref[1]), (name, tap8, form, pp)
            assert torch.equal(rec, ref[2]), (name, tap8, form, pp)
        debug_set(codec, "tap8", -1)
        debug_set(codec, "tap8_form", 0)
        debug_set(codec, "tap8_spread", 1)
