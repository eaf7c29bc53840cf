"""fold_weight_norm accepts both spellings of weight-norm (CPU; ADVICE r1: the Hub's EnCodec file may carry the old one)."""
import torch

from audiocodecs_amd import checkpoint


def test_old_style_weight_g_weight_v_are_folded_like_parametrizations():
    g = torch.rand(6, 1, 1) + 0.5
    v = torch.randn(6, 4, 3)
    new = checkpoint.fold_weight_norm({"enc.conv.parametrizations.weight.original0": g, "enc.conv.parametrizations.weight.original1": v,
                                       "enc.conv.bias": torch.zeros(6)})
    old = checkpoint.fold_weight_norm({"enc.conv.weight_g": g, "enc.conv.weight_v": v, "enc.conv.bias": torch.zeros(6)})
    assert set(old) == {"enc.conv.weight", "enc.conv.bias"} == set(new)
    assert torch.equal(old["enc.conv.weight"], new["enc.conv.weight"])
    ref = v * (g / v.flatten(1).norm(dim=1).view(-1, 1, 1))
    assert torch.allclose(old["enc.conv.weight"], ref, atol=1e-6)
    # a lone weight_g (no partner) is passed through untouched
    lone = checkpoint.fold_weight_norm({"x.weight_g": g})
    assert set(lone) == {"x.weight_g"}
