"""Checkpoints for the EnCodec path: synthetic (seeded) ones and weight-norm folding.

There are no pretrained weights offline, so parity and benchmarks run on
seeded synthetic checkpoints in the exact key/shape layout of the third-party HF
``EncodecModel.state_dict()`` the reference wrapper loads
(/root/reference/audiocodecs/encodec.py:51; key list in SURVEY.md Appendix A.3).  A real
``facebook/encodec_24khz`` state dict drops into :func:`fold_weight_norm` unchanged.

Synthetic recipe (made platform-exact):
  * effective conv weight  w ~ N(0, gain^2 / fan_in) * U(0.9, 1.1) per index of dim 0,
    fan_in = Cin*k (ConvTranspose: Cin*k/stride); biases ~ N(0, 0.02^2);
  * stored as  original1 = w,  original0 = ||w||  (norm over dims 1,2 per index of dim 0), so the
    weight-norm parametrisation  g * v / ||v||  reproduces w bit-for-bit wherever it is evaluated;
  * LSTM weights ~ N(0, 1/in_features), LSTM biases ~ N(0, 0.02^2);
  * codebook k ~ N(0, (s0 * rho^k)^2): a geometric residual-scale schedule straight from the PRNG
    (no encoder pass needed, so the GPU box rebuilds identical codebooks without any oracle).
All draws come from :mod:`audiocodecs_amd.prng`.
"""

from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np
import torch

from . import prng
from .config import EncodecConfig

__all__ = ["conv_specs", "synthetic_state_dict", "fold_weight_norm"]

CODEBOOK_S0 = 0.10
CODEBOOK_RHO = 0.94


def conv_specs(cfg: EncodecConfig) -> List[Tuple[str, str, int, int, int, int]]:
    """(key prefix, kind, Cin, Cout, kernel, stride) for every conv of encoder and decoder.

    Layer indices follow the HF module lists (EncodecEncoder/EncodecDecoder; ELU layers occupy an
    index of their own): SURVEY.md Appendix A.1 / A.2.
    """
    F, H = cfg.num_filters, cfg.hidden_size
    specs: List[Tuple[str, str, int, int, int, int]] = []

    def resblock(prefix: str, dim: int):
        hid = dim // cfg.compress
        specs.append((f"{prefix}.block.1.conv", "conv", dim, hid, cfg.residual_kernel_size, 1))
        specs.append((f"{prefix}.block.3.conv", "conv", hid, dim, 1, 1))
        specs.append((f"{prefix}.shortcut.conv", "conv", dim, dim, 1, 1))

    # encoder
    specs.append(("encoder.layers.0.conv", "conv", 1, F, cfg.kernel_size, 1))
    i, c = 1, F
    for r in reversed(cfg.upsampling_ratios):
        resblock(f"encoder.layers.{i}", c)
        specs.append((f"encoder.layers.{i + 2}.conv", "conv", c, 2 * c, 2 * r, r))
        i, c = i + 3, 2 * c
    # i -> LSTM, i+1 -> ELU, i+2 -> final conv
    specs.append((f"encoder.layers.{i + 2}.conv", "conv", c, H, cfg.last_kernel_size, 1))
    # decoder
    specs.append(("decoder.layers.0.conv", "conv", H, c, cfg.kernel_size, 1))
    i = 2  # 1 -> LSTM
    for r in cfg.upsampling_ratios:
        specs.append((f"decoder.layers.{i + 1}.conv", "convtr", c, c // 2, 2 * r, r))
        resblock(f"decoder.layers.{i + 2}", c // 2)
        i, c = i + 3, c // 2
    specs.append((f"decoder.layers.{i + 1}.conv", "conv", c, 1, cfg.last_kernel_size, 1))
    return specs


def lstm_prefixes(cfg: EncodecConfig) -> Tuple[str, str]:
    n = len(cfg.upsampling_ratios)
    return f"encoder.layers.{1 + 3 * n}.lstm", "decoder.layers.1.lstm"


def _f32(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a.astype(np.float32)))


def _norm_dim0(w: torch.Tensor) -> torch.Tensor:
    """||w|| per index of dim 0, from the same ATen kernel `torch._weight_norm` divides by, so that
    g / ||v|| == 1.0 exactly and the parametrised weight equals `w` bit-for-bit (checked in tests)."""
    try:
        return torch._weight_norm_interface(w, torch.ones(w.shape[0], 1, 1), 0)[1].contiguous()
    except Exception:  # pragma: no cover - older/newer torch without the fused CPU kernel
        return torch.norm_except_dim(w, 2, 0)


def synthetic_state_dict(cfg: EncodecConfig, seed: int = 0, gain: float = 1.0) -> Dict[str, torch.Tensor]:
    sd: Dict[str, torch.Tensor] = {}
    for prefix, kind, cin, cout, k, s in conv_specs(cfg):
        if kind == "conv":
            shape, fan_in = (cout, cin, k), cin * k
        else:  # ConvTranspose1d weight is [Cin, Cout, k]; weight-norm dim 0 is Cin there
            shape, fan_in = (cin, cout, k), cin * k / s
        w = prng.normal(seed, prefix + ".w", shape) * (gain / np.sqrt(fan_in))
        w = w * prng.uniform(seed, prefix + ".g", (shape[0], 1, 1), 0.9, 1.1)
        w32 = _f32(w)
        sd[f"{prefix}.bias"] = _f32(prng.normal(seed, prefix + ".b", (cout,)) * 0.02)
        sd[f"{prefix}.parametrizations.weight.original0"] = _norm_dim0(w32)
        sd[f"{prefix}.parametrizations.weight.original1"] = w32
    D = cfg.lstm_dim
    for prefix in lstm_prefixes(cfg):
        for layer in range(cfg.num_lstm_layers):
            for nm in ("ih", "hh"):
                sd[f"{prefix}.weight_{nm}_l{layer}"] = _f32(
                    prng.normal(seed, f"{prefix}.w{nm}{layer}", (4 * D, D)) / np.sqrt(D)
                )
                sd[f"{prefix}.bias_{nm}_l{layer}"] = _f32(
                    prng.normal(seed, f"{prefix}.b{nm}{layer}", (4 * D,)) * 0.02
                )
    for q in range(cfg.num_quantizers):
        scale = CODEBOOK_S0 * CODEBOOK_RHO**q
        e = prng.normal(seed, f"quantizer.layers.{q}.embed", (cfg.codebook_size, cfg.hidden_size)) * scale
        p = f"quantizer.layers.{q}.codebook"
        sd[f"{p}.embed"] = _f32(e)
        sd[f"{p}.embed_avg"] = sd[f"{p}.embed"].clone()
        sd[f"{p}.cluster_size"] = torch.ones(cfg.codebook_size)
        sd[f"{p}.inited"] = torch.tensor([1.0])
    return sd


def fold_weight_norm(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """HF-format state dict -> plain ``<prefix>.weight`` tensors (fp32, host).

    ``w = g * v / ||v||`` over dims (1, 2) per index of dim 0 -- exactly what
    ``torch.nn.utils.parametrizations.weight_norm`` evaluates on every forward of the third-party
    model.  Folded once here with the same torch primitive so the HIP path
    and the reference see identical effective weights.
    """
    out: Dict[str, torch.Tensor] = {}
    suffix0 = ".parametrizations.weight.original0"
    for k, v in sd.items():
        if k.endswith(suffix0):
            prefix = k[: -len(suffix0)]
            g = v.detach().to(torch.float32).cpu()
            vv = sd[prefix + ".parametrizations.weight.original1"].detach().to(torch.float32).cpu()
            out[prefix + ".weight"] = torch._weight_norm(vv, g, 0).contiguous()
        elif ".parametrizations.weight.original1" in k:
            continue
        else:
            out[k] = v.detach().cpu().contiguous()
    return out
