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

__all__ = ["conv_specs", "synthetic_state_dict", "fold_weight_norm", "mimi_conv_specs", "synthetic_mimi_state_dict", "mimi_codebook", "dac_conv_specs", "dac_snake_specs", "synthetic_dac_state_dict", "wavtok_encoder_specs", "wavtok_lstm_prefix", "synthetic_wavtok_state_dict"]

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
        elif k.endswith(".weight_g") and k[: -len("_g")] + "_v" in sd:
            # old-style torch.nn.utils.weight_norm naming -- what checkpoints saved before the parametrization API
            # (and the Hub's model.safetensors, which transformers renames at load time) carry
            prefix = k[: -len(".weight_g")]
            g = v.detach().to(torch.float32).cpu()
            vv = sd[prefix + ".weight_v"].detach().to(torch.float32).cpu()
            out[prefix + ".weight"] = torch._weight_norm(vv, g, 0).contiguous()
        elif ".parametrizations.weight.original1" in k or (k.endswith(".weight_v") and k[: -len("_v")] + "_g" in sd):
            continue
        else:
            out[k] = v.detach().cpu().contiguous()
    return out


# ---------------------------------------------------------------------------------------------
# Mimi: key/shape layout of the third-party HF ``MimiModel.state_dict()`` the
# reference wrapper loads (/root/reference/audiocodecs/mimi.py:45).  No weight-norm in this model.
# ---------------------------------------------------------------------------------------------
MIMI_CODEBOOK_S0 = 0.20
MIMI_CODEBOOK_RHO = 0.97


def mimi_conv_specs(cfg) -> List[Tuple[str, str, int, int, int, int, bool]]:
    """(key prefix, kind, Cin, Cout, kernel, stride, has_bias) for every conv of the Mimi SEANet
    encoder/decoder plus the stride-2 down/up-sample pair ([HF] mimi/modeling_mimi.py:462-484,
    :934-955, :1194-1216).  ResBlocks have no shortcut conv (use_conv_shortcut=False)."""
    F, H = cfg.num_filters, cfg.hidden_size
    specs: List[Tuple[str, str, int, int, int, int, bool]] = []

    def resblock(prefix: str, dim: int):
        hid = dim // cfg.compress
        specs.append((f"{prefix}.block.1.conv", "conv", dim, hid, cfg.residual_kernel_size, 1, True))
        specs.append((f"{prefix}.block.3.conv", "conv", hid, dim, 1, 1, True))

    specs.append(("encoder.layers.0.conv", "conv", 1, F, cfg.kernel_size, 1, True))
    i, c = 1, F
    for r in reversed(cfg.upsampling_ratios):
        resblock(f"encoder.layers.{i}", c)
        specs.append((f"encoder.layers.{i + 2}.conv", "conv", c, 2 * c, 2 * r, r, True))
        i, c = i + 3, 2 * c
    specs.append((f"encoder.layers.{i + 1}.conv", "conv", c, H, cfg.last_kernel_size, 1, True))
    rs = cfg.resample_stride
    specs.append(("downsample.conv", "conv", H, H, 2 * rs, rs, False))
    specs.append(("upsample.conv", "convtr_dw", H, H, 2 * rs, rs, False))
    specs.append(("decoder.layers.0.conv", "conv", H, c, cfg.kernel_size, 1, True))
    i = 1
    for r in cfg.upsampling_ratios:
        specs.append((f"decoder.layers.{i + 1}.conv", "convtr", c, c // 2, 2 * r, r, True))
        resblock(f"decoder.layers.{i + 2}", c // 2)
        i, c = i + 3, c // 2
    specs.append((f"decoder.layers.{i + 1}.conv", "conv", c, 1, cfg.last_kernel_size, 1, True))
    return specs


def synthetic_mimi_state_dict(cfg, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Seeded HF-format Mimi checkpoint (platform-exact draws from :mod:`prng`).

    convs as for EnCodec (no weight-norm); linear layers ~ N(0, 1/in); LayerNorm weight ~ U(0.8, 1.2),
    bias ~ N(0, 0.02^2); LayerScale ~ U(0.1, 0.4) (the trained model's scales are O(0.1-1), the 0.01
    init would make the transformers near-identity and untested); codebooks stored the way the
    model holds them, as ``embed_sum`` and ``cluster_usage`` with usage ~ U(0.5, 2):
    ``embed = embed_sum / clamp(cluster_usage, 1e-5)`` ([HF] :980-983)."""
    sd: Dict[str, torch.Tensor] = {}
    for prefix, kind, cin, cout, k, s, has_bias in mimi_conv_specs(cfg):
        if kind == "conv":
            shape, fan_in = (cout, cin, k), cin * k
        elif kind == "convtr":
            shape, fan_in = (cin, cout, k), cin * k / s
        else:  # depthwise transposed conv: groups == channels, weight [C, 1, k]
            shape, fan_in = (cin, 1, k), k / s
        w = prng.normal(seed, prefix + ".w", shape) * (1.0 / np.sqrt(fan_in))
        w = w * prng.uniform(seed, prefix + ".g", (shape[0], 1, 1), 0.9, 1.1)
        sd[f"{prefix}.weight"] = _f32(w)
        if has_bias:
            sd[f"{prefix}.bias"] = _f32(prng.normal(seed, prefix + ".b", (cout,)) * 0.02)
    H, I = cfg.hidden_size, cfg.intermediate_size
    A = cfg.num_attention_heads * cfg.head_dim
    for part in ("encoder_transformer", "decoder_transformer"):
        for l in range(cfg.num_hidden_layers):
            p = f"{part}.layers.{l}"
            for nm, (o, i_) in {"self_attn.q_proj": (A, H), "self_attn.k_proj": (A, H), "self_attn.v_proj": (A, H),
                                "self_attn.o_proj": (H, A), "mlp.fc1": (I, H), "mlp.fc2": (H, I)}.items():
                sd[f"{p}.{nm}.weight"] = _f32(prng.normal(seed, f"{p}.{nm}", (o, i_)) / np.sqrt(i_))
            for nm in ("input_layernorm", "post_attention_layernorm"):
                sd[f"{p}.{nm}.weight"] = _f32(prng.uniform(seed, f"{p}.{nm}.w", (H,), 0.8, 1.2))
                sd[f"{p}.{nm}.bias"] = _f32(prng.normal(seed, f"{p}.{nm}.b", (H,)) * 0.02)
            for nm in ("self_attn_layer_scale", "mlp_layer_scale"):
                sd[f"{p}.{nm}.scale"] = _f32(prng.uniform(seed, f"{p}.{nm}", (H,), 0.1, 0.4))
    Dq = cfg.codebook_dim
    nsem = cfg.num_semantic_quantizers
    for part, nq, q0 in (("semantic", nsem, 0), ("acoustic", cfg.num_quantizers - nsem, 1)):
        base = f"quantizer.{part}_residual_vector_quantizer"
        sd[f"{base}.input_proj.weight"] = _f32(prng.normal(seed, f"{base}.in", (Dq, H, 1)) / np.sqrt(H))
        sd[f"{base}.output_proj.weight"] = _f32(prng.normal(seed, f"{base}.out", (H, Dq, 1)) / np.sqrt(Dq))
        for q in range(nq):
            scale = MIMI_CODEBOOK_S0 * MIMI_CODEBOOK_RHO ** (q + q0 - (1 if part == "acoustic" else 0))
            usage = prng.uniform(seed, f"{base}.{q}.usage", (cfg.codebook_size,), 0.5, 2.0)
            e = prng.normal(seed, f"{base}.{q}.embed", (cfg.codebook_size, Dq)) * scale
            cb = f"{base}.layers.{q}.codebook"
            sd[f"{cb}.initialized"] = torch.tensor([1.0])
            sd[f"{cb}.cluster_usage"] = _f32(usage)
            sd[f"{cb}.embed_sum"] = _f32(e * usage[:, None])
    return sd


def mimi_codebook(sd: Dict[str, torch.Tensor], part: str, q: int) -> torch.Tensor:
    """embed = embed_sum / clamp(cluster_usage, 1e-5)[:, None]  ([HF] mimi :980-983), fp32."""
    cb = f"quantizer.{part}_residual_vector_quantizer.layers.{q}.codebook"
    return sd[f"{cb}.embed_sum"].float() / sd[f"{cb}.cluster_usage"].float().clamp(min=1e-5)[:, None]


# ---------------------------------------------------------------------------------------------
# DAC.  The reference's backend (descript-audio-codec 1.0.0) is not on disk; the key
# layout below is that of the same-architecture third-party HF ``DacModel.state_dict()`` (plain weights).
# ---------------------------------------------------------------------------------------------
def dac_conv_specs(cfg) -> List[Tuple[str, str, int, int, int, int, int]]:
    """(key prefix, kind, Cin, Cout, kernel, stride, dilation) for every conv ([HF] dac :175-264, :407-474)."""
    specs: List[Tuple[str, str, int, int, int, int, int]] = []

    def res_units(prefix: str, dim: int):
        for u, d in enumerate(cfg.dilations, start=1):
            specs.append((f"{prefix}.res_unit{u}.conv1", "conv", dim, dim, 7, 1, d))
            specs.append((f"{prefix}.res_unit{u}.conv2", "conv", dim, dim, 1, 1, 1))

    c = cfg.encoder_hidden_size
    specs.append(("encoder.conv1", "conv", 1, c, 7, 1, 1))
    for i, s in enumerate(cfg.downsampling_ratios):
        res_units(f"encoder.block.{i}", c)
        specs.append((f"encoder.block.{i}.conv1", "conv", c, 2 * c, 2 * s, s, 1))
        c *= 2
    specs.append(("encoder.conv2", "conv", c, cfg.hidden_size, 3, 1, 1))
    c = cfg.decoder_hidden_size
    specs.append(("decoder.conv1", "conv", cfg.hidden_size, c, 7, 1, 1))
    for i, s in enumerate(cfg.upsampling_ratios):
        specs.append((f"decoder.block.{i}.conv_t1", "convtr", c, c // 2, 2 * s, s, 1))
        res_units(f"decoder.block.{i}", c // 2)
        c //= 2
    specs.append(("decoder.conv2", "conv", c, 1, 7, 1, 1))
    return specs


def dac_snake_specs(cfg) -> List[Tuple[str, int]]:
    """(key prefix, channels) of every Snake1d ([HF] dac :86-100): alpha has shape [1, C, 1]."""
    out: List[Tuple[str, int]] = []

    def res_units(prefix: str, dim: int):
        for u in range(1, len(cfg.dilations) + 1):
            out.append((f"{prefix}.res_unit{u}.snake1", dim))
            out.append((f"{prefix}.res_unit{u}.snake2", dim))

    c = cfg.encoder_hidden_size
    for i in range(len(cfg.downsampling_ratios)):
        res_units(f"encoder.block.{i}", c)
        out.append((f"encoder.block.{i}.snake1", c))
        c *= 2
    out.append(("encoder.snake1", c))
    c = cfg.decoder_hidden_size
    for i in range(len(cfg.upsampling_ratios)):
        out.append((f"decoder.block.{i}.snake1", c))
        res_units(f"decoder.block.{i}", c // 2)
        c //= 2
    out.append(("decoder.snake1", c))
    return out


def synthetic_dac_state_dict(cfg, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Seeded checkpoint in HF DacModel layout: conv weights ~ N(0, 1/fan_in) * U(0.9, 1.1) (the k1 conv of a
    residual unit at 0.5 gain so 3 units per block do not blow the scale up), biases ~ N(0, 0.02^2), Snake
    alpha ~ U(0.5, 2), quantiser in_proj ~ N(0, 1/in), out_proj ~ in_proj^T + noise, codebook k ~ N(0, 0.85^2k)
    (L2-normalised for the search; the un-normalised vectors feed the output projection)."""
    sd: Dict[str, torch.Tensor] = {}
    for prefix, kind, cin, cout, k, s, d in dac_conv_specs(cfg):
        if kind == "conv":
            shape, fan_in = (cout, cin, k), cin * k
        else:
            shape, fan_in = (cin, cout, k), cin * k / s
        gain = 0.5 if prefix.endswith("conv2") and ".res_unit" in prefix else 1.0
        w = prng.normal(seed, prefix + ".w", shape) * (gain / np.sqrt(fan_in))
        w = w * prng.uniform(seed, prefix + ".g", (shape[0], 1, 1), 0.9, 1.1)
        sd[f"{prefix}.weight"] = _f32(w)
        sd[f"{prefix}.bias"] = _f32(prng.normal(seed, prefix + ".b", (cout,)) * 0.02)
    for prefix, c in dac_snake_specs(cfg):
        sd[f"{prefix}.alpha"] = _f32(prng.uniform(seed, prefix + ".alpha", (1, c, 1), 0.5, 2.0))
    H, D = cfg.hidden_size, cfg.codebook_dim
    for q in range(cfg.n_codebooks):
        p = f"quantizer.quantizers.{q}"
        sd[f"{p}.in_proj.weight"] = _f32(prng.normal(seed, p + ".in", (D, H, 1)) / np.sqrt(H))
        sd[f"{p}.in_proj.bias"] = _f32(prng.normal(seed, p + ".inb", (D,)) * 0.02)
        w_in = prng.normal(seed, p + ".in", (D, H, 1)) / np.sqrt(H)
        # out_proj ~ in_proj^T (W W^T ~ I_D) + noise: subtracting out_proj(q) then really removes the quantised component
        w_out = np.transpose(w_in, (1, 0, 2)) * prng.uniform(seed, p + ".outg", (1, D, 1), 0.9, 1.1)
        w_out = w_out + prng.normal(seed, p + ".out", (H, D, 1)) * (0.1 / np.sqrt(H))
        sd[f"{p}.out_proj.weight"] = _f32(w_out)
        sd[f"{p}.out_proj.bias"] = _f32(prng.normal(seed, p + ".outb", (H,)) * 0.002)
        sd[f"{p}.codebook.weight"] = _f32(prng.normal(seed, p + ".cb", (cfg.codebook_size, D)) * (1.0 * 0.85**q))
    return sd


# ---------------------------------------------------------------------------------------------
# WavTokenizer: key/shape layout of the checkpoint the reference wrapper loads through
# `wavtokenizer.WavTokenizer.from_pretrained0802` (/root/reference/audiocodecs/wavtokenizer.py:76-78) -- the
# `state_dict` of a Lightning checkpoint whose keys start with feature_extractor. / backbone. / head.  The backend
# source is NOT on disk: names restate the published module tree (encodec-style SEANetEncoder under
# feature_extractor.encodec.encoder.model.{i}, old-style weight-norm `conv.conv.weight_g/weight_v`, the codebook at
# feature_extractor.encodec.quantizer.vq.layers.0._codebook.embed, VocosBackbone with pos_net, ISTFTHead).
# ---------------------------------------------------------------------------------------------
WAVTOK_CODEBOOK_S0 = 0.35


def wavtok_encoder_specs(cfg) -> List[Tuple[str, int, int, int, int]]:
    """(key prefix, Cin, Cout, kernel, stride) for every conv of the SEANet encoder, module-list order."""
    F = cfg.num_filters
    p = "feature_extractor.encodec.encoder.model"
    specs = [(f"{p}.0", 1, F, cfg.kernel_size, 1)]
    i, c = 1, F
    for r in reversed(cfg.ratios):
        hid = c // cfg.compress
        specs.append((f"{p}.{i}.block.1", c, hid, cfg.residual_kernel_size, 1))
        specs.append((f"{p}.{i}.block.3", hid, c, 1, 1))
        specs.append((f"{p}.{i}.shortcut", c, c, 1, 1))
        specs.append((f"{p}.{i + 2}", c, 2 * c, 2 * r, r))
        i, c = i + 3, 2 * c
    specs.append((f"{p}.{i + 2}", c, cfg.dimension, cfg.last_kernel_size, 1))
    return specs


def wavtok_lstm_prefix(cfg) -> str:
    return f"feature_extractor.encodec.encoder.model.{1 + 3 * len(cfg.ratios)}.lstm"


def synthetic_wavtok_state_dict(cfg, seed: int = 0) -> Dict[str, torch.Tensor]:
    sd: Dict[str, torch.Tensor] = {}
    for prefix, cin, cout, k, s in wavtok_encoder_specs(cfg):
        w = prng.normal(seed, prefix + ".w", (cout, cin, k)) / np.sqrt(cin * k)
        w = w * prng.uniform(seed, prefix + ".g", (cout, 1, 1), 0.9, 1.1)
        w32 = _f32(w)
        sd[f"{prefix}.conv.conv.bias"] = _f32(prng.normal(seed, prefix + ".b", (cout,)) * 0.02)
        sd[f"{prefix}.conv.conv.weight_g"] = _norm_dim0(w32)
        sd[f"{prefix}.conv.conv.weight_v"] = w32
    D = cfg.lstm_dim
    lp = wavtok_lstm_prefix(cfg)
    for layer in range(cfg.num_lstm_layers):
        for nm in ("ih", "hh"):
            sd[f"{lp}.weight_{nm}_l{layer}"] = _f32(prng.normal(seed, f"{lp}.w{nm}{layer}", (4 * D, D)) / np.sqrt(D))
            sd[f"{lp}.bias_{nm}_l{layer}"] = _f32(prng.normal(seed, f"{lp}.b{nm}{layer}", (4 * D,)) * 0.02)
    q = "feature_extractor.encodec.quantizer.vq.layers.0._codebook"
    sd[f"{q}.embed"] = _f32(prng.normal(seed, q + ".embed", (cfg.codebook_size, cfg.dimension)) * WAVTOK_CODEBOOK_S0)
    sd[f"{q}.embed_avg"] = sd[f"{q}.embed"].clone()
    sd[f"{q}.cluster_size"] = torch.ones(cfg.codebook_size)
    sd[f"{q}.inited"] = torch.tensor([1.0])

    C, I = cfg.backbone_dim, cfg.intermediate_dim

    def conv(prefix, cin, cout, k, gain=1.0):
        sd[f"{prefix}.weight"] = _f32(prng.normal(seed, prefix + ".w", (cout, cin, k)) * (gain / np.sqrt(cin * k)))
        sd[f"{prefix}.bias"] = _f32(prng.normal(seed, prefix + ".b", (cout,)) * 0.02)

    def norm(prefix, n):
        sd[f"{prefix}.weight"] = _f32(prng.uniform(seed, prefix + ".w", (n,), 0.8, 1.2))
        sd[f"{prefix}.bias"] = _f32(prng.normal(seed, prefix + ".b", (n,)) * 0.05)

    def adanorm(prefix):
        sd[f"{prefix}.scale.weight"] = _f32(prng.uniform(seed, prefix + ".s", (cfg.adanorm_num_embeddings, C), 0.8, 1.2))
        sd[f"{prefix}.shift.weight"] = _f32(prng.normal(seed, prefix + ".h", (cfg.adanorm_num_embeddings, C)) * 0.05)

    conv("backbone.embed", cfg.dimension, C, 7)
    for i in (0, 1, 3, 4):     # ResnetBlocks of pos_net
        pp = f"backbone.pos_net.{i}"
        norm(f"{pp}.norm1", C)
        conv(f"{pp}.conv1", C, C, 3)
        norm(f"{pp}.norm2", C)
        conv(f"{pp}.conv2", C, C, 3, gain=0.5)
    pa = "backbone.pos_net.2"  # AttnBlock
    norm(f"{pa}.norm", C)
    for nm in ("q", "k", "v"):
        conv(f"{pa}.{nm}", C, C, 1, gain=2.0 if nm != "v" else 1.0)   # q, k gain: scores spread enough for a non-flat softmax
    conv(f"{pa}.proj_out", C, C, 1, gain=0.5)
    norm("backbone.pos_net.5", C)
    adanorm("backbone.norm")
    for l in range(cfg.num_layers):
        pl = f"backbone.convnext.{l}"
        sd[f"{pl}.dwconv.weight"] = _f32(prng.normal(seed, pl + ".dw.w", (C, 1, 7)) / np.sqrt(7.0))
        sd[f"{pl}.dwconv.bias"] = _f32(prng.normal(seed, pl + ".dw.b", (C,)) * 0.02)
        adanorm(f"{pl}.norm")
        sd[f"{pl}.pwconv1.weight"] = _f32(prng.normal(seed, pl + ".p1.w", (I, C)) / np.sqrt(C))
        sd[f"{pl}.pwconv1.bias"] = _f32(prng.normal(seed, pl + ".p1.b", (I,)) * 0.02)
        sd[f"{pl}.pwconv2.weight"] = _f32(prng.normal(seed, pl + ".p2.w", (C, I)) / np.sqrt(I))
        sd[f"{pl}.pwconv2.bias"] = _f32(prng.normal(seed, pl + ".p2.b", (C,)) * 0.02)
        sd[f"{pl}.gamma"] = _f32(prng.uniform(seed, pl + ".gamma", (C,), 0.05, 0.3))
    norm("backbone.final_layer_norm", C)
    sd["head.out.weight"] = _f32(prng.normal(seed, "head.out.w", (cfg.n_fft + 2, C)) * (0.7 / np.sqrt(C)))
    sd["head.out.bias"] = _f32(prng.normal(seed, "head.out.b", (cfg.n_fft + 2,)) * 0.02)
    # head.istft.window: torch.hann_window(n_fft) (periodic) -- a registered buffer of the upstream module
    n = np.arange(cfg.n_fft, dtype=np.float64)
    sd["head.istft.window"] = _f32(0.5 - 0.5 * np.cos(2.0 * np.pi * n / cfg.n_fft))
    return sd
