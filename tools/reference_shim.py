"""Build-container-only helper: import the reference wrapper (/root/reference) offline.

Three shims: (a) `transformers` is imported first, (b) a stub
`torchaudio` whose `functional.resample` is the identity for equal rates (what torchaudio itself
does) is installed in sys.modules because audiocodecs/codec.py:22 imports torchaudio at module top,
(c) `EncodecModel.from_pretrained` is replaced by a constructor that builds the default-config
model and loads OUR seeded synthetic state dict (no network, no pretrained weights on disk).

Nothing here travels to the GPU box as a requirement; only the fixtures it helps produce do.
"""
import sys
from types import ModuleType


def load_reference_encodec(state_dict_for_cfg, hf_config_kwargs=None):
    import transformers  # noqa: F401  (must precede the torchaudio stub)
    from transformers import EncodecConfig as HFConfig
    from transformers import EncodecModel

    def _resample(w, o, n, **k):
        if o != n:
            raise NotImplementedError("torchaudio is absent offline; golden vectors use equal rates")
        return w

    ta, taf = ModuleType("torchaudio"), ModuleType("torchaudio.functional")
    taf.resample = _resample
    ta.functional = taf
    sys.modules.setdefault("torchaudio", ta)
    sys.modules.setdefault("torchaudio.functional", taf)
    if "/root/reference" not in sys.path:
        sys.path.insert(0, "/root/reference")

    def fake_from_pretrained(name, *a, **k):
        model = EncodecModel(HFConfig(**(hf_config_kwargs or {})))
        missing, unexpected = model.load_state_dict(state_dict_for_cfg, strict=True)
        return model.eval()

    EncodecModel.from_pretrained = staticmethod(fake_from_pretrained)
    from audiocodecs.encodec import Encodec

    return Encodec


def hf_mimi_config(cfg):
    """Our MimiConfig dataclass -> the third-party transformers.MimiConfig with the same fields."""
    from transformers import MimiConfig as HFConfig

    return HFConfig(
        sampling_rate=cfg.sampling_rate, num_filters=cfg.num_filters, hidden_size=cfg.hidden_size,
        upsampling_ratios=list(cfg.upsampling_ratios), kernel_size=cfg.kernel_size, last_kernel_size=cfg.last_kernel_size,
        residual_kernel_size=cfg.residual_kernel_size, compress=cfg.compress, codebook_size=cfg.codebook_size,
        codebook_dim=cfg.codebook_dim, vector_quantization_hidden_dimension=cfg.codebook_dim,
        num_quantizers=cfg.num_quantizers, num_semantic_quantizers=cfg.num_semantic_quantizers,
        num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
        num_key_value_heads=cfg.num_attention_heads, head_dim=cfg.head_dim, intermediate_size=cfg.intermediate_size,
        sliding_window=cfg.sliding_window, rope_theta=cfg.rope_theta, norm_eps=cfg.norm_eps, upsample_groups=cfg.hidden_size,
    )


def load_reference_mimi(state_dict, cfg):
    """audiocodecs.mimi.Mimi from /root/reference with `MimiModel.from_pretrained` replaced by a
    constructor that builds MimiModel(hf_mimi_config(cfg)) and loads OUR synthetic state dict."""
    import transformers  # noqa: F401
    from transformers import MimiModel

    def _resample(w, o, n, **k):
        if o != n:
            raise NotImplementedError("torchaudio is absent offline; golden vectors use equal rates")
        return w

    ta, taf = ModuleType("torchaudio"), ModuleType("torchaudio.functional")
    taf.resample = _resample
    ta.functional = taf
    sys.modules.setdefault("torchaudio", ta)
    sys.modules.setdefault("torchaudio.functional", taf)
    if "/root/reference" not in sys.path:
        sys.path.insert(0, "/root/reference")

    def fake_from_pretrained(name, *a, **k):
        model = MimiModel(hf_mimi_config(cfg))
        model.load_state_dict(state_dict, strict=True)
        return model.eval()

    MimiModel.from_pretrained = staticmethod(fake_from_pretrained)
    from audiocodecs.mimi import Mimi

    return Mimi
