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
