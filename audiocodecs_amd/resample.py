"""Sample-rate conversion at the Codec boundary.

The reference calls ``torchaudio.functional.resample(sig, orig, new)`` with torchaudio's defaults
(/root/reference/audiocodecs/codec.py:59-63,95-99): ``sinc_interp_hann``, lowpass_filter_width 6,
rolloff 0.99.  torchaudio is not on disk here (pinned 2.4.0 in downstream/environment.yml:244), so
the filter bank below restates its published algorithm -- **parity with
torchaudio is unpinned**; tests check it against an fp64 restatement and against the analytic
response on band-limited tones.  Equal rates return the input unchanged, as torchaudio does.
The FIR itself runs in the HIP library (``ac_resample``); there is no CPU fallback.
"""

from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Tuple

import torch

__all__ = ["resample", "sinc_kernel"]

_LOWPASS_FILTER_WIDTH = 6
_ROLLOFF = 0.99


def sinc_kernel(orig_freq: int, new_freq: int, dtype=torch.float32) -> Tuple[torch.Tensor, int, int, int]:
    """Hann-windowed sinc filter bank [n, taps] for orig -> new (after gcd reduction) and (n, o, width).
    Computed in `dtype` like torchaudio computes it in the waveform's dtype."""
    g = math.gcd(int(orig_freq), int(new_freq))
    o, n = int(orig_freq) // g, int(new_freq) // g
    base_freq = min(o, n) * _ROLLOFF
    width = math.ceil(_LOWPASS_FILTER_WIDTH * o / base_freq)
    idx = torch.arange(-width, width + o, dtype=dtype)[None] / o            # [1, taps]
    t = torch.arange(0, -n, -1, dtype=dtype)[:, None] / n + idx             # phase i: -i/n + idx
    t = (t * base_freq).clamp_(-_LOWPASS_FILTER_WIDTH, _LOWPASS_FILTER_WIDTH)
    window = torch.cos(t * math.pi / _LOWPASS_FILTER_WIDTH / 2) ** 2
    t = t * math.pi
    scale = base_freq / o
    kernels = torch.where(t == 0, torch.ones_like(t), t.sin() / t) * window * scale
    return kernels.contiguous(), n, o, width


_BANKS: Dict[tuple, tuple] = {}


def resample(sig: torch.Tensor, orig_freq, new_freq) -> torch.Tensor:
    """sig [B, L] -> [B, ceil(new/orig * L)] on the GPU (identity when the rates are equal)."""
    if int(orig_freq) == int(new_freq):
        return sig
    from . import _native

    if sig.shape[0] == 0:   # an empty shard: the output shape only (the library is not called)
        g = math.gcd(int(orig_freq), int(new_freq))
        return torch.empty(0, int(math.ceil((int(new_freq) // g) * sig.shape[1] / (int(orig_freq) // g))), dtype=torch.float32, device=sig.device)
    if not sig.is_cuda:
        raise _native.NativeError("audiocodecs_amd.resample runs on MI355X only: move the signal to a cuda device")
    key = (int(orig_freq), int(new_freq), sig.device.index)
    if key not in _BANKS:
        k, n, o, width = sinc_kernel(orig_freq, new_freq)
        _BANKS[key] = (k.to(sig.device), n, o, width)
    kern, n, o, width = _BANKS[key]
    x = sig.to(torch.float32).contiguous()
    B, L = x.shape
    L_out = int(math.ceil(n * L / o))
    y = torch.empty(B, L_out, dtype=torch.float32, device=sig.device)
    with torch.cuda.device(sig.device):
        rc = _native.lib().ac_resample(
            C.c_void_p(x.data_ptr()), B, L, C.c_void_p(kern.data_ptr()), n, o, kern.shape[1], width,
            C.c_void_p(y.data_ptr()), L_out, C.c_void_p(torch.cuda.current_stream().cuda_stream),
        )
    _native.check(rc, None, "ac_resample")
    return y
