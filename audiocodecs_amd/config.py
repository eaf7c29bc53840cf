"""Architecture description of the EnCodec family handled by the HIP path.

Field names and defaults follow the third-party ``transformers.EncodecConfig`` that the reference
wrapper instantiates through ``EncodecModel.from_pretrained("facebook/encodec_24khz")``
(/root/reference/audiocodecs/encodec.py:49-51).  Only the causal,
weight-normed, mono, un-chunked (24 kHz) variant is supported -- the one BASELINE.json names.
"""

from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Tuple

__all__ = ["EncodecConfig", "TINY", "ENCODEC_24KHZ"]


@dataclass(frozen=True)
class EncodecConfig:
    sampling_rate: int = 24000
    num_filters: int = 32
    hidden_size: int = 128  # latent width == codebook_dim
    upsampling_ratios: Tuple[int, ...] = (8, 5, 4, 2)
    kernel_size: int = 7
    last_kernel_size: int = 7
    residual_kernel_size: int = 3
    compress: int = 2
    num_lstm_layers: int = 2
    codebook_size: int = 1024
    num_quantizers: int = 32
    target_bandwidths: Tuple[float, ...] = (1.5, 3.0, 6.0, 12.0, 24.0)

    @property
    def hop_length(self) -> int:
        return int(math.prod(self.upsampling_ratios))

    @property
    def frame_rate(self) -> int:
        return math.ceil(self.sampling_rate / self.hop_length)

    @property
    def lstm_dim(self) -> int:
        return self.num_filters * 2 ** len(self.upsampling_ratios)

    def num_frames(self, num_samples: int) -> int:
        """Frames produced for `num_samples` input samples (ceil at every strided conv)."""
        n = num_samples
        for r in reversed(self.upsampling_ratios):
            n = -(-n // r)
        return n

    def num_quantizers_for_bandwidth(self, bandwidth: float) -> int:
        bw_per_q = math.log2(self.codebook_size) * self.frame_rate
        return int(max(1, math.floor(bandwidth * 1000 / bw_per_q)))


ENCODEC_24KHZ = EncodecConfig()
# Same topology, 234 942 parameters: small enough that every activation fits a fixture
# (SURVEY.md Appendix A.5).  codebook_size stays 1024 so the wrapper's bandwidth mapping holds.
TINY = EncodecConfig(num_filters=4, hidden_size=16)
