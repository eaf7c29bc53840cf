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

__all__ = ["EncodecConfig", "TINY", "ENCODEC_24KHZ", "MimiConfig", "MIMI_24KHZ", "MIMI_TINY", "DacConfig", "DAC_44KHZ", "DAC_24KHZ", "DAC_16KHZ", "DAC_TINY", "WavTokenizerConfig", "WAVTOK_40", "WAVTOK_75", "WAVTOK_TINY"]


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


@dataclass(frozen=True)
class MimiConfig:
    """Fields of the third-party ``transformers.MimiConfig`` the Mimi path depends on (defaults =
    ``kyutai/mimi``, what /root/reference/audiocodecs/mimi.py:45 loads).
    Causal convs with constant (zero) padding, identity ResBlock shortcuts, no weight-norm, MHA
    (num_key_value_heads == num_attention_heads), RoPE "default", exact-erf GELU."""

    sampling_rate: int = 24000
    num_filters: int = 64
    hidden_size: int = 512
    upsampling_ratios: Tuple[int, ...] = (8, 6, 5, 4)
    kernel_size: int = 7
    last_kernel_size: int = 3
    residual_kernel_size: int = 3
    compress: int = 2
    codebook_size: int = 2048
    codebook_dim: int = 256  # == vector_quantization_hidden_dimension
    num_quantizers: int = 32
    num_semantic_quantizers: int = 1
    num_hidden_layers: int = 8
    num_attention_heads: int = 8
    head_dim: int = 64
    intermediate_size: int = 2048
    sliding_window: int = 250
    rope_theta: float = 10000.0
    norm_eps: float = 1e-5
    resample_stride: int = 2  # encodec_frame_rate / frame_rate: the stride-2 down/up-sample pair

    @property
    def hop_length(self) -> int:
        return int(math.prod(self.upsampling_ratios)) * self.resample_stride

    @property
    def frame_rate(self) -> float:
        return self.sampling_rate / self.hop_length

    @property
    def seanet_dim(self) -> int:
        return self.num_filters * 2 ** len(self.upsampling_ratios)

    def num_frames(self, num_samples: int) -> int:
        """ceil at every strided conv, then at the stride-2 down-sampler ([HF] mimi :1264-1275)."""
        n = num_samples
        for r in reversed(self.upsampling_ratios):
            n = -(-n // r)
        return -(-n // self.resample_stride)


MIMI_24KHZ = MimiConfig()
# Same topology at 1/8 width and 2 transformer layers: every activation fits a fixture.
MIMI_TINY = MimiConfig(
    num_filters=8, hidden_size=64, codebook_dim=32, num_hidden_layers=2, num_attention_heads=4, head_dim=16,
    intermediate_size=128, sliding_window=6,
)


@dataclass(frozen=True)
class DacConfig:
    """Architecture of the Descript Audio Codec as the reference wrapper uses it
    (/root/reference/audiocodecs/dac.py:28-60 loads `dac.DAC` of descript-audio-codec 1.0.0 -- NOT on
    disk).  Field names follow the same-architecture third-party ``transformers.DacConfig``
    (SURVEY.md Appendix D); defaults = the 44.1 kHz model of BASELINE.json configs[2]."""

    sampling_rate: int = 44100
    encoder_hidden_size: int = 64
    downsampling_ratios: Tuple[int, ...] = (2, 4, 8, 8)
    decoder_hidden_size: int = 1536
    upsampling_ratios: Tuple[int, ...] = (8, 8, 4, 2)
    n_codebooks: int = 9
    codebook_size: int = 1024
    codebook_dim: int = 8
    dilations: Tuple[int, ...] = (1, 3, 9)  # residual units per block ([HF] dac :218-220)

    @property
    def hidden_size(self) -> int:  # latent width
        return self.encoder_hidden_size * 2 ** len(self.downsampling_ratios)

    @property
    def hop_length(self) -> int:
        return int(math.prod(self.downsampling_ratios))

    def num_frames(self, num_samples: int) -> int:
        """Encoder output length: symmetric padding, so each strided conv (k = 2s, pad = ceil(s/2)) gives
        floor((L + 2*ceil(s/2) - 2s) / s) + 1; the stride-1 convs keep the length."""
        n = num_samples
        for s in self.downsampling_ratios:
            n = (n + 2 * math.ceil(s / 2) - 2 * s) // s + 1
        return n

    def num_samples(self, num_frames: int) -> int:
        """Decoder output length: each transposed conv (k = 2s, pad = ceil(s/2)) gives (L-1)*s - 2*pad + 2s."""
        n = num_frames
        for s in self.upsampling_ratios:
            n = (n - 1) * s - 2 * math.ceil(s / 2) + 2 * s
        return n


DAC_44KHZ = DacConfig()
DAC_24KHZ = DacConfig(sampling_rate=24000, downsampling_ratios=(2, 4, 5, 8), upsampling_ratios=(8, 5, 4, 2), n_codebooks=32)
DAC_16KHZ = DacConfig(sampling_rate=16000, downsampling_ratios=(2, 4, 5, 8), upsampling_ratios=(8, 5, 4, 2), n_codebooks=12)
# 1/8 width, odd stride included, every activation fits a fixture
DAC_TINY = DacConfig(encoder_hidden_size=8, decoder_hidden_size=64, downsampling_ratios=(2, 4, 5, 8),
                     upsampling_ratios=(8, 5, 4, 2), n_codebooks=4)


@dataclass(frozen=True)
class WavTokenizerConfig:
    """Architecture of WavTokenizer as the reference wrapper uses it (/root/reference/audiocodecs/wavtokenizer.py:31-135
    loads `wavtokenizer.WavTokenizer.from_pretrained0802(config, checkpoint)` of lucadellalib/WavTokenizer -- NOT on
    disk; field values restate the published YAML configs the wrapper names, wavtokenizer.py:37-40: PARITY UNPINNED).
    EnCodec-style SEANet encoder (non-causal, reflect padding, weight-norm, 2-layer LSTM) -> ONE Euclidean codebook
    4096 x 512 -> Vocos backbone (k7 embed conv, pos_net = 2 ResnetBlocks + single-head attention + 2 ResnetBlocks +
    GroupNorm, AdaLayerNorm, ConvNeXt blocks, LayerNorm) -> iSTFT head ("same" padding)."""

    sampling_rate: int = 24000
    num_filters: int = 32
    dimension: int = 512               # encoder output width == codebook dim == backbone input_channels
    ratios: Tuple[int, ...] = (6, 5, 5, 4)   # `dowmsamples` of the YAML; the encoder applies them reversed
    kernel_size: int = 7
    last_kernel_size: int = 7
    residual_kernel_size: int = 3
    compress: int = 2
    num_lstm_layers: int = 2
    codebook_size: int = 4096
    backbone_dim: int = 768
    intermediate_dim: int = 2304
    num_layers: int = 12
    adanorm_num_embeddings: int = 4
    num_groups: int = 32               # GroupNorm groups of pos_net
    n_fft: int = 2400
    bandwidth_id: int = 0              # wavtokenizer.py:94,116: always 0

    @property
    def hop_length(self) -> int:
        return int(math.prod(self.ratios))

    @property
    def hidden_size(self) -> int:
        return self.dimension

    @property
    def lstm_dim(self) -> int:
        return self.num_filters * 2 ** len(self.ratios)

    def num_frames(self, num_samples: int) -> int:
        """ceil at every strided conv (extra right padding completes the last frame)."""
        n = num_samples
        for r in reversed(self.ratios):
            n = -(-n // r)
        return n


WAVTOK_40 = WavTokenizerConfig()                                         # ...frame40_3s_nq1_code4096_dim512_kmeans200_attn.yaml
WAVTOK_75 = WavTokenizerConfig(ratios=(8, 5, 4, 2), n_fft=1280)          # ...frame75_3s_nq1_code4096_dim512_kmeans200_attn.yaml
# 1/8 width, hop 48: every activation fits a fixture
WAVTOK_TINY = WavTokenizerConfig(num_filters=4, dimension=32, ratios=(4, 3, 2, 2), codebook_size=128, backbone_dim=256,
                                 intermediate_dim=512, num_layers=2, num_groups=32, n_fft=192)
