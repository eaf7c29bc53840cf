"""MI355X-native EnCodec / Mimi / DAC / WavTokenizer encode/decode paths behind the `audiocodecs.Codec` API."""

from .codec import Codec
from .config import DAC_16KHZ, DAC_24KHZ, DAC_44KHZ, DAC_TINY, ENCODEC_24KHZ, MIMI_24KHZ, MIMI_TINY, TINY, WAVTOK_40, WAVTOK_75, WAVTOK_TINY, DacConfig, EncodecConfig, MimiConfig, WavTokenizerConfig
from .dac import DAC
from .encodec import Encodec
from .mimi import Mimi
from .wavtokenizer import WavTokenizer

__all__ = ["Codec", "Encodec", "Mimi", "DAC", "WavTokenizer", "WavTokenizerConfig", "WAVTOK_40", "WAVTOK_75", "WAVTOK_TINY", "EncodecConfig", "MimiConfig", "DacConfig", "ENCODEC_24KHZ", "TINY", "MIMI_24KHZ", "MIMI_TINY",
           "DAC_44KHZ", "DAC_24KHZ", "DAC_16KHZ", "DAC_TINY"]
__version__ = "0.1.0"
