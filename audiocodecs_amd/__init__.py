"""MI355X-native EnCodec (and Mimi) encode/decode path behind the `audiocodecs.Codec` API."""

from .codec import Codec
from .config import DAC_16KHZ, DAC_24KHZ, DAC_44KHZ, DAC_TINY, ENCODEC_24KHZ, MIMI_24KHZ, MIMI_TINY, TINY, DacConfig, EncodecConfig, MimiConfig
from .dac import DAC
from .encodec import Encodec
from .mimi import Mimi

__all__ = ["Codec", "Encodec", "Mimi", "DAC", "EncodecConfig", "MimiConfig", "DacConfig", "ENCODEC_24KHZ", "TINY", "MIMI_24KHZ", "MIMI_TINY",
           "DAC_44KHZ", "DAC_24KHZ", "DAC_16KHZ", "DAC_TINY"]
__version__ = "0.1.0"
