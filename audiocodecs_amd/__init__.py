"""MI355X-native EnCodec (and Mimi) encode/decode path behind the `audiocodecs.Codec` API."""

from .codec import Codec
from .config import ENCODEC_24KHZ, MIMI_24KHZ, MIMI_TINY, TINY, EncodecConfig, MimiConfig
from .encodec import Encodec
from .mimi import Mimi

__all__ = ["Codec", "Encodec", "Mimi", "EncodecConfig", "MimiConfig", "ENCODEC_24KHZ", "TINY", "MIMI_24KHZ", "MIMI_TINY"]
__version__ = "0.1.0"
