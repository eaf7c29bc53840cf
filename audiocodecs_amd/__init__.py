"""MI355X-native EnCodec encode/decode path behind the `audiocodecs.Codec` API."""

from .codec import Codec
from .config import ENCODEC_24KHZ, TINY, EncodecConfig
from .encodec import Encodec

__all__ = ["Codec", "Encodec", "EncodecConfig", "ENCODEC_24KHZ", "TINY"]
__version__ = "0.1.0"
