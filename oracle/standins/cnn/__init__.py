"""Stand-in for cnn 3.1.1 (absent): the build-defined architecture of oracle/ref_cnn.py."""
from oracle.ref_cnn import Decoder, Encoder

__all__ = ["Decoder", "Encoder"]
