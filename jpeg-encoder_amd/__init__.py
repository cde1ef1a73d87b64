"""MI355X-native JPEG encode path, drop-in for uriGrif/jpeg-encoder's pipeline.

The directory name carries a hyphen; import it as `jpeg_encoder_amd` through the
loader module of that name at the repository root.
"""
from . import _native, host, stripes
from ._native import DCT_BIN, DCT_REAL, DCT_REAL_STRICT, JpgError
from .jpeg_image import Encoder, JpegImage, get_downsampling_factor

__all__ = ["Encoder", "JpegImage", "JpgError", "get_downsampling_factor", "DCT_REAL", "DCT_BIN",
           "DCT_REAL_STRICT", "host", "stripes", "_native"]
