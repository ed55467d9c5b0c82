"""daspeech_amd — MI355X-native (gfx950) hot path of DASpeech behind the reference's operator surface.

`daspeech_amd.custom_ops` mirrors `DASpeech/custom_ops/__init__.py:1` (same 8 names, same signatures).
"""
__version__ = "0.1.0"
