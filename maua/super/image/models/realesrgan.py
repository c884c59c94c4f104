"""Drop-in name for maua/super/image/models/realesrgan.py:22-49: re-exports the MI355X-native implementation in maua_amd."""
from maua_amd.super import RealESRGANer, RRDBNet, SRVGGNetCompact, load_model, upscale  # noqa: F401
