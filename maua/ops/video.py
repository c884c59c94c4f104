"""Drop-in name for maua/ops/video.py:15-128: re-exports the MI355X-native implementation in maua_amd."""
from maua_amd.video import VideoWriter  # noqa: F401
