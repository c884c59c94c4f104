"""Drop-in name for maua/ops/video.py:15-155: re-exports the MI355X-native implementation in maua_amd."""
from maua_amd.video import VideoWriter, write_video  # noqa: F401
