"""Drop-in name for maua/ops/io.py:47-70 (tensor2bytes, the float -> rgb24 conversion in front of ffmpeg): re-exports the MI355X-native implementation in maua_amd."""
from maua_amd.video import tensor2bytes  # noqa: F401
