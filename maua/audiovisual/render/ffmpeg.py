"""Drop-in name for maua/audiovisual/render/ffmpeg.py:21-75: re-exports the MI355X-native implementation in maua_amd."""
from maua_amd.audiovisual.render.ffmpeg import FFMPEG  # noqa: F401
