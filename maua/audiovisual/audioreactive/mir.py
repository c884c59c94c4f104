"""Drop-in name for maua/audiovisual/audioreactive/mir.py:16-61: re-exports the MI355X-native implementation in maua_amd."""
from maua_amd.audiovisual.audioreactive import onsets, rms  # noqa: F401
