"""Drop-in name for maua/audiovisual/audioreactive/audio.py:15-112: re-exports the MI355X-native implementation in maua_amd."""
from maua_amd.audiovisual.audioreactive import (band_pass, harmonic, high_pass, load_audio, low_pass,  # noqa: F401
                                                percussive)
