"""Drop-in name for maua/audiovisual/audioreactive/selfsupervised/features/audio.py: re-exports the MI355X-native implementation in maua_amd."""
from maua_amd.audio import (chromagram, drop_strength, harmonic, mfcc, onsets, percussive, pulse, rms, spectral_contrast,  # noqa: F401
                            spectral_flatness, tonnetz)
