"""Drop-in name for .../features/rosa/beat.py:24-75: re-exports the MI355X-native implementation in maua_amd."""
from maua_amd.audio import fourier_tempo_frequencies, fourier_tempogram, onset_strength, plp  # noqa: F401
