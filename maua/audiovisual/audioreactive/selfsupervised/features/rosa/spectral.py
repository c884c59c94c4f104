"""Drop-in name for .../features/rosa/spectral.py: re-exports the MI355X-native implementation in maua_amd."""
from maua_amd.audio import (dct, hpss, istft, magphase, median_filter2d, mel, mel_frequencies, melspectrogram,  # noqa: F401
                            power_to_db, spectrogram, stft)
from maua_amd.cqt import chroma_cens, chroma_cqt, spline_quantize  # noqa: F401
