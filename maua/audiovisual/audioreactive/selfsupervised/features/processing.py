"""Drop-in name for maua/audiovisual/audioreactive/selfsupervised/features/processing.py: re-exports the MI355X-native implementation in maua_amd."""
from maua_amd.audio import (emphasize, gaussian_filter, median_filter2d, normalize, quantile,  # noqa: F401
                            salience_weighted, standardize)
