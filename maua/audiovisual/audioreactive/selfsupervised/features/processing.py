"""Drop-in name for maua/audiovisual/audioreactive/selfsupervised/features/processing.py: re-exports the MI355X-native implementation in maua_amd."""
from maua_amd.audio import (emphasize, gaussian_filter, normalize, quantile, salience_weighted,  # noqa: F401
                            standardize)
