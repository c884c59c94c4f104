"""Drop-in name for maua/audiovisual/audioreactive/signal.py: re-exports the MI355X-native implementation in maua_amd."""
from maua_amd.signal import (compress, expand, gaussian_filter, normalize, percentile, percentile_clip,  # noqa: F401
                             resample)
