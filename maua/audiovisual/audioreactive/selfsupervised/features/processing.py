"""Drop-in name for maua/audiovisual/audioreactive/selfsupervised/features/processing.py: re-exports the MI355X-native implementation in maua_amd."""
from maua_amd.audio import (clamp_lower_percentile, clamp_peaks_percentile, clamp_upper_percentile,  # noqa: F401
                            contrast_enhance, emphasize, gaussian_filter, high_pass, low_pass, median_filter2d, mid_pass,
                            normalize, quantile, salience_weighted, standardize)
