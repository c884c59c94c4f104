"""Drop-in name for maua/audiovisual/audioreactive/selfsupervised/features/efficient_quantile/ (the C++ extension): re-exports the MI355X-native implementation in maua_amd."""
from maua_amd.audio import quantile  # noqa: F401
