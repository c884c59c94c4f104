"""Drop-in name for maua/audiovisual/audioreactive/selfsupervised/noise.py: re-exports the MI355X-native implementation in maua_amd."""
from maua_amd.noise import Average, Blend, Loop, Modulate, Multiply, Noise, ScaleBias, noise_patch  # noqa: F401
