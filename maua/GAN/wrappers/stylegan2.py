"""Drop-in name for maua/GAN/wrappers/stylegan2.py:22-340: re-exports the MI355X-native implementation in maua_amd."""
from maua_amd.stylegan2 import StyleGAN2, StyleGAN2Mapper, StyleGAN2Synthesizer, resize_strategy  # noqa: F401
