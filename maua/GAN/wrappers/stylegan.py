"""Drop-in name for maua/GAN/wrappers/stylegan.py:11-77: re-exports the MI355X-native implementation in maua_amd."""
from maua_amd.stylegan2 import StyleGAN2Mapper as StyleGANMapper  # noqa: F401
from maua_amd.stylegan2 import get_z_latents, parse_seeds  # noqa: F401
