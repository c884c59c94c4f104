"""Drop-in name for maua/GAN/wrappers/stylegan.py:11-77: re-exports the MI355X-native implementation in maua_amd."""
from maua_amd.stylegan2 import StyleGAN2Mapper as StyleGANMapper  # noqa: F401
from maua_amd.stylegan2 import get_z_latents, parse_seeds  # noqa: F401
from maua_amd.stylegan2 import StyleGAN2 as StyleGAN  # noqa: F401,E402  (stylegan.py:39: the StyleGAN2 wrapper's base class)
from maua_amd.stylegan2 import StyleGAN2Synthesizer as StyleGANSynthesizer  # noqa: F401,E402  (stylegan.py:35)
