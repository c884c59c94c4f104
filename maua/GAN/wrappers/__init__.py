"""Drop-in name for maua/GAN/wrappers/__init__.py:20-99: re-exports the MI355X-native implementation in maua_amd."""
from maua_amd.stylegan2 import MauaGenerator, MauaMapper, MauaSynthesizer, get_generator_class  # noqa: F401
