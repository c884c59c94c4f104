"""Drop-in name for maua/audiovisual/patches/base/stylegan2.py:7-53: re-exports the MI355X-native implementation in maua_amd."""
from maua_amd.audiovisual.patches.base.stylegan2 import StyleGAN2Patch  # noqa: F401
