"""Drop-in name for maua/GAN/wrappers/inference/stylegan2.py:195-470: re-exports the MI355X-native implementation in maua_amd."""
from maua_amd.load import Generator  # noqa: F401
from maua_amd.stylegan2 import MappingNetwork, SynthesisNetwork  # noqa: F401
