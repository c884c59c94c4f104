"""Drop-in name for maua/GAN/wrappers/inference/stylegan2.py:29-470: re-exports the MI355X-native implementation in maua_amd."""
from maua_amd.load import Generator  # noqa: F401
from maua_amd.modules import Conv2dLayer, FullyConnectedLayer, SynthesisBlock, SynthesisLayer, ToRGBLayer  # noqa: F401
from maua_amd.stylegan2 import MappingNetwork, SynthesisNetwork  # noqa: F401
