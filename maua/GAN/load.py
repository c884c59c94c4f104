"""Drop-in name for maua/GAN/load.py:18-207: re-exports the MI355X-native implementation in maua_amd."""
from maua_amd.load import *  # noqa: F401,F403
from maua_amd.load import load_network  # noqa: F401
