"""Drop-in name for maua/audiovisual/render/memmap.py: re-exports the MI355X-native implementation in maua_amd."""
from maua_amd.audiovisual.render.memmap import *  # noqa: F401,F403
