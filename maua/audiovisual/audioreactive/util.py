"""Drop-in name for maua/audiovisual/audioreactive/util.py: re-exports the MI355X-native implementation in maua_amd."""
from maua_amd.stylegan2 import get_z_latents, parse_seeds  # noqa: F401
