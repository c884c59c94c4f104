"""Drop-in name for maua/audiovisual/audioreactive/__init__.py (the `ar` namespace): re-exports the MI355X-native implementation in maua_amd."""
from maua_amd.audiovisual.audioreactive import *  # noqa: F401,F403
from . import audio, latent, mir, signal, util  # noqa: F401
