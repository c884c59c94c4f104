"""Drop-in name for maua/audiovisual/render/__init__.py: re-exports the MI355X-native implementation in maua_amd."""
from maua_amd.audiovisual.render import *  # noqa: F401,F403
from maua_amd.audiovisual.render import Renderer, get_output_class  # noqa: F401
