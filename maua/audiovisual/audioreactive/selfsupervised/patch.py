"""Drop-in name for maua/audiovisual/audioreactive/selfsupervised/patch.py:34-197: re-exports the MI355X-native implementation in maua_amd."""
from maua_amd.audiovisual.sample import Patch, random_choice  # noqa: F401
