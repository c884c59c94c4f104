"""Drop-in name for maua/audiovisual/audioreactive/selfsupervised/mir.py:24-45: re-exports the MI355X-native implementation in maua_amd."""
from maua_amd.audiovisual.sample import retrieve_music_information  # noqa: F401
from maua_amd.audio import salience_weighted  # noqa: F401
