"""Drop-in name for maua/audiovisual/audioreactive/selfsupervised/sample.py:36-107: re-exports the MI355X-native implementation in maua_amd."""
from maua_amd.audiovisual.sample import *  # noqa: F401,F403
from maua_amd.audiovisual.sample import generate, main  # noqa: F401

if __name__ == "__main__":
    main()
