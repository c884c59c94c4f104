"""Drop-in name for maua/audiovisual/generate.py:16-98: re-exports the MI355X-native implementation in maua_amd."""
from maua_amd.audiovisual.generate import *  # noqa: F401,F403
from maua_amd.audiovisual.generate import generate_audiovisal_from_patch, main  # noqa: F401

if __name__ == "__main__":
    main()
