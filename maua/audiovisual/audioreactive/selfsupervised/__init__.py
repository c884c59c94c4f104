"""Drop-in name for maua/audiovisual/audioreactive/selfsupervised/: re-exports the MI355X-native implementation in maua_amd."""
