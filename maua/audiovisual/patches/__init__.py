"""Drop-in name for maua/audiovisual/patches/: re-exports the MI355X-native implementation in maua_amd."""
