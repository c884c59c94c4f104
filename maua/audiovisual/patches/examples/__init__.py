"""Drop-in name for maua/audiovisual/patches/examples/: re-exports the MI355X-native implementation in maua_amd."""
