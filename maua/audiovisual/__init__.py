"""Drop-in name for maua/audiovisual/__init__.py: re-exports the MI355X-native implementation in maua_amd."""
