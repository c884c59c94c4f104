"""Drop-in name for maua/ops/: re-exports the MI355X-native implementation in maua_amd."""
