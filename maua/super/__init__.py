"""Drop-in package path of the reference (maua/super/...): re-exports the MI355X-native implementation in maua_amd."""
