"""Drop-in name for maua/GAN/: re-exports the MI355X-native implementation in maua_amd."""
