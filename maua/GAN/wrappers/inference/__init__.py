"""Drop-in name for maua/GAN/wrappers/inference/: re-exports the MI355X-native implementation in maua_amd."""
