"""`maua` import surface of the MI355X-native hot path: the reference's module paths for the audio-reactive StyleGAN2
render, each re-exporting the maua_amd implementation (HIP kernels behind libmaua_hip.so).  Only the
modules on that path exist; everything else of the reference package is out of scope (DESIGN_LOG.md section 7)."""
