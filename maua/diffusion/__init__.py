"""Drop-in namespace for the guided-diffusion slice of maua/diffusion (BASELINE configs[3]): re-exports maua_amd.diffusion."""
