"""Drop-in name for maua/diffusion/image.py:76-125: get_diffusion_model for the guided processor (the grad-module list around
GuidedDiffusion); the multi-resolution image pipeline around it (tiling, super-resolution between scales) is out of scope."""
from maua_amd.diffusion import get_diffusion_model  # noqa: F401
