"""Drop-in name for maua/diffusion/processors/guided.py:164-339: re-exports the MI355X-native implementation in maua_amd."""
from maua_amd.diffusion import (GradientGuidedConditioning, GuidedDiffusion, create_model_and_diffusion, create_models,  # noqa: F401
                                model_and_diffusion_defaults)
