"""Drop-in name for maua/diffusion/processors/guided.py:33-339: re-exports the MI355X-native implementation in maua_amd."""
from maua_amd.diffusion import (DiffusionOutput, GradientGuidedConditioning, GuidedDiffusion, SecondaryDiffusionImageNet2,  # noqa: F401
                                create_model_and_diffusion, create_models, model_and_diffusion_defaults)
