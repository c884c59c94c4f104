"""Drop-in name for maua/grad.py: re-exports the MI355X-native grad modules in maua_amd (GradModule, CLIPGrads :96-165,
ColorMatchGrads :50-70, VGGGrads :73-93, LPIPSGrads :178-196; LossGrads names what it cannot do without autograd)."""
from maua_amd.grad import (CLIPGrads, ColorMatchGrads, GradModule, LossGrads, LPIPSGrads, VGGGrads,  # noqa: F401
                           differentiable_histogram)
