"""Drop-in name for maua/grad.py:15-25, 96-165: re-exports the MI355X-native grad modules in maua_amd (GradModule, CLIPGrads)."""
from maua_amd.grad import CLIPGrads, GradModule  # noqa: F401
