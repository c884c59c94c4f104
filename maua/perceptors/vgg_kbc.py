"""Drop-in name for maua/perceptors/vgg_kbc.py:10-71."""
from maua_amd.perceptors import KBCPerceptor  # noqa: F401
