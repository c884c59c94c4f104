"""Drop-in name for maua/perceptors/__init__.py:10-101 and vgg_kbc.py:10-71: re-exports the MI355X-native perceptors."""
from maua_amd.perceptors import KBCPerceptor, Perceptor, load_perceptor  # noqa: F401
