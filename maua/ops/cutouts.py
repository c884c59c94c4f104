"""Drop-in name for maua/ops/cutouts.py:8-217: re-exports the MI355X-native cutouts in maua_amd.grad."""
from maua_amd.grad import Cutouts, DangoCutouts, MauaCutouts, make_cutouts, random_cutouts  # noqa: F401
