"""Drop-in name for maua/ops/image.py:198-240: re-exports the MI355X-native implementation in maua_amd."""
from maua_amd.ops import resample  # noqa: F401
