"""Drop-in name for .../features/rosa/pitch.py:9-123: re-exports the MI355X-native implementation in maua_amd."""
from maua_amd.cqt import estimate_tuning, piptrack, pitch_tuning  # noqa: F401
