"""Drop-in name for .../features/rosa/convert.py:15-66: re-exports the MI355X-native implementation in maua_amd."""
from maua_amd.audio import hz_to_mel, mel_frequencies, mel_to_hz  # noqa: F401
from maua_amd.cqt import cq_to_chroma  # noqa: F401
from maua_amd.audio import power_to_db  # noqa: F401,E402
from maua_amd.cqt import hz_to_midi, hz_to_octs, note_to_hz  # noqa: F401,E402
