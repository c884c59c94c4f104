"""Drop-in name for .../features/rosa/constantq.py:13-293: re-exports the MI355X-native implementation in maua_amd."""
from maua_amd.cqt import constant_q, constant_q_lengths, cqt_frequencies  # noqa: F401
from maua_amd import cqt as _Q


def cqt(y, sr, hop_length=1024, fmin=None, n_bins=84, bins_per_octave=12, tuning=0.0, filter_scale=1, sparsity=0.01):
    """constantq.py:13-26 (complex, like the reference)."""
    return _Q.cqt(y, sr, hop_length, fmin, n_bins, bins_per_octave, tuning, filter_scale, sparsity, magnitude=False)


def vqt(y, sr, hop_length=1024, fmin=None, n_bins=84, gamma=None, bins_per_octave=12, tuning=0.0, filter_scale=1, sparsity=0.01):
    """constantq.py:29-115 (complex, like the reference)."""
    return _Q.vqt(y, sr, hop_length, fmin, n_bins, gamma, bins_per_octave, tuning, filter_scale, sparsity, magnitude=False)
