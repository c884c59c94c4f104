"""Drop-in name for .../features/rosa/constantq.py:13-293: re-exports the MI355X-native implementation in maua_amd."""
from maua_amd.cqt import constant_q, constant_q_lengths, cqt, cqt_frequencies  # noqa: F401


def vqt(y, sr, hop_length=1024, fmin=None, n_bins=84, gamma=None, bins_per_octave=12, tuning=0.0, filter_scale=1, sparsity=0.01):
    """constantq.py:43-115; only the constant-Q case (gamma = 0) is built."""
    if gamma not in (None, 0, 0.0):
        raise NotImplementedError("vqt with gamma != 0")
    return cqt(y, sr, hop_length, fmin, n_bins, bins_per_octave, tuning, filter_scale, sparsity, magnitude=False)
