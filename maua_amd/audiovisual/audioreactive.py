"""The ``ar`` namespace patches use (maua/audiovisual/audioreactive/__init__.py): audio loading, features,
envelope post-processing and latent builders, all executed on the HIP device."""
import torch

from ..audio import (gaussian_filter as _gf_selfsup, harmonic, hpss, istft, melspectrogram, onset_strength, percussive,  # noqa
                     rms as _rms, spectrogram, stft)
from ..audio_io import load_audio as _load
from ..latent import (copeerp, eerp, multi_weighted, select_modulo, single_weighted, slerp, slerp_loops,  # noqa
                      spline_loops, tempo_loops)
from ..signal import compress, expand, gaussian_filter, normalize, percentile, percentile_clip, resample  # noqa


def load_audio(audio_file, offset=0, duration=-1, cache=True):
    """audioreactive/audio.py:15-48 -> (audio tensor, sr, duration)."""
    audio, sr = _load(audio_file, offset, None if duration in (-1, None) else duration)
    return audio, sr, len(audio) / sr


def onsets(audio, sr, type="rosa", prepercussive=4, hop_length=512):
    """audioreactive/mir.py:16-61, type="rosa": ``rosa.effects.percussive(audio, margin)`` then
    ``rosa.onset.onset_strength(y, sr)`` then ``percentile_clip(95)`` at librosa's own framing (n_fft 2048, hop 512, 128
    mels up to sr / 2, lag 1, centre compensation) - librosa is un-vendored, its published algorithm is what the HIP
    kernels implement (HPSS medians / soft masks / mel / dB / rectified difference; parity unpinned).  Pass
    ``hop_length=1024`` for the hop-aligned framing of the selfsupervised features.  type="mm" needs madmom's filterbank
    and five spectral-flux variants (un-vendored) and is rejected."""
    if type != "rosa":
        raise NotImplementedError('onsets(type="mm") needs madmom, which the reference does not vendor; use type="rosa"')
    a = torch.as_tensor(audio)
    if prepercussive:
        a = percussive(a, margin=float(prepercussive), hop_length=hop_length)
    return percentile_clip(onset_strength(a, sr, hop_length=hop_length, fmax=sr / 2), 95).squeeze()


def rms(audio, sr):
    return _rms(torch.as_tensor(audio), sr).squeeze(-1)


def _butter(audio, sr, cutoff, kind, db_per_octave):
    """maua/audiovisual/audioreactive/audio.py:96-112: a Butterworth IIR (scipy sosfilt) over the decoded numpy
    signal — a serial recurrence on the host, once per clip, exactly as the reference runs it (scipy is a listed
    dependency of the reference and part of this image)."""
    from scipy import signal as sps
    a = audio.detach().cpu().numpy() if isinstance(audio, torch.Tensor) else audio
    return sps.sosfilt(sps.butter(db_per_octave, cutoff, kind, fs=sr, output="sos"), a)


def low_pass(audio, sr, fmax=200, db_per_octave=12):
    return _butter(audio, sr, fmax, "low", db_per_octave)


def high_pass(audio, sr, fmin=3000, db_per_octave=12):
    return _butter(audio, sr, fmin, "high", db_per_octave)


def band_pass(audio, sr, fmin=200, fmax=3000, db_per_octave=12):
    return _butter(audio, sr, [fmin, fmax], "band", db_per_octave)
