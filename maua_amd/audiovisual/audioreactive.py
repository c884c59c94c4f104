"""The ``ar`` namespace patches use (maua/audiovisual/audioreactive/__init__.py): audio loading, features,
envelope post-processing and latent builders, all executed on the HIP device."""
import torch

from ..audio import (gaussian_filter as _gf_selfsup, harmonic as _harmonic, hpss, istft, melspectrogram, onset_strength,  # noqa
                     percussive as _percussive, rms as _rms, spectrogram, stft)
from ..audio_io import load_audio as _load
from ..latent import (copeerp, eerp, multi_weighted, select_modulo, single_weighted, slerp, slerp_loops,  # noqa
                      spline_loops, tempo_loops)
from ..signal import (compress, expand, gaussian_filter, normalize, percentile, percentile_clip, resample,  # noqa
                      sosfilt)


def load_audio(audio_file, offset=0, duration=-1, cache=True):
    """audioreactive/audio.py:15-48 -> (audio tensor, sr, duration)."""
    audio, sr = _load(audio_file, offset, None if duration in (-1, None) else duration)
    return audio, sr, len(audio) / sr


def onsets(audio, sr, type="mm", prepercussive=4, hop_length=512):
    """audioreactive/mir.py:16-61 -> onset envelope, hop 512.  Both back ends of the reference are un-vendored; their
    published algorithms are what the device code implements (parity unpinned):
    type="mm" (the reference's default): madmom's framed STFT -> 24-bands-per-octave filterbank -> mean of five
    normalised onset detection functions (maua_amd/mmonsets.py);
    type="rosa": ``rosa.onset.onset_strength(y, sr)`` at librosa's own framing (n_fft 2048, hop 512, 128 mels up to
    sr / 2, lag 1, centre compensation).  Pass ``hop_length=1024`` for the hop-aligned framing of the selfsupervised
    features (type="rosa" only).  Before either: ``percussive(audio, sr)`` at its default margin of 8 whenever
    ``prepercussive`` is truthy (the reference uses the argument only as a flag, mir.py:29-30, audio.py:91-93); after:
    percentile_clip(95).  type="rosa" keeps librosa's frame count (1 + len // hop; the in-tree spectrogram's dropped
    last column, spectral.py:59-62, is not on this path)."""
    a = torch.as_tensor(audio)
    if prepercussive:
        a = _percussive(a, margin=8.0, hop_length=hop_length)
    if type == "mm":
        if hop_length != 512:
            raise NotImplementedError('onsets(type="mm") runs at madmom\'s framing of the reference (hop 512)')
        from ..mmonsets import mm_onset_envelope
        return percentile_clip(mm_onset_envelope(a, sr), 95).squeeze()
    if type != "rosa":
        raise ValueError(f"unknown onset type {type!r}")
    return percentile_clip(onset_strength(a, sr, hop_length=hop_length, fmax=sr / 2, keep_last=True), 95).squeeze()


def rms(audio, sr):
    return _rms(torch.as_tensor(audio), sr).squeeze(-1)


def harmonic(audio, sr, margin=8):
    """audio.py:85-88: librosa.effects.harmonic(y, margin) - median-filter HPSS at librosa's own framing (n_fft 2048, hop 512;
    librosa un-vendored: the published effect, on the HPSS kernels of the self-supervised path)."""
    return _harmonic(torch.as_tensor(audio), margin=float(margin), hop_length=512)


def percussive(audio, sr, margin=8):
    """audio.py:91-93, see harmonic."""
    return _percussive(torch.as_tensor(audio), margin=float(margin), hop_length=512)


def _butter(audio, sr, cutoff, kind, db_per_octave):
    """maua/audiovisual/audioreactive/audio.py:96-112: ``signal.sosfilt(signal.butter(db_per_octave, cutoff, kind, fs=sr,
    output="sos"), audio)``.  The filter design (at most 12 second-order sections) stays scipy's, as in the reference; the
    recurrence over the clip runs on the device (signal.sosfilt -> maua_sosfilt)."""
    from scipy.signal import butter
    return sosfilt(butter(db_per_octave, cutoff, kind, fs=sr, output="sos"), audio)


def low_pass(audio, sr, fmax=200, db_per_octave=12):
    return _butter(audio, sr, fmax, "low", db_per_octave)


def high_pass(audio, sr, fmin=3000, db_per_octave=12):
    return _butter(audio, sr, fmin, "high", db_per_octave)


def band_pass(audio, sr, fmin=200, fmax=3000, db_per_octave=12):
    return _butter(audio, sr, [fmin, fmax], "band", db_per_octave)
