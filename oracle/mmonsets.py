"""Oracle restatement of the madmom branch of the classic onset envelope (test infrastructure only; numpy / scipy).

Follows maua/audiovisual/audioreactive/mir.py:35-56 (type="mm", the reference's default): madmom Signal ->
FramedSignal(frame_size=2048, hop_size=512) -> ShortTimeFourierTransform(circular_shift=True) -> Spectrogram ->
FilteredSpectrogram(num_bands=24) -> mean of five onset detection functions, each divided by its maximum.

PARITY UNPINNED: madmom is un-vendored (setup.py lists it without a version) and cannot be imported here; what is restated
is its published processing chain (madmom 0.16, madmom/audio/{signal,stft,spectrogram,filters}.py and
madmom/features/onsets.py; Boeck & Widmer, "Maximum filter vibrato suppression for onset detection", DAFx 2013; Boeck &
Widmer, "Local group delay based vibrato and tremolo suppression for onset detection", ISMIR 2013; Brossier's modified
Kullback-Leibler):
  frames     n = ceil(len / hop) frames of 2048 samples centred on n * hop, zeros outside the signal
  STFT       symmetric Hann (np.hanning), frame rotated by half its length before the FFT (phase relative to the frame
             centre), bins 0 .. 1023
  filterbank triangular filters on the FFT bins closest to 24-per-octave frequencies around 440 Hz within [30, 17000] Hz,
             duplicate bins removed, neighbouring centres as corners, each filter normalised to unit sum
  differences against the previous frame (a Hann window falls to half its maximum one hop from the centre -> lag 1),
             first frame zero, positive part
  spectral_diff = sum d^2; spectral_flux = sum d; superflux = sum of d against the 3-band maximum of the previous
  frame; complex_flux = that, weighted per band by the minimum over the band's bins (one more either side) of the
  |local group delay| / pi after a 3-frame maximum in time; modified_kullback_leibler = mean log(1 + x[n] / (x[n-1] + eps))."""
import numpy as np
from scipy.ndimage import maximum_filter

EPS = np.spacing(1.0)


def log_frequencies(bands_per_octave, fmin, fmax, fref=440.0):
    left = np.floor(np.log2(float(fmin) / fref) * bands_per_octave)
    right = np.ceil(np.log2(float(fmax) / fref) * bands_per_octave)
    f = fref * 2.0 ** (np.arange(left, right) / float(bands_per_octave))
    f = f[np.searchsorted(f, fmin):]
    return f[:np.searchsorted(f, fmax, "right")]


def frequencies2bins(frequencies, bin_frequencies):
    idx = np.clip(bin_frequencies.searchsorted(frequencies), 1, len(bin_frequencies) - 1)
    left, right = bin_frequencies[idx - 1], bin_frequencies[idx]
    idx = idx - (frequencies - left < right - frequencies)
    return np.unique(idx)


def log_filterbank(sr, fft_size=2048, num_bands=24, fmin=30.0, fmax=17000.0, fref=440.0):
    """[fft_size / 2, n_filters] float32 and the (first, last) non-zero bin of every filter."""
    n_bins = fft_size // 2
    bin_f = np.arange(n_bins) * (sr / float(fft_size))
    bins = frequencies2bins(log_frequencies(num_bands, fmin, fmax, fref), bin_f)
    fb = np.zeros((n_bins, len(bins) - 2), dtype=np.float32)
    for b in range(len(bins) - 2):
        start, center, stop = int(bins[b]), int(bins[b + 1]), int(bins[b + 2])
        if stop - start < 2:
            center, stop = start, start + 1
        data = np.zeros(stop - start, dtype=np.float32)
        data[:center - start] = np.linspace(0, 1, center - start, endpoint=False)
        data[center - start:] = np.linspace(1, 0, stop - center, endpoint=False)
        fb[start:stop, b] = data / data.sum()
    corners = [(int(np.nonzero(fb[:, b])[0][0]), int(np.nonzero(fb[:, b])[0][-1])) for b in range(fb.shape[1])]
    return fb, corners


def stft_frames(audio, frame_size=2048, hop=512):
    y = np.asarray(audio, dtype=np.float32).reshape(-1)
    n = int(np.ceil(len(y) / float(hop)))
    pad = np.concatenate([np.zeros(frame_size // 2, np.float32), y, np.zeros(frame_size + hop, np.float32)])
    idx = np.arange(n)[:, None] * hop + np.arange(frame_size)[None, :]
    frames = pad[idx] * np.hanning(frame_size).astype(np.float32)
    frames = np.concatenate([frames[:, frame_size // 2:], frames[:, :frame_size // 2]], 1)   # circular shift
    return np.fft.fft(frames.astype(np.float64), axis=1)[:, :frame_size // 2]


def onset_functions(audio, sr):
    X = stft_frames(audio)
    mag = np.abs(X)
    fb, corners = log_filterbank(sr)
    S = mag @ fb.astype(np.float64)
    d = np.zeros_like(S)
    d[1:] = S[1:] - S[:-1]
    d = np.maximum(d, 0)
    dm = np.zeros_like(S)
    dm[1:] = S[1:] - maximum_filter(S, size=[1, 3])[:-1]
    dm = np.maximum(dm, 0)
    phase = np.angle(X)
    lgd = np.zeros_like(phase)
    up = np.unwrap(phase, axis=1)
    lgd[:, :-1] = up[:, :-1] - up[:, 1:]
    lgd = maximum_filter(np.abs(lgd) / np.pi, size=[3, 1])
    mask = np.zeros_like(S)
    for b, (lo, hi) in enumerate(corners):
        mask[:, b] = lgd[:, max(lo - 1, 0):min(hi + 2, lgd.shape[1])].min(axis=1)
    mkl = np.zeros_like(S)
    mkl[1:] = S[1:] / (S[:-1] + EPS)
    return {"spectral_diff": (d ** 2).sum(1), "spectral_flux": d.sum(1), "superflux": dm.sum(1),
            "complex_flux": (dm * mask).sum(1), "modified_kullback_leibler": np.log(1 + mkl).mean(1)}


def mm_onset_envelope(audio, sr):
    """mir.py:48-56: the mean of the five functions, each divided by its maximum (before percentile_clip(95))."""
    f = onset_functions(audio, sr)
    return np.mean([v / v.max() for v in f.values()], axis=0)
