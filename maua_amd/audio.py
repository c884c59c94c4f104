"""Audio features on the HIP device (drop-in names for the reference's pure-torch librosa layer).

  stft / istft / spectrogram / melspectrogram / mel / mel_frequencies / hpss / magphase
      <- maua/audiovisual/audioreactive/selfsupervised/features/rosa/spectral.py:10-161
  power_to_db / hz_to_mel / mel_to_hz <- rosa/convert.py:7-66       onset_strength <- rosa/beat.py:10-23
  harmonic / percussive / onsets / rms <- features/audio.py:13-37
  gaussian_filter / normalize / standardize / median_filter2d <- features/processing.py:11-85
  quantile <- features/efficient_quantile/__init__.py:6-7 (C++ efficient_quantile.cpp)
  salience_weighted <- selfsupervised/mir.py:13-21

Inputs may be CPU or device tensors; results live on the HIP device.  n_fft = 2048 / hop = 1024 (the only
framing the reference's features use).  Every function goes through libmaua_hip.so; small constant tables (mel
basis, Gaussian taps, linspace grids) are built on the host exactly as the reference builds them.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib as L

N_FFT, HOP, N_BINS = 2048, 1024, 1025


def _f32(t):
    return L.dev_tensor(torch.as_tensor(t), torch.float32)


def _default_framing(n_fft, hop_length):
    """the features' own framing (n_fft 2048, hop 1024: one hop per video frame) runs the specialised kernels; every other
    power-of-two n_fft <= 2048 / hop (librosa's 2048 / 512 of the classic API) the general ones"""
    return n_fft == N_FFT and hop_length == HOP


# ------------------------------------------------------------------------------------------------ spectra
def _window_tensor(window, n_fft):
    """rosa/spectral.py:10-32's ``window`` argument: a window FUNCTION (default torch.hann_window), None = rectangular (what
    torch.stft does without a window), or a ready tensor.  -> None for the Hann window of the fast path, else the tensor."""
    if window is torch.hann_window:
        return None
    if window is None:
        return torch.ones(n_fft)
    return window(n_fft) if callable(window) else window


def stft(y, n_fft=2048, hop_length=1024, center=True, window=torch.hann_window, pad_mode="reflect", return_complex=True):
    """-> complex64 [n_fft//2 + 1, 1 + len(y)//hop] (a transposed view of the frame-major device buffer)."""
    if not center or pad_mode != "reflect":
        raise NotImplementedError("center=True / reflect padding only")
    if not return_complex:
        raise NotImplementedError("return_complex=True only")
    win = _window_tensor(window, n_fft)
    if win is not None or not _default_framing(n_fft, hop_length):
        return stft_general(y, n_fft, hop_length, win)
    y = _f32(y).reshape(-1)
    frames = 1 + y.numel() // HOP
    out = torch.empty((frames, N_BINS, 2), dtype=torch.float32, device=y.device)
    L.check(L.lib().maua_stft(L.ctx(y.device), L.ptr(y), y.numel(), L.ptr(out)))
    return torch.view_as_complex(out).T


def _frame_major(spec):
    """complex [1025, frames] (any strides) -> contiguous float view [frames, 1025, 2]."""
    s = spec.T.contiguous() if spec.T.is_contiguous() is False else spec.T
    return torch.view_as_real(s.contiguous())


def istft(spec, n_fft=2048, hop_length=1024, center=True, window=torch.hann_window, length=None):
    if not center:
        raise NotImplementedError("center=True only")
    win = _window_tensor(window, n_fft)
    if win is not None or not _default_framing(n_fft, hop_length):
        return istft_general(spec, n_fft, hop_length, hop_length * (spec.shape[1] - 1) if length is None else length, win)
    spec = L.dev_tensor(spec, torch.complex64) if not spec.is_cuda else spec
    buf = _frame_major(spec)
    frames = buf.shape[0]
    if length is None:
        length = HOP * (frames - 1)
    y = torch.empty((length,), dtype=torch.float32, device=buf.device)
    L.check(L.lib().maua_istft(L.ctx(buf.device), L.ptr(buf), frames, int(length), L.ptr(y)))
    return y


def spectrogram(y, n_fft=2048, hop_length=1024, power=1, window=torch.hann_window, center=True, pad_mode="reflect"):
    """spectral.py:59-62 (the last STFT column is dropped)."""
    D = stft(y, n_fft, hop_length, center, window, pad_mode)[:, :-1]
    buf = _frame_major(D)
    mag = torch.empty(buf.shape[:2], dtype=torch.float32, device=buf.device)
    L.check(L.lib().maua_magnitude(L.ctx(buf.device), L.ptr(buf), C.c_long(mag.numel()), C.c_float(float(power)),
                                   L.ptr(mag)))
    return mag.T


def magphase(D, power=1.0):
    buf = _frame_major(D)
    mag = torch.empty(buf.shape[:2], dtype=torch.float32, device=buf.device)
    L.check(L.lib().maua_magnitude(L.ctx(buf.device), L.ptr(buf), C.c_long(mag.numel()), C.c_float(1.0), L.ptr(mag)))
    mag = mag.T
    phase = torch.where(mag > 0, D / mag.clamp_min(1e-38), torch.ones_like(D))
    return mag ** power, phase


def hz_to_mel(frequencies, htk=False, device="cpu"):
    """convert.py:15-40 (host constant math)."""
    f = torch.as_tensor(frequencies, dtype=torch.float32)
    if htk:
        return 2595.0 * torch.log10(1.0 + f / 700.0)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    if f.ndim:
        t = f >= min_log_hz
        mels[t] = min_log_mel + torch.log(f[t] / min_log_hz) / logstep
    elif f >= min_log_hz:
        mels = min_log_mel + torch.log(f / min_log_hz) / logstep
    return mels


def mel_to_hz(mels, htk=False):
    if htk:
        return 700.0 * (10.0 ** (mels / 2595.0) - 1.0)
    f_sp = 200.0 / 3
    freqs = f_sp * mels
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    if mels.ndim:
        t = mels >= min_log_mel
        freqs[t] = min_log_hz * torch.exp(logstep * (mels[t] - min_log_mel))
    elif mels >= min_log_mel:
        freqs = min_log_hz * torch.exp(logstep * (mels - min_log_mel))
    return freqs


def mel_frequencies(n_mels=128, fmin=0.0, fmax=11025.0, htk=False, device="cpu"):
    return mel_to_hz(torch.linspace(hz_to_mel(fmin, htk), hz_to_mel(fmax, htk), n_mels), htk)


def mel(sr, n_fft, n_mels=128, fmin=0.0, fmax=None, htk=False, dtype=torch.float, device="cpu"):
    """spectral.py:81-110 — [n_mels, 1 + n_fft//2] Slaney filterbank, a host-built constant."""
    if fmax is None:
        fmax = float(sr) / 2
    fftfreqs = torch.linspace(0, float(sr) / 2, int(1 + n_fft // 2))
    mel_f = mel_frequencies(n_mels + 2, fmin=fmin, fmax=fmax, htk=htk)
    fdiff = torch.diff(mel_f)
    ramps = mel_f.reshape(-1, 1) - fftfreqs
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    weights = torch.clamp(torch.minimum(lower, upper), min=0)
    enorm = 2.0 / (mel_f[2: n_mels + 2] - mel_f[:n_mels])
    return (weights * enorm[:, None]).to(dtype)


def melspectrogram(y, sr, n_fft=2048, hop_length=1024, window=torch.hann_window, center=True, pad_mode="reflect", power=2.0,
                   fmax=None, keep_last=False):
    """spectral.py:65-70 (drops the last STFT column like the in-tree spectrogram, :59-62); ``keep_last``: librosa's
    own framing (1 + len // hop frames), what the classic mir.onsets(type="rosa") sees."""
    if power != 2.0:
        raise NotImplementedError("power must be 2 (the onset path)")
    D = stft(y, n_fft, hop_length, center, window, pad_mode)
    buf = _frame_major(D)
    T = buf.shape[0] - (0 if keep_last else 1)
    basis = _f32(mel(sr, n_fft, fmax=fmax))
    out = torch.empty((basis.shape[0], T), dtype=torch.float32, device=buf.device)
    L.check(L.lib().maua_mel_power(L.ctx(buf.device), L.ptr(buf), T, L.ptr(basis), basis.shape[0], L.ptr(out)))
    return out


def power_to_db(magnitude, ref_value=1.0, amin=1e-10, top_db=80.0):
    """convert.py:7-12 on a device tensor (elementwise + one max; runs in the onset kernel on the hot path —
    this standalone form composes torch device ops for API parity)."""
    m = _f32(magnitude)
    log_spec = 10.0 * torch.log10(torch.clamp(m, min=amin)) - 10.0 * np.log10(max(amin, float(ref_value)))
    if top_db is not None:
        log_spec = torch.maximum(log_spec, log_spec.max() - top_db)
    return log_spec


def onset_strength(y, sr, hop_length=1024, n_fft=2048, aggregate=torch.mean, fmax=11025.0, keep_last=False):
    """beat.py:10-23 -> [T] on device.  ``aggregate``: None / torch.mean, or "median" (what plp passes, beat.py:44).
    ``fmax``: the in-tree function fixes 11 025 Hz; librosa's own default (the classic mir.onsets) is sr / 2."""
    if n_fft != N_FFT:
        raise NotImplementedError("onset_strength: n_fft must be 2048 (the mel kernel's bin count)")
    S = melspectrogram(y, sr, n_fft=n_fft, hop_length=hop_length, fmax=fmax, keep_last=keep_last)
    n_mels, T = S.shape
    env = torch.empty((T,), dtype=torch.float32, device=S.device)
    pad_width = 1 + n_fft // (2 * hop_length)
    if aggregate is None or aggregate is torch.mean or aggregate == "mean":
        agg = 0
    elif aggregate == "median":
        agg = 1
    else:
        raise NotImplementedError("onset_strength: aggregate must be torch.mean or 'median'")
    L.check(L.lib().maua_onset_from_mel(L.ctx(S.device), L.ptr(S), n_mels, T, C.c_float(1e-10), C.c_float(80.0),
                                        pad_width, agg, L.ptr(env)))
    return env


def hpss(S, ks=31, power=2.0, margin=1.0):
    """spectral.py:145-161 on a complex STFT [1025, frames] -> (harmonic, percussive) complex spectra."""
    if ks != 31:
        raise NotImplementedError("median window must be 31 (the reference default)")
    if not torch.is_complex(S):
        raise NotImplementedError("pass the complex STFT")
    S = S if S.is_cuda else L.dev_tensor(S, torch.complex64)
    buf = _frame_major(S)
    frames = buf.shape[0]
    if buf.shape[1] != N_BINS:
        raise ValueError("expected 1025 frequency bins")
    h = torch.empty_like(buf)
    p = torch.empty_like(buf)
    L.check(L.lib().maua_hpss(L.ctx(buf.device), L.ptr(buf), frames, C.c_float(float(margin)), C.c_float(float(power)),
                              L.ptr(h), L.ptr(p)))
    return torch.view_as_complex(h).T, torch.view_as_complex(p).T


def median_filter2d(x, k=(3, 3), s=(1, 1), p=(1, 1, 1, 1), mode="reflect"):
    """processing.py:75-85 for the two shapes hpss uses: k=(1,31),p=(15,15,0,0) and k=(31,1),p=(0,0,15,15);
    x [1,1,bins,frames]."""
    x = _f32(x)
    if mode != "reflect" or tuple(s) != (1, 1):
        raise NotImplementedError
    bins, frames = x.shape[-2:]
    fm = x.reshape(bins, frames).T.contiguous()  # frame-major
    out = torch.empty_like(fm)
    if tuple(k) == (1, 31) and tuple(p) == (15, 15, 0, 0):
        axis = 0  # along time
    elif tuple(k) == (31, 1) and tuple(p) == (0, 0, 15, 15):
        axis = 1  # along frequency
    else:
        raise NotImplementedError("only the 31-tap harmonic / percussive medians of hpss")
    L.check(L.lib().maua_median31(L.ctx(fm.device), L.ptr(fm), frames, bins, axis, L.ptr(out)))
    return out.T.reshape(x.shape)


def harmonic(audio, margin=8.0, hop_length=1024):
    y = _f32(audio)
    return istft(hpss(stft(y, hop_length=hop_length), margin=margin)[0], hop_length=hop_length, length=y.numel())


def percussive(audio, margin=8.0, hop_length=1024):
    y = _f32(audio)
    return istft(hpss(stft(y, hop_length=hop_length), margin=margin)[1], hop_length=hop_length, length=y.numel())


# ------------------------------------------------------------------------------------------------ envelopes
def normalize(array, eps=1e-8):
    """processing.py:53-56: (x - min) / (max(x - min) + 1e-8)."""
    x = _f32(array)
    y = torch.empty_like(x)
    L.check(L.lib().maua_normalize(L.ctx(x.device), L.ptr(x), C.c_long(x.numel()), C.c_float(eps), L.ptr(y)))
    return y


def onsets(audio, sr):
    """features/audio.py:27-28 -> [T, 1]."""
    return normalize(onset_strength(percussive(audio), sr).unsqueeze(-1))


def rms(y, sr=None, frame_length=2048, hop_length=1024, center=True, pad_mode="reflect"):
    """features/audio.py:31-37 -> [T, 1] (drops the last frame)."""
    if not center or pad_mode != "reflect":
        raise NotImplementedError
    y = _f32(y).reshape(-1)
    n_frames = (y.numel() + 2 * (frame_length // 2) - frame_length) // hop_length + 1 - 1
    out = torch.empty((max(n_frames, 0),), dtype=torch.float32, device=y.device)
    L.check(L.lib().maua_rms(L.ctx(y.device), L.ptr(y), y.numel(), frame_length, hop_length, n_frames, L.ptr(out)))
    return out.unsqueeze(-1)


def gaussian_taps(sigma, n_frames, causal=None, classic=False):
    """The reference's kernel construction (signal.py:124-132 / processing.py:18-24), on the host."""
    radius = min(int(sigma * 4), 3 * n_frames)
    k = torch.arange(-radius, radius + 1, dtype=torch.float32)
    k = torch.exp(-0.5 / sigma ** 2 * k ** 2)
    if classic and causal is not None:
        k[radius + 1:] *= causal if isinstance(causal, float) else 0
    return k / k.sum(), radius


def gaussian_filter(x, sigma, mode="circular", causal=1, _classic=False):
    """processing.py:11-49 (ignores ``causal``, like the reference's selfsupervised version).  Filters along
    dim 0; any trailing shape."""
    x = _f32(x)
    shape = x.shape
    T = shape[0]
    Cn = x.numel() // max(T, 1)
    taps, radius = gaussian_taps(sigma, T, causal, _classic)
    if radius > T:
        print(f"WARNING: Gaussian filter radius ({int(sigma * 4)}) is larger than number of frames ({T}).\n\t "
              f"Filter size has been lowered to ({radius}). You might want to consider lowering sigma ({sigma}).")
    y = torch.empty_like(x)
    L.check(L.lib().maua_gaussian_filter1d(L.ctx(x.device), L.ptr(x), L.ptr(_f32(taps)), radius, T, C.c_long(Cn),
                                           L.PAD_MODES[mode], L.ptr(y)))
    if x.dim() < 3:  # the reference lifts to 3-D and then squeezes EVERY singleton axis (processing.py:46-48):
        y = y.reshape(shape + (1,) * (3 - x.dim())).squeeze()  # [T, 1] comes back as [T]
    return y


def order_stat(x, mode, q=0.0, k=1, mask=None):
    """device triple {result, x_(lo), x_(hi)} and int64 ranks {lo, hi}; see maua_order_stat in the header."""
    x = _f32(x).reshape(-1)
    out = torch.empty((3,), dtype=torch.float32, device=x.device)
    ranks = torch.empty((2,), dtype=torch.int64, device=x.device)
    if mask is not None:
        mask = mask.to(device=x.device, dtype=torch.uint8).contiguous()
    L.check(L.lib().maua_order_stat(L.ctx(x.device), L.ptr(x), L.ptr(mask), C.c_long(x.numel()), mode,
                                    C.c_float(float(q)), C.c_long(int(k)), L.ptr(out), L.ptr(ranks)))
    return out, ranks


def quantile(tensor, q):
    """efficient_quantile/__init__.py:6-7: midpoint quantile with a float32 q, NaNs ignored -> 0-dim tensor."""
    return order_stat(tensor, 0, q=q)[0][0]


def standardize(array):
    """processing.py:59-62"""
    x = _f32(array)
    lo, _ = order_stat(x, 0, q=0.25)
    hi, _ = order_stat(x, 0, q=0.75)
    y = torch.empty_like(x)
    L.check(L.lib().maua_clamp(L.ctx(x.device), L.ptr(x), L.ptr(lo), L.ptr(hi), C.c_float(0), C.c_float(1e-10),
                               C.c_long(x.numel()), L.ptr(y)))
    return normalize(y)


def salience_weighted(envelope, short_sigma=5, long_sigma=80):
    """selfsupervised/mir.py:13-21 (three elementwise ops on a [T] envelope between two HIP filters)."""
    e = _f32(envelope)
    if e.dim() > 1:
        e = e.squeeze(1)
    short = gaussian_filter(e, short_sigma, mode="reflect")
    long = gaussian_filter(e, long_sigma, mode="reflect")
    w = (short / long) ** 2 * e
    return w.unsqueeze(1) if w.dim() < 2 else w


# ------------------------------------------------------------------------------------------------ SURVEY 8(f) N3
# further features of selfsupervised/features/audio.py (first batch: the ones that do not need the constant-Q stack)
def emphasize(envs, strength, percentile):
    """processing.py:133-139 on a [T] / [T, 1] envelope (the reference reduces over dim 0; one column here)."""
    x = _f32(envs)
    shape = x.shape
    if x.numel() != x.shape[0]:
        raise NotImplementedError("emphasize: one envelope column at a time")
    x = x.reshape(-1)
    lib, ctx = L.lib(), L.ctx(x.device)
    mm = torch.empty((2,), dtype=torch.float32, device=x.device)
    L.check(lib.maua_minmax(ctx, L.ptr(x), C.c_long(x.numel()), L.ptr(mm)))
    xn = torch.empty_like(x)
    L.check(lib.maua_normalize(ctx, L.ptr(x), C.c_long(x.numel()), C.c_float(0.0), L.ptr(xn)))
    q, _ = order_stat(xn, 2, q=percentile / 100)
    y = torch.empty_like(x)
    L.check(lib.maua_emphasize(ctx, L.ptr(xn), C.c_long(x.numel()), L.ptr(mm), L.ptr(q), C.c_float(float(strength)), L.ptr(y)))
    return y.reshape(shape)


def _biquad(waveform, b, a):
    """torchaudio.functional.biquad = lfilter(waveform, a, b, clamp=True) for one second-order section (torchaudio un-vendored:
    the published filter; the recurrence runs in float64 on the device and is rounded to float32 once, torchaudio's runs in the
    waveform's float32)."""
    from .signal import sosfilt
    x = _f32(waveform)
    y = sosfilt([[b[0], b[1], b[2], a[0], a[1], a[2]]], x)
    out = torch.empty_like(x)
    one = torch.ones((1,), dtype=torch.float32, device=x.device)
    y32 = y.float()
    L.check(L.lib().maua_clamp(L.ctx(x.device), L.ptr(y32), None, L.ptr(one), C.c_float(-1.0), C.c_float(0.0),
                               C.c_long(x.numel()), L.ptr(out)))
    return out


def low_pass(audio, sr, fmax=200):
    """processing.py:142-143: torchaudio.functional.lowpass_biquad(audio, sr, fmax) (Q = 0.707; RBJ cookbook coefficients)."""
    w0 = 2 * math.pi * fmax / sr
    alpha = math.sin(w0) / 2 / 0.707
    b0 = (1 - math.cos(w0)) / 2
    return _biquad(audio, (b0, 1 - math.cos(w0), b0), (1 + alpha, -2 * math.cos(w0), 1 - alpha))


def high_pass(audio, sr, fmin=4000):
    """processing.py:150-151: torchaudio.functional.highpass_biquad(audio, sr, fmin) (Q = 0.707)."""
    w0 = 2 * math.pi * fmin / sr
    alpha = math.sin(w0) / 2.0 / 0.707
    b0 = (1 + math.cos(w0)) / 2
    return _biquad(audio, (b0, -1 - math.cos(w0), b0), (1 + alpha, -2 * math.cos(w0), 1 - alpha))


def mid_pass(audio, sr, fmin=200, fmax=4000):
    """processing.py:146-147, as written there: the HIGH pass runs at fmax and the LOW pass at fmin."""
    return low_pass(high_pass(audio, sr, fmax), sr, fmin)


def contrast_enhance(audio, sr, strength=75):
    """processing.py:154-155: torchaudio.functional.contrast(audio, strength)."""
    x = _f32(audio)
    y = torch.empty_like(x)
    L.check(L.lib().maua_contrast(L.ctx(x.device), L.ptr(x), C.c_long(x.numel()), C.c_float(float(strength)), L.ptr(y)))
    return y


def _clamp_columns(signal, bound):
    """clamp every column of a [T] / [T, C] signal by its own device-scalar bounds: bound(col) -> (lo or None, hi or None)."""
    x = _f32(signal)
    cols = x.reshape(x.shape[0], -1)
    inf = torch.full((1,), float("inf"), dtype=torch.float32, device=x.device)
    out = []
    for col in cols.unbind(1):
        col = col.contiguous()
        lo, hi = bound(col)
        y = torch.empty_like(col)
        L.check(L.lib().maua_clamp(L.ctx(x.device), L.ptr(col), L.ptr(lo), L.ptr(inf if hi is None else hi),
                                   C.c_float(float("-inf")), C.c_float(0.0), C.c_long(col.numel()), L.ptr(y)))
        out.append(y)
    return torch.stack(out, dim=1).reshape(x.shape)


def clamp_upper_percentile(signal, percentile):
    """processing.py:125-126: clamp(signal, None, torch.quantile(signal, percentile / 100, dim=0))."""
    return _clamp_columns(signal, lambda col: (None, order_stat(col, 2, q=percentile / 100)[0]))


def clamp_lower_percentile(signal, percentile):
    """processing.py:129-130: clamp(signal, torch.quantile(signal, percentile / 100, dim=0), None)."""
    return _clamp_columns(signal, lambda col: (order_stat(col, 2, q=percentile / 100)[0], None))


def clamp_peaks_percentile(signal, percent):
    """processing.py:102-122: every column clamped from above at the ``percent`` quantile (torch.quantile) of its local peaks
    (strictly greater than both neighbours, indices clamped at the ends) -> [T, C] ([T, 1] for a 1-D signal)."""
    def bound(col):
        mask = torch.empty((col.numel(),), dtype=torch.uint8, device=col.device)
        L.check(L.lib().maua_peak_mask(L.ctx(col.device), L.ptr(col), col.numel(), L.ptr(mask)))
        return None, order_stat(col, 2, q=percent / 100, mask=mask)[0]
    x = _f32(signal)
    return _clamp_columns(x.unsqueeze(1) if x.ndim < 2 else x, bound)


def drop_strength(audio, sr):
    """features/audio.py:40-41 -> [T, 1, 1] (the reference unsqueezes the [T, 1] envelope once more)."""
    return emphasize(gaussian_filter(rms(audio, sr), 10), strength=10, percentile=50).unsqueeze(1)


def _matmul_nt(a, b):
    a, b = _f32(a), _f32(b)
    M, K = a.shape
    N = b.shape[0]
    c = torch.empty((M, N), dtype=torch.float32, device=a.device)
    L.check(L.lib().maua_matmul_nt(L.ctx(a.device), L.ptr(a), L.ptr(b), L.ptr(c), M, N, K))
    return c


def dct(x, norm=None, n_keep=None):
    """rosa/spectral.py:35-56: DCT-II along the last axis (the reference goes through an FFT; here the cosine basis
    [n_keep, N], built in float64 on the host, multiplies the rows on the device)."""
    x = _f32(x)
    N = x.shape[-1]
    n_keep = N if n_keep is None else n_keep
    k = np.arange(n_keep, dtype=np.float64)[:, None]
    n = np.arange(N, dtype=np.float64)[None, :]
    basis = 2.0 * np.cos(np.pi * (2 * n + 1) * k / (2 * N))
    if norm == "ortho":
        basis[0] /= np.sqrt(N) * 2
        basis[1:] /= np.sqrt(N / 2) * 2
    out = _matmul_nt(x.reshape(-1, N), torch.from_numpy(basis.astype(np.float32)).to(x.device))
    return out.reshape(*x.shape[:-1], n_keep)


def mfcc(y, sr, n_mfcc=20, norm=False, **kwargs):
    """features/audio.py:65-70 -> [T, n_mfcc]."""
    S = power_to_db(melspectrogram(y, sr, **kwargs))           # [n_mels, T]
    M = dct(S.permute(1, 0).contiguous(), norm="ortho", n_keep=n_mfcc)  # [T, n_mfcc]
    if norm is True:
        M = M / M.norm(p=2)
    return M


def chromagram(audio, sr):
    """features/audio.py:44-45 -> [T, 12] (constant-Q chain: maua_amd/cqt.py)."""
    from .cqt import chromagram as _chromagram
    return _chromagram(audio, sr)


def tonnetz(y=None, sr=None, chroma_fn=None, chroma=None):
    """features/audio.py:50-62 -> [T, 6]: tonal centroids of the chromagram (``chroma_fn(y, sr)`` -> [12, T], default the
    CENS chromagram like the reference; ``chroma`` hands a ready one in)."""
    if chroma is None:
        chroma = chroma_fn(y, sr) if chroma_fn is not None else chromagram(y, sr).T
    ch = _f32(chroma)
    n = ch.shape[0]
    dim_map = torch.linspace(0, 12, n)
    scale = torch.tensor([7.0 / 6, 7.0 / 6, 3.0 / 2, 3.0 / 2, 2.0 / 3, 2.0 / 3])
    V = scale.reshape(-1, 1) * dim_map
    V[::2] -= 0.5
    R = torch.tensor([1, 1, 1, 1, 0.5, 0.5])
    phi = R[:, None] * torch.cos(torch.pi * V)                  # [6, n] (host constants)
    chn = ch / ch.norm(p=1, dim=0)
    return _matmul_nt(chn.T.contiguous(), phi.to(ch.device))     # [T, 6]


def spectral_flatness(y, sr=None, n_fft=2048, hop_length=1024, window=torch.hann_window, center=True, pad_mode="reflect",
                      amin=1e-10, power=2.0):
    """features/audio.py:118-126 -> [T, 1]."""
    buf = _frame_major(stft(y, n_fft, hop_length, center, window, pad_mode))
    T = buf.shape[0] - 1                                         # spectrogram drops the last column
    out = torch.empty((T,), dtype=torch.float32, device=buf.device)
    L.check(L.lib().maua_spectral_flatness(L.ctx(buf.device), L.ptr(buf), T, buf.shape[1], C.c_float(amin),
                                           C.c_float(power), L.ptr(out)))
    return out.unsqueeze(-1)


def contrast_bands(sr, n_fft=2048, fmin=200.0, n_bands=6, quantile=0.02):
    """The band bookkeeping of features/audio.py:89-105 in the reference's own float32 arithmetic:
    [(lo, hi, k)] bin ranges [lo, hi) of every band's sub_band and the number of sorted bins averaged."""
    freq = torch.linspace(0, float(sr) / 2, int(1 + n_fft // 2))
    octa = torch.zeros(n_bands + 2)
    octa[1:] = fmin * (2.0 ** torch.arange(0, n_bands + 1))
    out = []
    for k, (f_low, f_high) in enumerate(zip(octa[:-1], octa[1:])):
        band = torch.logical_and(freq >= f_low, freq <= f_high)
        idx = band.flatten().nonzero()
        if k > 0:
            band[idx[0] - 1] = True
        if k == n_bands:
            band[idx[-1] + 1:] = True
        sel = band.nonzero().flatten()
        lo, hi = int(sel[0]), int(sel[-1]) + 1
        assert int(band.sum()) == hi - lo                        # contiguous by construction
        kk = int(max(torch.round(quantile * torch.sum(band)), torch.ones(())))
        if k < n_bands:
            hi -= 1                                              # sub_band[:-1]
        out.append((lo, hi, kk))
    return out


def spectral_contrast(y, sr, n_fft=2048, hop_length=1024, window=torch.hann_window, center=True, pad_mode="reflect", fmin=200.0,
                      n_bands=6, quantile=0.02, linear=False):
    """features/audio.py:76-115 -> [T, n_bands + 1]."""
    buf = _frame_major(stft(y, n_fft, hop_length, center, window, pad_mode))
    T = buf.shape[0] - 1
    bands = contrast_bands(sr, n_fft, fmin, n_bands, quantile)
    valley = torch.empty((len(bands), T), dtype=torch.float32, device=buf.device)
    peak = torch.empty_like(valley)
    for k, (lo, hi, kk) in enumerate(bands):
        L.check(L.lib().maua_band_sorted_means(L.ctx(buf.device), L.ptr(buf), T, buf.shape[1], lo, hi, kk,
                                               L.ptr(valley[k]), L.ptr(peak[k])))
    if linear:
        return (peak - valley).T
    return (power_to_db(peak) - power_to_db(valley)).T


# ---- predominant local pulse (rosa/beat.py:24-75) ----------------------------------------------------------------
def _pow2(n):
    return n >= 2 and (n & (n - 1)) == 0


def stft_general(y, n_fft, hop_length, window=None):
    """rosa/spectral.py:10-21 for any n_fft <= 8192 and any hop (centre / reflect; powers of two on the LDS FFT - round 6: above 2048
    too -, other lengths as an exact-f32 DFT GEMM) -> complex64
    [n_fft // 2 + 1, 1 + len(y) // hop] as a transposed view of the frame-major buffer."""
    y = _f32(y).reshape(-1)
    win = _f32(torch.hann_window(n_fft) if window is None else window).to(y.device)
    if not _pow2(n_fft):
        if n_fft > 8192:
            raise NotImplementedError("stft: lengths up to 8192")
        return _stft_dft(y, n_fft, hop_length, win)
    if n_fft > 8192:
        raise NotImplementedError("the HIP FFT handles power-of-two lengths up to 8192 (two buffers of n_fft complex values in LDS)")
    frames = 1 + y.numel() // hop_length
    out = torch.empty((frames, n_fft // 2 + 1, 2), dtype=torch.float32, device=y.device)
    L.check(L.lib().maua_stft_general(L.ctx(y.device), L.ptr(y), y.numel(), n_fft, hop_length, L.ptr(win), L.ptr(out)))
    return torch.view_as_complex(out).T


def istft_general(spec, n_fft, hop_length, length, window=None):
    buf = _frame_major(spec if spec.is_cuda else L.dev_tensor(spec, torch.complex64))
    win = _f32(torch.hann_window(n_fft) if window is None else window).to(buf.device)
    if not _pow2(n_fft):
        return _istft_dft(buf, n_fft, hop_length, length, win)
    y = torch.empty((int(length),), dtype=torch.float32, device=buf.device)
    L.check(L.lib().maua_istft_general(L.ctx(buf.device), L.ptr(buf), buf.shape[0], n_fft, hop_length, L.ptr(win),
                                       int(length), L.ptr(y)))
    return y


_DFT_CACHE = {}


def _dft_matrices(n_fft, window):
    """Real DFT / inverse real DFT of length n_fft as GEMM operands with the window folded in (host f64 -> f32):
    fwd [2 nb, n_fft]: row 2k = w cos, row 2k+1 = -w sin;  inv [n_fft, 2 nb]: x[n] = w[n] / N sum_k c_k (Re cos - Im sin),
    c = 1 at DC / Nyquist and 2 elsewhere, their imaginary parts ignored (what a C2R transform does)."""
    # keyed on the window's CONTENTS: two windows of one length and equal sum (an asymmetric window and its flip) must not share
    # matrices with the first one folded in
    w = window.detach().cpu().double()
    key = (n_fft, str(window.device), hash(w.numpy().tobytes()))
    if key not in _DFT_CACHE:
        nb = n_fft // 2 + 1
        n = torch.arange(n_fft, dtype=torch.float64)
        k = torch.arange(nb, dtype=torch.float64)
        ang = 2 * math.pi * ((k[:, None] * n[None, :]) % n_fft) / n_fft
        fwd = torch.stack([ang.cos() * w, -ang.sin() * w], 1).reshape(2 * nb, n_fft)
        c = torch.full((nb,), 2.0, dtype=torch.float64)
        c[0] = 1.0
        s_im = torch.ones(nb, dtype=torch.float64)
        s_im[0] = 0.0
        if n_fft % 2 == 0:
            c[-1], s_im[-1] = 1.0, 0.0
        inv = torch.stack([ang.cos() * c[:, None], -ang.sin() * (c * s_im)[:, None]], 1).reshape(2 * nb, n_fft).T * (w / n_fft)[:, None]
        if len(_DFT_CACHE) >= 8:
            _DFT_CACHE.clear()
        _DFT_CACHE[key] = (L.dev_tensor(fwd.float(), torch.float32), L.dev_tensor(inv.float(), torch.float32))
    return _DFT_CACHE[key]


def _stft_dft(y, n_fft, hop_length, win):
    """stft_general for lengths the Stockham FFT does not take: frames gathered on the device, one exact-f32 GEMM."""
    pad = n_fft // 2
    if pad >= y.numel():
        raise ValueError("stft: reflect padding needs n_fft // 2 < len(y)")
    padded = torch.cat([y[1:pad + 1].flip(0), y, y[-pad - 1:-1].flip(0)])
    frames = padded.unfold(0, n_fft, hop_length).contiguous()
    fwd, _ = _dft_matrices(n_fft, win)
    out = torch.empty((frames.shape[0], n_fft // 2 + 1, 2), dtype=torch.float32, device=y.device)
    L.check(L.lib().maua_matmul_nt(L.ctx(y.device), L.ptr(frames), L.ptr(fwd), L.ptr(out), frames.shape[0], 2 * (n_fft // 2 + 1), n_fft))
    return torch.view_as_complex(out).T


def _istft_dft(buf, n_fft, hop_length, length, win):
    _, inv = _dft_matrices(n_fft, win)
    F_ = buf.shape[0]
    frames = torch.empty((F_, n_fft), dtype=torch.float32, device=buf.device)
    L.check(L.lib().maua_matmul_nt(L.ctx(buf.device), L.ptr(buf), L.ptr(inv), L.ptr(frames), F_, n_fft, 2 * buf.shape[1]))
    y = torch.empty((int(length),), dtype=torch.float32, device=buf.device)
    L.check(L.lib().maua_overlap_add(L.ctx(buf.device), L.ptr(frames), F_, n_fft, hop_length, L.ptr(win), C.c_long(n_fft // 2),
                                     C.c_long(int(length)), L.ptr(y)))
    return y


def fourier_tempo_frequencies(sr, win_length=1024, hop_length=1024, device="cpu"):
    """beat.py:24-28 (BPM of every tempogram bin; float32 linspace like the reference)."""
    rate = sr * 60 / float(hop_length)
    return torch.linspace(0, float(rate) / 2, int(1 + win_length // 2), device=device)


def fourier_tempogram(y=None, sr=22050, onset_envelope=None, hop_length=1024, win_length=1024, center=True,
                      window=torch.hann_window):
    """beat.py:31-39: STFT of the onset envelope with hop 1."""
    if onset_envelope is None:
        onset_envelope = onset_strength(y, sr, hop_length=hop_length)
    if not center:
        raise NotImplementedError("center=True only")
    win = _window_tensor(window, win_length)
    return stft_general(onset_envelope, win_length, 1, win)


def plp(y, sr, hop_length=1024, win_length=1024, tempo_min=60, tempo_max=180):
    """beat.py:42-75 predominant local pulse -> [T] in [0, 1]."""
    env = onset_strength(y, sr, hop_length=hop_length, aggregate="median")
    T = env.numel()
    W = min(T, win_length)
    if W < 4:
        raise ValueError(f"plp: {T} envelope frames are too few for a tempogram")
    ft = fourier_tempogram(onset_envelope=env, sr=sr, hop_length=hop_length, win_length=W)
    buf = _frame_major(ft)                                      # [T + 1, W/2 + 1, 2], owns its memory
    freqs = _f32(fourier_tempo_frequencies(sr, W, hop_length))
    L.check(L.lib().maua_plp_select(L.ctx(buf.device), L.ptr(buf), buf.shape[0], buf.shape[1], L.ptr(freqs),
                                    C.c_float(-1e30 if tempo_min is None else tempo_min),
                                    C.c_float(1e30 if tempo_max is None else tempo_max)))
    pulse = istft_general(torch.view_as_complex(buf).T, W, 1, T)
    mm = torch.empty((2,), dtype=torch.float32, device=pulse.device)
    L.check(L.lib().maua_minmax(L.ctx(pulse.device), L.ptr(pulse), C.c_long(T), L.ptr(mm)))
    clamped = torch.empty_like(pulse)
    L.check(L.lib().maua_clamp(L.ctx(pulse.device), L.ptr(pulse), None, L.ptr(mm[1:]), C.c_float(0.0), C.c_float(0.0),
                               C.c_long(T), L.ptr(clamped)))
    return normalize(clamped)


def pulse(audio, sr):
    """features/audio.py:72-73 -> [T, 1]."""
    return plp(percussive(audio), sr).unsqueeze(-1)


# ------------------------------------------------------------------------------------------------ tempo
def tempo_frequencies(n_bins, hop_length=512, sr=22050):
    """librosa.tempo_frequencies: BPM of autocorrelation lag i = 60 sr / (hop i); lag 0 -> inf."""
    bpm = torch.full((n_bins,), float("inf"), dtype=torch.float64)
    bpm[1:] = 60.0 * sr / (hop_length * torch.arange(1, n_bins, dtype=torch.float64))
    return bpm


def tempo(onset_envelope, sr=22050, hop_length=1024, max_tempo=240.0, ac_size=120.0, prior_scale=400.0, prior_s=1.0):
    """selfsupervised/mir.py:27-30: ``rosa.beat.tempo(onset_envelope=env, max_tempo=240, prior=lognorm(loc=0, scale=400,
    s=1), ac_size=120, hop_length=1024)`` - librosa is un-vendored and unpinned (setup.py:60), so this restates its
    published algorithm (librosa.beat.tempo / feature.tempogram, 0.9-0.10): hop-1 frames of ``ac_size`` seconds of the
    envelope (padded by half a window of linear ramp to zero), Hann-windowed autocorrelation per frame, max-normalised,
    averaged over time, log1p(1e6 x) + log-prior over the lag BPMs, lags faster than ``max_tempo`` excluded, argmax.
    Reference quirk kept (Q11): the call does not pass ``sr``, so lags are converted with librosa's default 22 050 Hz
    although the envelope's true frame rate is that of the video.  Returns a float (BPM).  Parity unpinned."""
    import math
    env = _f32(onset_envelope).reshape(-1)
    T = env.numel()
    W = int(math.floor(ac_size * sr / hop_length))          # time_to_frames
    W = max(2, min(W, 16384))
    half = W // 2
    dev = env.device
    ramp_l = torch.linspace(0, 1, half + 1, device=dev)[:-1] * env[0]       # np.pad(mode="linear_ramp", end_values=0)
    ramp_r = torch.linspace(1, 0, half + 1, device=dev)[1:] * env[-1]
    padded = torch.cat([ramp_l, env, ramp_r, torch.zeros(W, device=dev)]).contiguous()
    n = torch.arange(W, dtype=torch.float64)
    window = (0.5 - 0.5 * torch.cos(2 * math.pi * n / W)).float().to(dev)   # scipy get_window("hann", fftbins=True)
    ac = torch.empty((T, W), dtype=torch.float32, device=dev)
    L.check(L.lib().maua_autocorr_frames(L.ctx(dev), L.ptr(padded), L.ptr(window), T, W, W, L.ptr(ac)))
    tg = ac / ac.abs().amax(1, keepdim=True).clamp_min(torch.finfo(torch.float32).tiny)   # util.normalize(norm=inf)
    tg = tg.mean(0).double().cpu()
    bpms = tempo_frequencies(W, hop_length=hop_length, sr=sr)
    x = bpms.clamp_min(1e-300)
    logprior = -torch.log(x * prior_s * math.sqrt(2 * math.pi)) - torch.log(x / prior_scale) ** 2 / (2 * prior_s ** 2)
    logprior[0] = float("-inf")                                              # lognorm.logpdf(inf)
    if max_tempo is not None:
        logprior[: int(torch.argmax((bpms < max_tempo).to(torch.int8)))] = float("-inf")
    best = int(torch.argmax(torch.log1p(1e6 * tg) + logprior))
    return float(bpms[best])


# ---- torchaudio.functional.resample (selfsupervised/sample.py:7,29; audioreactive/audio.py:53-59) --------------------
def sinc_resample_kernel(orig, new, lowpass_filter_width=6, rolloff=0.99, resampling_method="sinc_interp_hann", beta=None):
    """torchaudio's polyphase filter bank [new, 2 * width + orig] for the reduced rates (float64 index arithmetic, float32
    result) - built on the host once per (rate pair): the published _get_sinc_resample_kernel (torchaudio un-vendored)."""
    import math
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None] / orig
    t = torch.arange(0, -new, -1, dtype=torch.float64)[:, None] / new + idx
    t = (t * base).clamp(-lowpass_filter_width, lowpass_filter_width)
    if resampling_method in ("sinc_interp_hann", "sinc_interpolation"):
        window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    elif resampling_method in ("sinc_interp_kaiser", "kaiser_window"):
        beta = 14.769656459379492 if beta is None else beta
        window = torch.i0(beta * torch.sqrt(1 - (t / lowpass_filter_width) ** 2)) / torch.i0(torch.tensor(float(beta), dtype=torch.float64))
    else:
        raise ValueError(f"Invalid resampling method: {resampling_method}")
    t = t * math.pi
    return (torch.where(t == 0, torch.ones_like(t), t.sin() / t) * window * (base / orig)).float(), width


def resample_sinc(waveform, orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99, resampling_method="sinc_interp_hann",
                  beta=None):
    """Drop-in for ``torchaudio.functional.resample`` on a mono signal, on the HIP device: the strided correlation with
    the filter bank is ONE exact-f32 GEMM - rows = hops of `orig` input samples (windows of 2 * width + orig), columns = the
    `new` output phases (``maua_matmul_nt``); window extraction is a strided view of the padded signal."""
    import math
    y = _f32(waveform).reshape(-1)
    if int(orig_freq) == int(new_freq):
        return y
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    kern, width = sinc_resample_kernel(orig, new, lowpass_filter_width, rolloff, resampling_method, beta)
    n = y.numel()
    xp = torch.nn.functional.pad(y, (width, width + orig))
    frames = xp.unfold(0, kern.shape[1], orig).contiguous()          # [hops, 2 * width + orig]
    out = _matmul_nt(frames, kern.to(y.device))                       # [hops, new]
    return out.reshape(-1)[: int(math.ceil(new * n / orig))].contiguous()
