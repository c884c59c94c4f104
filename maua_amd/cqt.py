"""Constant-Q features on the HIP device (SURVEY 8(f) N3 remainder): cqt, chroma_cqt, chroma_cens / ``chromagram``,
estimate_tuning / piptrack.  Drop-in for maua/audiovisual/audioreactive/selfsupervised/features/rosa/constantq.py:13-293,
rosa/pitch.py:9-123, rosa/spectral.py:164-325 and features/audio.py:44-45.

Where the work goes: the per-octave STFTs (rectangular window, hop 1024 … 16) run on the library's general Stockham
FFT, the sparse-basis product ``fft_basis @ D`` is ONE real GEMM per octave on the frame-major complex spectrum (basis
rows interleaved (re, -im) / (im, re), so re and im of the response come out interleaved and go straight into
``maua_magnitude``), the octave-to-octave down-sampling is the 28-tap polyphase FIR torchaudio's kaiser-sinc resampler
amounts to (``maua_fir_decimate``), piptrack's peak picking, the CENS quantiser (spline + smooth step) and the temporal
smoothing are kernels (``maua_piptrack``, ``maua_spline_step``, ``maua_gaussian_filter1d`` with zero padding).  Built once
per call on the host (set-up, not data path): the 36 complex filters and their FFT, the cq→chroma matrix, the spline
coefficients.  torchaudio and torchcubicspline are un-vendored: their published algorithms are restated (oracle/cqt.py has
the same restatements for the CPU side; everything the reference itself can run is pinned by tests/golden/g21_cqt.npz)."""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib as L
from .audio import _f32, _frame_major, harmonic, order_stat, stft_general

C1_HZ = 32.70319566257483  # librosa.note_to_hz("C1")


# ------------------------------------------------------------------------------------------------ host-side set-up
def note_to_hz(note):
    """convert.py:25-26 (librosa.note_to_hz: scientific pitch notation, A4 = 440 Hz) -> float32 tensor."""
    import re
    names = [note] if isinstance(note, str) else list(note)
    out = []
    for nm in names:
        m = re.match(r"^([A-Ga-g])([#b!]*)(-?\d*)(?:([+-]\d+))?$", nm.strip())
        if not m:
            raise ValueError(f"Improper note format: {nm}")
        pitch = {"C": 0, "D": 2, "E": 4, "F": 5, "G": 7, "A": 9, "B": 11}[m.group(1).upper()]
        offset = sum({"#": 1, "b": -1, "!": -1}[c] for c in m.group(2))
        octave = int(m.group(3)) if m.group(3) else 0
        cents = int(m.group(4)) * 1e-2 if m.group(4) else 0.0
        midi = 12 * (octave + 1) + pitch + offset + cents
        out.append(440.0 * 2.0 ** ((midi - 69.0) / 12.0))
    return torch.tensor(out[0] if isinstance(note, str) else out).float()


def hz_to_octs(frequencies, tuning=0.0, bins_per_octave=12):
    """convert.py:15-17"""
    return torch.log2(torch.as_tensor(frequencies) / (float(440.0 * 2.0 ** (tuning / bins_per_octave)) / 16))


def hz_to_midi(frequencies):
    """convert.py:20-21"""
    return 12 * (np.log2(frequencies) - np.log2(440.0)) + 69


def cqt_frequencies(n_bins, fmin, bins_per_octave=12, tuning=0.0):
    """constantq.py:209-212"""
    correction = 2.0 ** (float(tuning) / bins_per_octave)
    return correction * fmin * 2.0 ** (torch.arange(0, n_bins, dtype=torch.float) / bins_per_octave)


def constant_q_lengths(sr, fmin, n_bins=84, bins_per_octave=12, filter_scale=1, gamma=0):
    alpha = 2.0 ** (1.0 / bins_per_octave) - 1.0
    freq = fmin * (2.0 ** (torch.arange(n_bins, dtype=torch.float) / bins_per_octave))
    return (float(filter_scale) / alpha) * sr / (freq + gamma / alpha)


def constant_q(sr, fmin=None, n_bins=84, bins_per_octave=12, filter_scale=1, pad_fft=True, gamma=0):
    """constantq.py:223-283 (pad_fft=True): Hann-windowed complex exponentials, L1-normalised, centred in 2^k taps."""
    if not pad_fft:
        raise NotImplementedError("pad_fft=True only (what the transform calls)")
    fmin = torch.tensor(C1_HZ).float() if fmin is None else fmin
    lengths = constant_q_lengths(sr, fmin, n_bins, bins_per_octave, filter_scale, gamma)
    freqs = fmin * (2.0 ** (torch.arange(n_bins, dtype=torch.float) / bins_per_octave))
    max_len = int(2.0 ** (torch.ceil(torch.log2(max(lengths)))))
    rows = []
    for ilen, freq in zip(lengths, freqs):
        half = torch.div(ilen, 2, rounding_mode="floor")
        sig = torch.exp(torch.arange(-half, half, dtype=torch.float) * 1j * 2 * torch.pi * freq / sr)
        sig = sig * torch.hann_window(len(sig))
        sig = sig / sig.norm(p=1, dim=0)
        lpad = int((max_len - len(sig)) // 2)
        rows.append(torch.nn.functional.pad(sig, (lpad, int(max_len - len(sig) - lpad))))
    return torch.stack(rows), lengths


def cqt_filter_fft(sr, fmin, n_bins, bins_per_octave, filter_scale=1, sparsity=0.01, gamma=0.0):
    """constantq.py:142-189: FFT of the filter bank (non-negative frequencies), rows sparsified at `sparsity` of their
    magnitude mass - kept dense here, the dropped entries are zeros."""
    basis, lengths = constant_q(sr, fmin, n_bins, bins_per_octave, filter_scale, True, gamma)
    n_fft = basis.shape[1]
    fft_basis = torch.fft.fft(basis * (lengths[:, None] / float(n_fft)), n=n_fft, dim=1)[:, : n_fft // 2 + 1]
    mags = fft_basis.abs()
    mag_sort = torch.sort(mags, dim=1).values
    cumulative = torch.cumsum(mag_sort / mags.sum(1, keepdim=True), dim=1)
    thr = mag_sort[torch.arange(len(mags)), torch.argmin((cumulative < sparsity).to(torch.uint8), dim=1)]
    return torch.where(mags >= thr[:, None], fft_basis, torch.zeros_like(fft_basis)), n_fft, lengths


def _kaiser_half_band():
    """torchaudio.functional.resample(y, sr, sr / 2, resampling_method="kaiser_window") as one FIR: 28 taps, stride 2,
    13 samples of left context (lowpass_filter_width 6, rolloff 0.99, beta 14.769656459379492)."""
    lowpass_filter_width, rolloff, beta, orig, new = 6, 0.99, 14.769656459379492, 2, 1
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    t = (torch.arange(-width, width + orig, dtype=torch.float64) / orig * base).clamp(-lowpass_filter_width, lowpass_filter_width)
    window = torch.i0(beta * torch.sqrt(1 - (t / lowpass_filter_width) ** 2)) / torch.i0(torch.tensor(beta, dtype=torch.float64))
    t = t * math.pi
    taps = torch.where(t == 0, torch.ones_like(t), t.sin() / t) * window * (base / orig)
    return taps.float(), width


def resample_half(y):
    """One octave down: y[n] -> ceil(n / 2) samples, scaled by sqrt(2) (constantq.py:92-93)."""
    y = _f32(y).reshape(-1)
    taps, width = _kaiser_half_band()
    n_out = (y.numel() + 1) // 2
    out = torch.empty((n_out,), dtype=torch.float32, device=y.device)
    td = L.dev_tensor(taps, torch.float32)
    L.check(L.lib().maua_fir_decimate(L.ctx(y.device), L.ptr(y), C.c_long(y.numel()), L.ptr(td), len(taps), 2, width,
                                      C.c_float(math.sqrt(2.0)), L.ptr(out), C.c_long(n_out)))
    return out


# ------------------------------------------------------------------------------------------------ pitch.py
def piptrack(y, sr, n_fft=2048, hop_length=None, fmin=150.0, fmax=4000.0, threshold=0.1, window=torch.hann_window, center=True,
             pad_mode="reflect"):
    """pitch.py:27-87 -> (pitches, mags) [1 + n_fft/2, frames] (hop = n_fft // 4: torch.stft's default, the reference
    passes hop_length=None; last STFT column dropped like its spectrogram)."""
    if not center or pad_mode != "reflect":
        raise NotImplementedError("center=True / reflect padding only")
    from .audio import _window_tensor
    hop = n_fft // 4 if hop_length is None else int(hop_length)
    D = stft_general(y, n_fft, hop, _window_tensor(window, n_fft))[:, :-1]
    buf = _frame_major(D)
    T, nb = buf.shape[0], buf.shape[1]
    S = torch.empty((T, nb), dtype=torch.float32, device=buf.device)
    L.check(L.lib().maua_magnitude(L.ctx(buf.device), L.ptr(buf), C.c_long(S.numel()), C.c_float(1.0), L.ptr(S)))
    fmax = min(fmax, float(sr) / 2)
    freqs = L.dev_tensor(torch.linspace(0, float(sr) / 2, nb), torch.float32)
    frame_max = S.amax(1).contiguous()
    pitch, mag = torch.empty_like(S), torch.empty_like(S)
    L.check(L.lib().maua_piptrack(L.ctx(S.device), L.ptr(S), T, nb, L.ptr(frame_max), C.c_float(threshold), L.ptr(freqs),
                                  C.c_float(float(sr) / n_fft), C.c_float(max(fmin, 0)), C.c_float(fmax), L.ptr(pitch),
                                  L.ptr(mag)))
    return pitch.T, mag.T


def pitch_tuning(frequencies, resolution=0.01, bins_per_octave=12):
    """pitch.py:98-123: histogram peak of the pitch residuals (device reductions on an already-selected vector)."""
    f = torch.atleast_1d(frequencies)
    f = f[f > 0]
    if f.numel() == 0:
        return 0.0
    residual = (bins_per_octave * torch.log2(f / (440.0 / 16))) % 1.0
    residual = torch.where(residual >= 0.5, residual - 1.0, residual)
    bins = int(np.ceil(1.0 / resolution))
    counts = torch.histc(residual, bins=bins, min=-0.5, max=0.5)
    return float(torch.linspace(-0.5, 0.5, bins + 1)[int(torch.argmax(counts))])


def estimate_tuning(y, sr, n_fft=2048, resolution=0.01, bins_per_octave=12):
    """pitch.py:9-24: median magnitude over the detected pitches (exact k-th value on the device), tuning of the louder half."""
    pitch, mag = piptrack(y, sr, n_fft=n_fft)
    mask = pitch > 0
    n = int(mask.sum())
    if n == 0:
        return 0.0
    thr, _ = order_stat(mag.contiguous(), 1, k=(n + 1) // 2, mask=mask.contiguous())   # torch.median = lower middle element
    return pitch_tuning(pitch[(mag >= thr[0]) & mask], resolution, bins_per_octave)


# ------------------------------------------------------------------------------------------------ constantq.py
_BASIS_CACHE = {}


def cqt(y, sr, hop_length=1024, fmin=None, n_bins=84, bins_per_octave=12, tuning=0.0, filter_scale=1, sparsity=0.01,
        magnitude=True):
    """constantq.py:13-26: the CQT is the VQT with gamma = 0."""
    return vqt(y, sr, hop_length, fmin, n_bins, 0, bins_per_octave, tuning, filter_scale, sparsity, magnitude)


def vqt(y, sr, hop_length=1024, fmin=None, n_bins=84, gamma=None, bins_per_octave=12, tuning=0.0, filter_scale=1,
        sparsity=0.01, magnitude=True):
    """constantq.py:29-115 -> |VQT| [n_bins, frames] on the device (``magnitude=False``: complex).  ``gamma`` widens the
    low filters' bandwidth (lengths Q sr / (f + gamma / alpha)); None = the ERB default 24.7 alpha / 0.108 (:53-54)."""
    y = _f32(y).reshape(-1)
    if gamma is None:
        gamma = 24.7 * (2.0 ** (1.0 / bins_per_octave) - 1) / 0.108
    gamma = float(gamma)
    n_octaves = int(np.ceil(float(n_bins) / bins_per_octave))
    n_filters = min(bins_per_octave, n_bins)
    two = 0
    while hop_length % (2 ** (two + 1)) == 0:
        two += 1
    if hop_length <= 0 or two < n_octaves - 1:
        raise Exception(f"hop_length must be a positive integer multiple of 2^{n_octaves - 1} for {n_octaves}-octave CQT/VQT")
    fmin = torch.tensor(C1_HZ).float() if fmin is None else torch.as_tensor(fmin).float().cpu()
    if tuning is None:
        tuning = estimate_tuning(y, sr, bins_per_octave=bins_per_octave)
    fmin = fmin * 2.0 ** (tuning / bins_per_octave)
    fmin_t = torch.min(cqt_frequencies(n_bins, fmin, bins_per_octave)[-bins_per_octave:])
    lengths_full = constant_q_lengths(sr, fmin, n_bins, bins_per_octave, filter_scale, gamma)
    dev = y.device
    blocks, my_y, my_sr, my_hop = [], y, float(sr), hop_length
    for i in range(n_octaves):
        if i > 0:
            my_y, my_sr, my_hop = resample_half(my_y), my_sr / 2.0, my_hop // 2
        hi = n_bins - i * bins_per_octave                      # this octave fills bins [hi - n_oct, hi) (trim_stack)
        lo = max(0, hi - n_filters)
        # (the filter bank of an octave depends only on the rates: built on the host once per configuration, kept on the device)
        key = (float(my_sr), float(fmin_t * 2.0 ** -i), n_filters, bins_per_octave, filter_scale, sparsity, lo, hi, i,
               float(fmin), n_bins, str(dev), gamma)
        if key not in _BASIS_CACHE:
            with L.host_threads(1):   # dozens of tiny host tensors (0.29 s -> 0.04 s for the seven octaves of one CQT)
                basis, n_fft, _ = cqt_filter_fft(my_sr, fmin_t * 2.0 ** -i, n_filters, bins_per_octave, filter_scale, sparsity, gamma)
                if n_fft > 8192:
                    raise NotImplementedError(f"cqt: the filters need a {n_fft}-point FFT at sr={sr}; the HIP FFT stops at 8192")
                basis = basis[n_filters - (hi - lo):] * (np.sqrt(2 ** i) / torch.sqrt(lengths_full[lo:hi])[:, None])
                # interleaved real matrix: row 2n = (re, -im) -> Re(resp_n), row 2n+1 = (im, re) -> Im(resp_n)
                re, im = basis.real, basis.imag
                rows = torch.stack([torch.stack([re, -im], -1).reshape(len(basis), -1), torch.stack([im, re], -1).reshape(len(basis), -1)], 1)
            if len(_BASIS_CACHE) >= 64:
                _BASIS_CACHE.clear()
            _BASIS_CACHE[key] = (L.dev_tensor(rows.reshape(2 * len(basis), -1).contiguous(), torch.float32), n_fft, len(basis))
        bmat, n_fft, nb = _BASIS_CACHE[key]
        D = _frame_major(stft_general(my_y, n_fft, my_hop, window=torch.ones(n_fft))[:, :-1])   # [T, n_fft/2+1, 2]
        T = D.shape[0]
        resp = torch.empty((T, 2 * nb), dtype=torch.float32, device=dev)
        L.check(L.lib().maua_matmul_nt(L.ctx(dev), L.ptr(D), L.ptr(bmat), L.ptr(resp), T, 2 * nb, 2 * D.shape[1]))
        blocks.append((lo, hi, resp))
    T = min(r.shape[0] for _, _, r in blocks)
    out = torch.empty((T, n_bins, 2), dtype=torch.float32, device=dev)
    for lo, hi, r in blocks:
        out[:, lo:hi] = r[:T].reshape(T, hi - lo, 2)
    if not magnitude:
        return torch.view_as_complex(out).T
    mag = torch.empty((T, n_bins), dtype=torch.float32, device=dev)
    L.check(L.lib().maua_magnitude(L.ctx(dev), L.ptr(out), C.c_long(mag.numel()), C.c_float(1.0), L.ptr(mag)))
    return mag.T


def cq_to_chroma(n_input, bins_per_octave=12, n_chroma=12, fmin=None):
    """convert.py:69-118 (window=None, base_c=True) -> [n_chroma, n_input] 0/1 matrix."""
    n_merge = float(bins_per_octave) / n_chroma
    fmin = C1_HZ if fmin is None else float(fmin)
    m = torch.repeat_interleave(torch.eye(n_chroma), round(n_merge), dim=1)
    m = torch.roll(m, -int(n_merge // 2), dims=1)
    m = torch.tile(m, (1, int(np.ceil(float(n_input) / bins_per_octave))))[:, :n_input]
    midi_0 = (12 * (np.log2(np.float32(fmin)) - np.log2(440.0)) + 69) % 12
    return torch.roll(m, int(torch.round(torch.tensor(midi_0 * (n_chroma / 12.0)))), dims=0).float()


def chroma_cqt(y, sr, hop_length=1024, fmin=None, threshold=0.0, tuning=None, n_chroma=12, n_octaves=7, window=None,
               bins_per_octave=36, norm=True):
    """spectral.py:286-325 -> [n_chroma, frames]."""
    if window is not None:
        raise NotImplementedError("cq_to_chroma window")
    Cq = cqt(y, sr, hop_length, fmin, n_octaves * bins_per_octave, bins_per_octave, tuning)      # [n_bins, T]
    Ct = Cq.T.contiguous()
    M = L.dev_tensor(cq_to_chroma(Ct.shape[1], bins_per_octave, n_chroma, fmin), torch.float32)
    chroma = torch.empty((Ct.shape[0], n_chroma), dtype=torch.float32, device=Ct.device)
    L.check(L.lib().maua_matmul_nt(L.ctx(Ct.device), L.ptr(Ct), L.ptr(M), L.ptr(chroma), Ct.shape[0], n_chroma, Ct.shape[1]))
    if threshold is not None:
        chroma = torch.where(chroma < threshold, torch.zeros_like(chroma), chroma)
    if norm:
        chroma = chroma / chroma.max()
    return chroma.T


_QUANT = {}


def _quantiser():
    """knots + (a, b, 2c, 3d) rows of torchcubicspline.natural_cubic_spline_coeffs on the knots of spectral.py:164-190 -
    the reference's spline_eval combines them as a + f (b + f (2c + f 3d)) (reference quirk Q12, kept)."""
    if not _QUANT:
        from scipy.interpolate import CubicSpline
        steps = [0.4, 0.2, 0.1, 0.05]
        p1, p2, p3, p4 = np.diff(list(reversed(steps + [0])))
        xs = [torch.linspace(-0.1, 0.025, 101)[:-1], torch.linspace(0.025, p1, 11)[:-1], torch.linspace(p1, p1 + p2, 11)[:-1],
              torch.linspace(p1 + p2, p1 + p2 + p3, 11)[:-1], torch.linspace(p1 + p2 + p3, 0.5, 11)[:-1],
              torch.linspace(0.5, 1.1, 100)]
        ys = torch.cat((0.5 * torch.ones(len(xs[0])), xs[1] / p1, (xs[2] - p1) / p2 + 1, (xs[3] - p1 - p2) / p3 + 2,
                        (xs[4] - p1 - p2 - p3) / p4 + 3, 4.5 * torch.ones(len(xs[5]))))
        xs = torch.cat(xs)
        d, c, b, a = CubicSpline(xs.double().numpy(), ys.double().numpy(), bc_type="natural").c
        _QUANT["xs"] = xs.float()
        _QUANT["coef"] = torch.from_numpy(np.stack([a, b, 2 * c, 3 * d]).astype(np.float32))
    return _QUANT["xs"], _QUANT["coef"]


def spline_quantize(chroma, h=0.25, alpha=20):
    """spectral.py:193-232 (spline_eval + step_function) elementwise on the device."""
    x = _f32(chroma).contiguous()
    xs, coef = _quantiser()
    xs_d, coef_d = L.dev_tensor(xs, torch.float32), L.dev_tensor(coef.contiguous(), torch.float32)
    out = torch.empty_like(x)
    L.check(L.lib().maua_spline_step(L.ctx(x.device), L.ptr(x), C.c_long(x.numel()), L.ptr(xs_d), L.ptr(coef_d), len(xs),
                                     C.c_float(h), C.c_float(alpha), 1, L.ptr(out)))
    return out


def chroma_cens(y, sr, hop_length=1024, fmin=None, tuning=None, n_chroma=12, n_octaves=7, bins_per_octave=36, window=None,
                win_len_smooth=41):
    """spectral.py:239-283 -> [12, frames]: L1-normalised chroma -> soft quantiser -> Hann smoothing (zero "same"
    padding) -> L2 normalisation."""
    chroma = chroma_cqt(y, sr, hop_length, fmin, 0.0, tuning, n_chroma, n_octaves, window, bins_per_octave, norm=False)
    ct = chroma.T.contiguous()                                   # [T, 12]
    ct = ct / ct.abs().sum(1, keepdim=True)
    q = spline_quantize(ct)
    if win_len_smooth:
        win = torch.hann_window(win_len_smooth + 2)
        win = L.dev_tensor(win / win.sum(), torch.float32)
        sm = torch.empty_like(q)
        L.check(L.lib().maua_gaussian_filter1d(L.ctx(q.device), L.ptr(q), L.ptr(win), (len(win) - 1) // 2, q.shape[0],
                                               C.c_long(q.shape[1]), 3, L.ptr(sm)))
        q = sm
    return (q / q.norm(p=2, dim=1, keepdim=True)).T


def chromagram(audio, sr):
    """features/audio.py:44-45 -> [T, 12]."""
    return chroma_cens(harmonic(audio), sr).T
