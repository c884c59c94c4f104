"""Oracle restatement of the constant-Q features (test infrastructure only; plain PyTorch-CPU / numpy).

Follows maua/audiovisual/audioreactive/selfsupervised/features/rosa/constantq.py:13-293 (cqt / vqt and helpers),
rosa/pitch.py:9-123 (estimate_tuning, piptrack, pitch_tuning), rosa/convert.py:60-118 (cq_to_chroma, hz_to_octs),
rosa/spectral.py:164-325 (spline quantiser, chroma_cens, chroma_cqt) and features/audio.py:44-45 (chromagram).

Pinned by tests/golden/g21_cqt.npz for everything the reference can run here (filter bank, sparsified FFT basis, top-octave
response, piptrack / tuning, cq_to_chroma).  PARITY UNPINNED for the two un-vendored pieces, restated from their published
code: torchaudio.functional.resample (kaiser-windowed sinc, lowpass_filter_width 6, rolloff 0.99, beta 14.769656459379492)
and torchcubicspline.natural_cubic_spline_coeffs (returns (t, a, b, 2c, 3d) of the natural cubic spline; note that the
reference's own spline_eval uses 2c and 3d as if they were c and d - restated as it is, quirk Q12)."""
import math

import numpy as np
import torch
import torch.nn.functional as F

C1_HZ = 32.70319566257483  # librosa.note_to_hz("C1") (A4 = 440 Hz, MIDI note 24)


# ---------------------------------------------------------------------------------------------- torchaudio resample
def sinc_resample_kernel(orig, new, lowpass_filter_width=6, rolloff=0.99, beta=14.769656459379492):
    """torchaudio.functional._get_sinc_resample_kernel for resampling_method="kaiser_window" -> (kernel [new, width*2+orig], width)."""
    g = math.gcd(int(orig), int(new))
    orig, new = int(orig) // g, int(new) // g
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None] / orig
    t = torch.arange(0, -new, -1, dtype=torch.float64)[:, None] / new + idx
    t = (t * base).clamp(-lowpass_filter_width, lowpass_filter_width)
    window = torch.i0(beta * torch.sqrt(1 - (t / lowpass_filter_width) ** 2)) / torch.i0(torch.tensor(beta, dtype=torch.float64))
    t = t * math.pi
    kernel = torch.where(t == 0, torch.ones_like(t), t.sin() / t) * window * (base / orig)
    return kernel.float(), width, orig, new


def resample(y, orig, new):
    """torchaudio.functional.resample(y, orig, new, resampling_method="kaiser_window") on a 1-D signal."""
    kernel, width, o, n = sinc_resample_kernel(orig, new)
    x = F.pad(y[None, None], (width, width + o))
    out = F.conv1d(x, kernel[:, None], stride=o).transpose(1, 2).reshape(-1)
    return out[: math.ceil(n * len(y) / o)]


# ---------------------------------------------------------------------------------------------- constantq.py
def cqt_frequencies(n_bins, fmin, bins_per_octave=12):
    return fmin * 2.0 ** (torch.arange(0, n_bins, dtype=torch.float) / bins_per_octave)


def constant_q_lengths(sr, fmin, n_bins=84, bins_per_octave=12, filter_scale=1, gamma=0):
    alpha = 2.0 ** (1.0 / bins_per_octave) - 1.0
    Q = float(filter_scale) / alpha
    freq = fmin * (2.0 ** (torch.arange(n_bins, dtype=torch.float) / bins_per_octave))
    return Q * sr / (freq + gamma / alpha)


def constant_q(sr, fmin, n_bins, bins_per_octave, filter_scale=1, gamma=0):
    """constantq.py:237-283 with pad_fft=True -> (filters [n_bins, 2^k] complex64, lengths [n_bins])."""
    lengths = constant_q_lengths(sr, fmin, n_bins, bins_per_octave, filter_scale, gamma)
    freqs = fmin * (2.0 ** (torch.arange(n_bins, dtype=torch.float) / bins_per_octave))
    filters = []
    for ilen, freq in zip(lengths, freqs):
        ilen2 = torch.div(ilen, 2, rounding_mode="floor")
        sig = torch.exp(torch.arange(-ilen2, ilen2, dtype=torch.float) * 1j * 2 * torch.pi * freq / sr)
        sig = sig * torch.hann_window(len(sig))
        filters.append(sig / sig.norm(p=1, dim=0))
    max_len = int(2.0 ** (torch.ceil(torch.log2(max(lengths)))))
    out = []
    for f in filters:
        lpad = int((max_len - len(f)) // 2)
        out.append(F.pad(f, (lpad, int(max_len - len(f) - lpad))))
    return torch.stack(out), lengths


def sparsify_rows(x, quantile=0.01):
    """constantq.py:170-189 as a dense matrix (entries below the row's cumulative-magnitude threshold zeroed)."""
    mags = x.abs()
    norms = mags.sum(1, keepdim=True)
    mag_sort = torch.sort(mags, dim=1).values
    cumulative = torch.cumsum(mag_sort / norms, dim=1)
    thr_idx = torch.argmin((cumulative < quantile).to(torch.uint8), dim=1)
    thr = mag_sort[torch.arange(len(x)), thr_idx]
    return torch.where(mags >= thr[:, None], x, torch.zeros_like(x))


def cqt_filter_fft(sr, fmin, n_bins, bins_per_octave, filter_scale=1, sparsity=0.01, gamma=0.0):
    basis, lengths = constant_q(sr, fmin, n_bins, bins_per_octave, filter_scale, gamma)
    n_fft = basis.shape[1]
    basis = basis * (lengths[:, None] / float(n_fft))
    fft_basis = torch.fft.fft(basis, n=n_fft, dim=1)[:, : n_fft // 2 + 1]
    return sparsify_rows(fft_basis, sparsity), n_fft, lengths


def stft_rect(y, n_fft, hop):
    """rosa/spectral.py:10-21 with window=None (rectangular), centre / reflect."""
    return torch.stft(y, n_fft=n_fft, hop_length=hop, center=True, window=None, pad_mode="reflect", return_complex=True)


def cqt(y, sr, hop_length=1024, fmin=None, n_bins=84, bins_per_octave=12, tuning=0.0, filter_scale=1, sparsity=0.01):
    """constantq.py:13-26 (vqt with gamma = 0)."""
    return vqt(y, sr, hop_length, fmin, n_bins, 0, bins_per_octave, tuning, filter_scale, sparsity)


def vqt(y, sr, hop_length=1024, fmin=None, n_bins=84, gamma=None, bins_per_octave=12, tuning=0.0, filter_scale=1, sparsity=0.01):
    """constantq.py:29-115."""
    if gamma is None:
        gamma = 24.7 * (2.0 ** (1.0 / bins_per_octave) - 1) / 0.108
    n_octaves = int(np.ceil(float(n_bins) / bins_per_octave))
    n_filters = min(bins_per_octave, n_bins)
    fmin = torch.tensor(C1_HZ).float() if fmin is None else torch.as_tensor(fmin).float()
    if tuning is None:
        tuning = estimate_tuning(y, sr, bins_per_octave=bins_per_octave)
    fmin = fmin * 2.0 ** (tuning / bins_per_octave)
    freqs = cqt_frequencies(n_bins, fmin, bins_per_octave)[-bins_per_octave:]
    fmin_t = torch.min(freqs)
    resp = []
    my_y, my_sr, my_hop = y, sr, hop_length
    for i in range(n_octaves):
        if i > 0:
            my_y = resample(my_y, my_sr, my_sr / 2) * np.sqrt(2)
            my_sr /= 2.0
            my_hop //= 2
        fft_basis, n_fft, _ = cqt_filter_fft(my_sr, fmin_t * 2.0 ** -i, n_filters, bins_per_octave, filter_scale, sparsity, gamma)
        fft_basis = fft_basis * np.sqrt(2 ** i)
        resp.append(fft_basis @ stft_rect(my_y, n_fft, my_hop)[:, :-1])
    max_col = min(c.shape[-1] for c in resp)
    out = torch.empty((n_bins, max_col), dtype=resp[0].dtype)
    end = n_bins
    for c in resp:
        n_oct = c.shape[0]
        if end < n_oct:
            out[:end] = c[-end:, :max_col]
        else:
            out[end - n_oct:end] = c[:, :max_col]
        end -= n_oct
    lengths = constant_q_lengths(sr, fmin, n_bins, bins_per_octave, filter_scale, gamma)
    return out / torch.sqrt(lengths[:, None])


# ---------------------------------------------------------------------------------------------- pitch.py
def hz_to_octs(frequencies, tuning=0.0, bins_per_octave=12):
    return torch.log2(frequencies / (float(440.0 * 2.0 ** (tuning / bins_per_octave)) / 16))


def localmax(x):
    xp = F.pad(x, (0, 0, 1, 1))
    return (x > xp[:-2]) & (x >= xp[2:])


def piptrack(y, sr, n_fft=2048, fmin=150.0, fmax=4000.0, threshold=0.1):
    """pitch.py:27-87; hop_length=None = torch.stft's default n_fft // 4, last STFT column dropped (spectral.py:59-62)."""
    S = torch.stft(y, n_fft=n_fft, hop_length=None, center=True, window=torch.hann_window(n_fft), pad_mode="reflect",
                   return_complex=True)[:, :-1].abs()
    fmax = min(fmax, float(sr) / 2)
    fft_freqs = torch.linspace(0, float(sr) / 2, int(1 + n_fft // 2))
    avg = 0.5 * (S[2:] - S[:-2])
    shift = 2 * S[1:-1] - S[2:] - S[:-2]
    shift = avg / (shift + (shift.abs() < torch.finfo(shift.dtype).tiny))
    avg = F.pad(avg, (0, 0, 1, 1))
    shift = F.pad(shift, (0, 0, 1, 1))
    dskew = 0.5 * avg * shift
    pitches, mags = torch.zeros_like(S), torch.zeros_like(S)
    freq_mask = ((fmin <= fft_freqs) & (fft_freqs < fmax)).reshape(-1, 1)
    ref_value = threshold * S.max(0).values
    idx = torch.argwhere(freq_mask & localmax(S * (S > ref_value)))
    pitches[idx[:, 0], idx[:, 1]] = (idx[:, 0] + shift[idx[:, 0], idx[:, 1]]) * float(sr) / n_fft
    mags[idx[:, 0], idx[:, 1]] = S[idx[:, 0], idx[:, 1]] + dskew[idx[:, 0], idx[:, 1]]
    return pitches, mags


def pitch_tuning(frequencies, resolution=0.01, bins_per_octave=12):
    frequencies = torch.atleast_1d(frequencies)
    frequencies = frequencies[frequencies > 0]
    if not torch.any(frequencies):
        return 0.0
    residual = (bins_per_octave * hz_to_octs(frequencies)) % 1.0
    residual[residual >= 0.5] -= 1.0
    bins = int(np.ceil(1.0 / resolution))
    counts = torch.histc(residual, bins=bins, min=-0.5, max=0.5)
    tuning = torch.linspace(-0.5, 0.5, bins + 1)
    return tuning[torch.argmax(counts)]


def estimate_tuning(y, sr, n_fft=2048, resolution=0.01, bins_per_octave=12):
    pitch, mag = piptrack(y, sr, n_fft=n_fft)
    mask = pitch > 0
    threshold = torch.median(mag[mask]) if mask.any() else 0.0
    return pitch_tuning(pitch[(mag >= threshold) & mask], resolution, bins_per_octave)


# ---------------------------------------------------------------------------------------------- chroma
def cq_to_chroma(n_input, bins_per_octave=12, n_chroma=12, fmin=None):
    """convert.py:69-118 (window=None, base_c=True)."""
    n_merge = float(bins_per_octave) / n_chroma
    fmin = C1_HZ if fmin is None else float(fmin)
    m = torch.repeat_interleave(torch.eye(n_chroma), round(n_merge), dim=1)
    m = torch.roll(m, -int(n_merge // 2), dims=1)
    n_octaves = np.ceil(float(n_input) / bins_per_octave)
    m = torch.tile(m, (1, int(n_octaves)))[:, :n_input]
    midi_0 = (12 * (np.log2(np.float32(fmin)) - np.log2(440.0)) + 69) % 12
    roll = int(torch.round(torch.tensor(midi_0 * (n_chroma / 12.0))))
    return torch.roll(m, roll, dims=0).float()


def quantiser_knots():
    """spectral.py:164-190: knots / values of the spline behind spline_quantize."""
    steps = [0.4, 0.2, 0.1, 0.05]
    p1, p2, p3, p4 = np.diff(list(reversed(steps + [0])))
    xs = [torch.linspace(-0.1, 0.025, 101)[:-1], torch.linspace(0.025, p1, 11)[:-1], torch.linspace(p1, p1 + p2, 11)[:-1],
          torch.linspace(p1 + p2, p1 + p2 + p3, 11)[:-1], torch.linspace(p1 + p2 + p3, 0.5, 11)[:-1],
          torch.linspace(0.5, 1.1, 100)]
    ys = torch.cat((0.5 * torch.ones(len(xs[0])), xs[1] / p1, (xs[2] - p1) / p2 + 1, (xs[3] - p1 - p2) / p3 + 2,
                    (xs[4] - p1 - p2 - p3) / p4 + 3, 4.5 * torch.ones(len(xs[5]))))
    return torch.cat(xs), ys


def quantiser_coeffs():
    """torchcubicspline.natural_cubic_spline_coeffs(xs, ys) -> knots and the (a, b, 2c, 3d) rows the reference's
    spline_eval combines as a + f (b + f (2c + f 3d))  (quirk Q12: it treats 2c, 3d as c, d)."""
    from scipy.interpolate import CubicSpline
    xs, ys = quantiser_knots()
    cs = CubicSpline(xs.double().numpy(), ys.double().numpy(), bc_type="natural")
    d, c, b, a = cs.c  # per interval, descending powers
    return xs, np.stack([a, b, 2 * c, 3 * d]).astype(np.float32)


def spline_quantize(chroma, h=0.25, alpha=20):
    xs, coef = quantiser_coeffs()
    coef = torch.from_numpy(coef)
    idx = (torch.bucketize(chroma, xs) - 1).clamp(0, len(xs) - 2)
    f = chroma - xs[idx]
    w = coef[0][idx] + (coef[1][idx] + (coef[2][idx] + coef[3][idx] * f) * f) * f
    r = (w - 0.5) - torch.floor(w - 0.5) - 0.5
    m = 1 / (1 + np.exp(-alpha)) - 0.5
    return h * (torch.floor(w - 0.5) + 1 / (2 * m) * 1 / (1 + torch.exp(-2 * alpha * r)))


def chroma_cqt(y, sr, hop_length=1024, tuning=None, n_chroma=12, n_octaves=7, bins_per_octave=36, norm=True):
    C = cqt(y, sr, hop_length, None, n_octaves * bins_per_octave, bins_per_octave, tuning).abs()
    chroma = cq_to_chroma(C.shape[0], bins_per_octave, n_chroma) @ C
    chroma[chroma < 0.0] = 0.0
    return chroma / chroma.max() if norm else chroma


def chroma_cens(y, sr, hop_length=1024, win_len_smooth=41):
    chroma = chroma_cqt(y, sr, hop_length, norm=False)
    chroma = chroma / torch.norm(chroma, p=1, dim=0)
    q = spline_quantize(chroma)
    win = torch.hann_window(win_len_smooth + 2)
    win = win / win.sum()
    cens = F.conv1d(q[None], win.tile(12, 1, 1), groups=12, padding="same")[0]
    return cens / torch.norm(cens, p=2, dim=0)


def chromagram(audio, sr):
    """features/audio.py:44-45 -> [T, 12]."""
    from .audio import harmonic
    return chroma_cens(harmonic(audio), sr).T
