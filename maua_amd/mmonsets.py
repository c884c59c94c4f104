"""The madmom branch of the classic onset envelope on the HIP device (drop-in for the type="mm" path of
maua/audiovisual/audioreactive/mir.py:35-56 - the reference's DEFAULT): FramedSignal(2048, hop 512) ->
ShortTimeFourierTransform(circular_shift=True) -> Spectrogram -> FilteredSpectrogram(num_bands=24) -> mean of
spectral_diff, spectral_flux, superflux, complex_flux and modified_kullback_leibler, each divided by its maximum.

madmom is un-vendored (setup.py lists it unversioned): this restates its published chain (madmom 0.16 audio/{signal,stft,
spectrogram,filters}.py, features/onsets.py; Boeck & Widmer DAFx 2013 / ISMIR 2013) - see oracle/mmonsets.py for the
step list.  PARITY UNPINNED.  Device work: the general-hop STFT kernel (maua_stft_general) on the zero-padded signal,
the filterbank product (maua_matmul_nt), element-wise differences / 3-wide maxima / wrapped phase differences / band
minima as device tensor operations; the 1024 x ~180 filterbank is built once on the host."""
import functools
import math

import numpy as np
import torch

from . import _lib as L
from . import audio as A

EPS = float(np.spacing(1.0))


@functools.lru_cache(maxsize=8)
def log_filterbank(sr, fft_size=2048, num_bands=24, fmin=30.0, fmax=17000.0, fref=440.0):
    """madmom LogarithmicFilterbank(fft_frequencies, num_bands, fmin, fmax, fref, norm_filters=True, unique_filters=True)
    -> (float32 [fft_size / 2, n_filters], first bins, last bins of the filters' supports)."""
    n_bins = fft_size // 2
    bin_f = np.arange(n_bins) * (sr / float(fft_size))
    left = math.floor(math.log2(fmin / fref) * num_bands)
    right = math.ceil(math.log2(fmax / fref) * num_bands)
    f = fref * 2.0 ** (np.arange(left, right) / float(num_bands))
    f = f[np.searchsorted(f, fmin):]
    f = f[:np.searchsorted(f, fmax, "right")]
    idx = np.clip(bin_f.searchsorted(f), 1, n_bins - 1)
    idx = np.unique(idx - (f - bin_f[idx - 1] < bin_f[idx] - f))         # the closer of the two neighbouring bins
    fb = np.zeros((n_bins, len(idx) - 2), dtype=np.float32)
    lo, hi = [], []
    for b in range(len(idx) - 2):
        start, center, stop = int(idx[b]), int(idx[b + 1]), int(idx[b + 2])
        if stop - start < 2:
            center, stop = start, start + 1
        tri = np.concatenate([np.linspace(0, 1, center - start, endpoint=False),
                              np.linspace(1, 0, stop - center, endpoint=False)]).astype(np.float32)
        fb[start:stop, b] = tri / tri.sum()
        nz = np.nonzero(fb[:, b])[0]
        lo.append(int(nz[0]))
        hi.append(int(nz[-1]))
    return fb, tuple(lo), tuple(hi)


def _max3(x, dim):
    """scipy.ndimage.maximum_filter(size 3) along ``dim`` (its 'reflect' border repeats the edge sample)."""
    x = x.movedim(dim, -1)
    shp = x.shape
    p = torch.nn.functional.pad(x.reshape(-1, 1, shp[-1]), (1, 1), mode="replicate")
    return torch.nn.functional.max_pool1d(p, 3, 1).reshape(shp).movedim(-1, dim)


@L.host_threads(1)
def onset_functions(audio, sr, frame_size=2048, hop=512, num_bands=24):
    """-> dict of the five onset detection functions, float32 [n_frames] on the device, n_frames = ceil(len / hop)."""
    y = A._f32(audio).reshape(-1)
    n = int(math.ceil(y.numel() / float(hop)))
    if n == 0:
        z = torch.zeros(0, dtype=torch.float32, device=y.device)
        return {k: z.clone() for k in ("spectral_diff", "spectral_flux", "superflux", "complex_flux", "modified_kullback_leibler")}
    half = frame_size // 2
    if half % hop:
        raise NotImplementedError("half a frame must be a whole number of hops")
    lead = half // hop                                   # frame j = n' + lead of the padded signal is centred on n' hop
    padded = torch.zeros((n + lead) * hop + half + hop, dtype=torch.float32, device=y.device)
    padded[half:half + y.numel()] = y                    # (zeros either side: the kernel's reflect framing never sees a border)
    k = torch.arange(frame_size, dtype=torch.float64)
    window = (0.5 - 0.5 * torch.cos(2 * math.pi * k / (frame_size - 1))).float()      # np.hanning: symmetric
    spec = A.stft_general(padded, frame_size, hop, window=window)                      # [frame_size / 2 + 1, frames]
    X = spec[:half, lead:lead + n].T                                                   # [n, 1024]
    sign = torch.ones(half, device=y.device)
    sign[1::2] = -1.0                                                                  # circular shift by half a frame
    X = X * sign
    mag = X.abs().contiguous()
    fb, lo, hi = log_filterbank(int(sr), frame_size, num_bands)
    fbt = L.dev_tensor(torch.from_numpy(np.ascontiguousarray(fb.T)), torch.float32)    # [n_filters, 1024]
    nb = fbt.shape[0]
    S = torch.empty((n, nb), dtype=torch.float32, device=y.device)
    L.check(L.lib().maua_matmul_nt(L.ctx(y.device), L.ptr(mag), L.ptr(fbt), L.ptr(S), n, nb, half))
    d = torch.zeros_like(S)
    d[1:] = (S[1:] - S[:-1]).clamp_min(0)
    dm = torch.zeros_like(S)
    dm[1:] = (S[1:] - _max3(S, 1)[:-1]).clamp_min(0)
    phase = torch.angle(X)
    dp = phase[:, 1:] - phase[:, :-1]
    wrapped = torch.remainder(dp + math.pi, 2 * math.pi) - math.pi                     # np.unwrap's difference
    lgd = torch.zeros_like(phase)
    lgd[:, :-1] = wrapped.abs() / math.pi
    lgd = _max3(lgd, 0)
    # per band: minimum of the local group delay over the filter's bins and one neighbour either side
    wmax = max(h - l for l, h in zip(lo, hi)) + 3
    starts = torch.tensor([max(l - 1, 0) for l in lo])
    stops = torch.tensor([min(h + 2, half) for h in hi])
    gidx = torch.minimum(starts[:, None] + torch.arange(wmax)[None, :], stops[:, None] - 1).to(y.device)   # [nb, wmax]
    mask = torch.empty_like(S)
    for c in range(0, n, 512):
        mask[c:c + 512] = lgd[c:c + 512][:, gidx].amin(-1)
    mkl = torch.zeros_like(S)
    mkl[1:] = S[1:] / (S[:-1] + EPS)
    return {"spectral_diff": (d * d).sum(1), "spectral_flux": d.sum(1), "superflux": dm.sum(1),
            "complex_flux": (dm * mask).sum(1), "modified_kullback_leibler": torch.log1p(mkl).mean(1)}


def mm_onset_envelope(audio, sr):
    """mir.py:48-56: mean of the five functions, each divided by its maximum (percentile_clip(95) is the caller's)."""
    f = onset_functions(audio, sr)
    if next(iter(f.values())).numel() == 0:
        return next(iter(f.values()))
    return torch.stack([v / v.max() for v in f.values()]).mean(0)
