"""Oracle restatement of the classic envelope post-processing (test infrastructure only).
Follows maua/audiovisual/audioreactive/signal.py: resample :5-24, normalize :27-38, percentile :41-52,
percentile_clip :55-81, gaussian_filter :108-157."""
import torch
import torch.nn.functional as F

from .audio import gaussian_filter as _gf


def resample(x, size):
    """signal.py:5-24 — linear interpolation (align_corners=False) along time."""
    y = x.squeeze()
    if y.ndim == 1:
        y = y[None, None]
    elif y.ndim == 2:
        y = y.permute(1, 0)[None]
    elif y.ndim == 3:
        y = y.permute(1, 2, 0)
    out = F.interpolate(y, size=size, mode="linear", align_corners=False)
    return out.permute(2, 0, 1).squeeze()


def normalize(x):
    """signal.py:27-38"""
    y = x - x.min()
    return y / y.max()


def percentile_index(n, p):
    """signal.py:51 — 1-based k for kthvalue; Python round = half-to-even."""
    return 1 + round(0.01 * float(p) * (n - 1))


def percentile(sig, p):
    """signal.py:41-52"""
    k = percentile_index(sig.numel(), p)
    return sig.reshape(-1).kthvalue(k).values.item()


def peak_mask(sig):
    """signal.py:69-76 — strictly greater than both neighbours; edge neighbours clamp to self."""
    n = sig.shape[0]
    locs = torch.arange(n)
    plus = sig[(locs + 1).clamp(0, n - 1)]
    minus = sig[(locs - 1).clamp(0, n - 1)]
    return (sig > plus) & (sig > minus)


def percentile_clip(signal, percent):
    """signal.py:55-81"""
    if signal.ndim < 2:
        signal = signal.unsqueeze(1)
    out = []
    for sig in signal.unbind(1):
        sig = sig.clamp(0, percentile(sig[peak_mask(sig)], percent))
        out.append(sig / sig.max())
    return torch.stack(out, dim=1)


def gaussian_filter(x, sigma, causal=None, mode="circular"):
    """signal.py:108-157"""
    return _gf(x, sigma, mode=mode, causal=causal, classic=True)
