"""Classic envelope post-processing on the HIP device (drop-in for maua/audiovisual/audioreactive/signal.py:
resample :5-24, normalize :27-38, percentile :41-52, percentile_clip :55-81, gaussian_filter :108-157)."""
import ctypes as C

import numpy as np
import torch

from . import _lib as L
from . import audio as A


def resample(x, size):
    x = A._f32(x)
    y = x.squeeze()
    if y.ndim == 0:
        y = y[None]
    n = y.shape[0]
    Cn = y.numel() // n
    out = torch.empty((size, *y.shape[1:]), dtype=torch.float32, device=y.device)
    L.check(L.lib().maua_resample_linear(L.ctx(y.device), L.ptr(y.contiguous()), n, C.c_long(Cn), int(size), L.ptr(out)))
    return out.squeeze()


def normalize(x):
    return A.normalize(x, eps=0.0)


def percentile(signal, p):
    s = A._f32(signal).reshape(-1)
    k = 1 + round(0.01 * float(p) * (s.numel() - 1))
    return A.order_stat(s, 1, k=k)[0][0].item()


def percentile_clip(signal, percent):
    sig = A._f32(signal)
    if sig.ndim < 2:
        sig = sig.unsqueeze(1)
    cols = []
    for col in sig.unbind(1):
        col = col.contiguous()
        n = col.numel()
        mask = torch.empty((n,), dtype=torch.uint8, device=col.device)
        L.check(L.lib().maua_peak_mask(L.ctx(col.device), L.ptr(col), n, L.ptr(mask)))
        n_peaks = int(mask.sum().item())  # host needs the count for the reference's k = 1 + round(p% * (n-1))
        k = 1 + round(0.01 * float(percent) * (n_peaks - 1))
        hi, _ = A.order_stat(col, 1, k=k, mask=mask)
        y = torch.empty_like(col)
        L.check(L.lib().maua_clamp(L.ctx(col.device), L.ptr(col), None, L.ptr(hi), C.c_float(0.0), C.c_float(0.0),
                                   C.c_long(n), L.ptr(y)))
        cols.append(y / y.max())
    return torch.stack(cols, dim=1)


def gaussian_filter(x, sigma, causal=None, mode="circular"):
    return A.gaussian_filter(x, sigma, mode=mode, causal=causal, _classic=True)


def compress(signal, threshold, ratio, invert=False):
    """signal.py:84-100: values above (invert: below) ``threshold`` are multiplied by ``ratio``, then normalize.
    (The reference scales its argument in place; this returns a new tensor.)"""
    x = A._f32(signal)
    y = torch.empty_like(x)
    L.check(L.lib().maua_threshold_scale(L.ctx(x.device), L.ptr(x), C.c_long(x.numel()), C.c_float(float(threshold)),
                                         C.c_float(float(ratio)), int(bool(invert)), L.ptr(y)))
    return normalize(y)


def expand(signal, threshold, ratio, invert=False):
    """signal.py:103-105: alias of compress"""
    return compress(signal, threshold, ratio, invert)


def sosfilt(sos, x):
    """scipy.signal.sosfilt(sos, x) along the last axis on the device: float64, zero initial state (maua_sosfilt - every section
    is scipy's direct form II transposed, sample for sample, over 128-sample chunks whose start states come from a scan).
    ``sos`` [n_sections, 6] is a host array (36 numbers at most on this path).  A numpy signal comes back as numpy, like scipy's;
    a tensor comes back as a float64 device tensor."""
    sos = np.ascontiguousarray(np.asarray(sos, dtype=np.float64))
    if sos.ndim != 2 or sos.shape[1] != 6:
        raise ValueError("sos must have shape [n_sections, 6]")
    as_numpy = not isinstance(x, torch.Tensor)
    xd = L.dev_tensor(torch.as_tensor(np.asarray(x)) if as_numpy else x.detach(), torch.float64)
    y = torch.empty_like(xd)
    n = xd.shape[-1] if xd.ndim else 0
    rows_in, rows_out = xd.reshape(-1, n) if n else xd.reshape(0, 0), y.reshape(-1, n) if n else y.reshape(0, 0)
    for r in range(rows_in.shape[0]):
        L.check(L.lib().maua_sosfilt(L.ctx(xd.device), sos.ctypes.data_as(C.c_void_p), sos.shape[0], L.ptr(rows_in[r]),
                                     C.c_long(n), L.ptr(rows_out[r])))
    return y.cpu().numpy() if as_numpy else y
