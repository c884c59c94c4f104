"""Oracle restatement of the StyleGAN2 operator layer (test infrastructure only).

Follows maua/GAN/wrappers/inference/ops.py of the reference; every function
cites the lines it restates.  Scalars are plain Python numbers here (SURVEY.md
Q1: the reference only runs with 0-dim tensors; the intent is scalar).
All arithmetic is PyTorch-CPU in the dtype of ``x`` (fp32 in the tests).
"""
from math import sqrt

import torch
import torch.nn.functional as F

_ACT_DEFAULTS = {  # ops.py:23-41  (alpha, gain)
    "linear": (0.0, 1.0),
    "relu": (0.0, sqrt(2)),
    "lrelu": (0.2, sqrt(2)),
    "tanh": (0.0, 1.0),
    "sigmoid": (0.0, 1.0),
    "elu": (0.0, 1.0),
    "selu": (0.0, 1.0),
    "softplus": (0.0, 1.0),
    "swish": (0.0, sqrt(2)),
}


def _activate(x, act, alpha):
    """ops.py:44-62"""
    if act == "linear":
        return x
    if act == "relu":
        return F.relu(x)
    if act == "lrelu":
        return F.leaky_relu(x, alpha)
    if act == "tanh":
        return torch.tanh(x)
    if act == "sigmoid":
        return torch.sigmoid(x)
    if act == "elu":
        return F.elu(x)
    if act == "selu":
        return F.selu(x)
    if act == "softplus":
        return F.softplus(x)
    if act == "swish":
        return torch.sigmoid(x) * x
    raise ValueError(act)


def bias_act(x, b=None, act="linear", alpha=None, gain=None, clamp=None):
    """ops.py:65-84 — add per-channel bias (dim 1), activate, scale, clamp."""
    def_alpha, def_gain = _ACT_DEFAULTS[act]
    alpha = def_alpha if alpha is None else float(alpha)
    gain = def_gain if gain is None else float(gain)
    clamp = -1.0 if clamp is None else float(clamp)
    if b is not None:
        shape = [1] * x.ndim
        shape[1] = -1
        x = x + b.reshape(shape).to(x.dtype)
    x = _activate(x, act, alpha)
    if gain != 1:
        x = x * gain
    if clamp >= 0:
        x = x.clamp(-clamp, clamp)
    return x


def setup_filter(f=(1, 3, 3, 1), normalize=True, gain=1.0):
    """ops.py:236-256 — 1-D taps with fewer than 8 entries become the outer product."""
    f = torch.as_tensor(f, dtype=torch.float32)
    if f.ndim == 0:
        f = f[None]
    separable = f.ndim == 1 and f.numel() >= 8
    if f.ndim == 1 and not separable:
        f = torch.outer(f, f)
    if normalize:
        f = f / f.sum()
    return f * (gain ** (f.ndim / 2))


def upfirdn2d(x, f, up=1, down=1, padding=(0, 0, 0, 0), gain=1.0):
    """ops.py:87-114 — zero-insert upsample, pad/crop, depthwise *correlation*
    with f (no flip), decimate.  padding = (px0, px1, py0, py1)."""
    if f is None:
        f = torch.ones([1, 1], dtype=torch.float32)
    n, c, h, w = x.shape
    px0, px1, py0, py1 = [int(p) for p in padding]
    if up > 1:
        z = x.new_zeros(n, c, h * up, w * up)
        z[:, :, ::up, ::up] = x
        x = z
    x = F.pad(x, [max(px0, 0), max(px1, 0), max(py0, 0), max(py1, 0)])
    x = x[:, :, max(-py0, 0): x.shape[2] - max(-py1, 0), max(-px0, 0): x.shape[3] - max(-px1, 0)]
    f = (f * (gain ** (f.ndim / 2))).to(x.dtype)
    if f.ndim == 2:
        x = F.conv2d(x, f[None, None].repeat(c, 1, 1, 1), groups=c)
    else:
        x = F.conv2d(x, f[None, None, None, :].repeat(c, 1, 1, 1), groups=c)
        x = F.conv2d(x, f[None, None, :, None].repeat(c, 1, 1, 1), groups=c)
    return x[:, :, ::down, ::down]


def upsample2d(x, f, up=2, padding=0, gain=1.0):
    """ops.py:117-133"""
    fw, fh = f.shape[-1], f.shape[0]
    p = (
        padding + (fw + up - 1) // 2,
        padding + (fw - up) // 2,
        padding + (fh + up - 1) // 2,
        padding + (fh - up) // 2,
    )
    return upfirdn2d(x, f, up=up, padding=p, gain=gain * up * up)


def normalize_2nd_moment(x, dim=1, eps=1e-8):
    """ops.py:142-143"""
    return x / ((x * x).mean(dim=dim, keepdim=True) + eps).sqrt()


def conv2d_resample(x, w, f=None, up=1, padding=0, groups=1, flip_weight=False):
    """ops.py:189-233 (down == 1 only; the path never down-samples).

    up == 1: plain correlation with symmetric padding.
    up  > 1: stride-``up`` transposed convolution (no kernel flip in-tree — SURVEY.md Q2;
             ``flip_weight=True`` gives the upstream NVIDIA behaviour) followed by
             upfirdn2d with the residual padding and gain up**2.
    """
    co, cig, kh, kw = w.shape
    fw, fh = (f.shape[-1], f.shape[0]) if f is not None else (1, 1)
    px0 = px1 = py0 = py1 = int(padding)
    if up > 1:
        px0 += (fw + up - 1) // 2
        px1 += (fw - up) // 2
        py0 += (fh + up - 1) // 2
        py1 += (fh - up) // 2
        if flip_weight:
            w = w.flip([2, 3])
        w = w.reshape(groups, co // groups, cig, kh, kw).transpose(1, 2)
        w = w.reshape(groups * cig, co // groups, kh, kw)
        px0 -= kw - 1
        px1 -= kw - up
        py0 -= kh - 1
        py1 -= kh - up
        pxt = max(min(-px0, -px1), 0)
        pyt = max(min(-py0, -py1), 0)
        x = F.conv_transpose2d(x, w, stride=up, padding=(pyt, pxt), groups=groups)
        return upfirdn2d(x, f, padding=(px0 + pxt, px1 + pxt, py0 + pyt, py1 + pyt), gain=up ** 2)
    assert px0 == px1 and py0 == py1 and px0 >= 0
    return F.conv2d(x, w, padding=(py0, px0), groups=groups)


def modulated_conv2d(x, weight, styles, noise=None, up=1, padding=0, resample_filter=None,
                     demodulate=True, flip_weight=False):
    """ops.py:146-186 — per-sample weights w = W * s, optional demodulation,
    executed as one grouped convolution, then + noise."""
    b, ci, h, w_ = x.shape
    co, _, kh, kw = weight.shape
    if x.dtype == torch.float16 and demodulate:  # ops.py:161-165 (fp16 only)
        weight = weight / (weight.abs().amax(dim=(1, 2, 3), keepdim=True) * sqrt(ci * kh * kw))
        styles = styles / styles.abs().amax(dim=1, keepdim=True)
    w = weight[None] * styles[:, None, :, None, None]
    if demodulate:
        w = w / ((w * w).sum((2, 3, 4), keepdim=True) + 1e-8).sqrt()
    y = conv2d_resample(x.reshape(1, b * ci, h, w_), w.reshape(b * co, ci, kh, kw), f=resample_filter,
                        up=up, padding=padding, groups=b, flip_weight=flip_weight)
    y = y.reshape(b, co, h * up, w_ * up)
    if noise is not None:
        y = y + noise
    return y


def resample(input, size, align_corners=True):
    """maua/ops/image.py:198-240 (sinc / lanczos / ramp / resample): lanczos-2 pre-filter on shrinking axes, then bicubic."""
    import math

    def sinc(v):
        return torch.where(v != 0, torch.sin(math.pi * v) / (math.pi * v), v.new_ones([]))

    def lanczos(v, a):
        cond = torch.logical_and(-a < v, v < a)
        out = torch.where(cond, sinc(v) * sinc(v / a), v.new_zeros([]))
        return out / out.sum()

    def ramp(ratio, width):
        n = math.ceil(width / ratio + 1)
        out = torch.empty([n])
        cur = 0
        for i in range(n):
            out[i] = cur
            cur += ratio
        return torch.cat([-out[1:].flip([0]), out])[1:-1]

    n, c, h, w = input.shape
    if isinstance(size, (int, float)):
        short, long = (w, h) if w <= h else (h, w)
        new_short, new_long = round(size), round(size * long / short)
        dw, dh = (new_short, new_long) if w <= h else (new_long, new_short)
    else:
        dh, dw = size
    x = input.reshape(n * c, 1, h, w)
    if dh < h:
        k = lanczos(ramp(dh / h, 2), 2).to(x)
        p = (k.shape[0] - 1) // 2
        x = F.conv2d(F.pad(x, (0, 0, p, p), "reflect"), k[None, None, :, None])
    if dw < w:
        k = lanczos(ramp(dw / w, 2), 2).to(x)
        p = (k.shape[0] - 1) // 2
        x = F.conv2d(F.pad(x, (p, p, 0, 0), "reflect"), k[None, None, None, :])
    x = x.reshape(n, c, h, w)
    return F.interpolate(x, (dh, dw), mode="bicubic", align_corners=align_corners)
