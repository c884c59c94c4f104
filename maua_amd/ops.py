"""Operator layer with the reference's names and argument meaning, executed by libmaua_hip.so.

Drop-in for maua/GAN/wrappers/inference/ops.py (reference): bias_act :65-84, upfirdn2d :87-114,
upsample2d :117-133, normalize_2nd_moment :142-143, modulated_conv2d :146-186, conv2d_resample :189-233,
setup_filter :236-256.  Scalars may be Python numbers or 0-dim tensors (the reference passes both).
Tensors are NCHW; float32 and bfloat16 are supported.  Every op raises MauaHipError when the HIP
library or device is missing — there is no CPU path here.
"""
import ctypes as C
from math import sqrt

import torch

from . import _lib as L

_DEFAULTS = {"linear": (0.0, 1.0), "relu": (0.0, sqrt(2)), "lrelu": (0.2, sqrt(2)), "tanh": (0.0, 1.0),
             "sigmoid": (0.0, 1.0), "elu": (0.0, 1.0), "selu": (0.0, 1.0), "softplus": (0.0, 1.0),
             "swish": (0.0, sqrt(2))}


def _num(v):
    return float(v.item()) if isinstance(v, torch.Tensor) else float(v)


def _int(v):
    return int(v.item()) if isinstance(v, torch.Tensor) else int(v)


def setup_filter(f=(1, 3, 3, 1), device=None, normalize=True, gain=1, separable=None):
    """ops.py:236-256 (host-side constant; 16 floats)."""
    if f is None:
        f = 1
    f = torch.as_tensor(f, dtype=torch.float32)
    if f.ndim == 0:
        f = f[None]
    if separable is None:
        separable = f.ndim == 1 and f.numel() >= 8
    if f.ndim == 1 and not separable:
        f = torch.outer(f, f)
    if normalize:
        f = f / f.sum()
    f = f * (_num(gain) ** (f.ndim / 2))
    return f.to(device) if device is not None else f


def bias_act(x, b=None, act="linear", alpha=None, gain=None, clamp=None):
    if act not in L.ACTS:
        raise ValueError(f"unknown activation {act!r}")
    da, dg = _DEFAULTS[act]
    alpha = da if alpha is None else _num(alpha)
    gain = dg if gain is None else _num(gain)
    clamp = -1.0 if clamp is None else _num(clamp)
    x = L.dev_tensor(x)
    if x.ndim != 4:
        raise ValueError("bias_act expects an NCHW tensor")
    n, c, h, w = x.shape
    if b is not None:
        b = L.dev_tensor(b, torch.float32)
        if b.numel() != c:
            raise ValueError("bias must have one entry per channel")
    y = torch.empty_like(x)
    L.check(L.lib().maua_bias_act(L.ctx(x.device), L.ptr(x), L.ptr(b), L.ptr(y), n, c, h, w, L.dtype_id(x),
                                  L.ACTS[act], C.c_float(alpha), C.c_float(gain), C.c_float(clamp)))
    return y


def get_activation_defaults(activation):
    """(alpha, gain) as scalar tensors - inference/ops.py:23-41; an unknown name gives the linear pair."""
    a, g = _DEFAULTS.get(activation, (0.0, 1.0))
    return torch.tensor(a), torch.tensor(g)


def activate(x, act, alpha):
    """The bare activation of inference/ops.py:44-62 on a tensor of any shape (no bias, gain 1, no clamp); an unknown
    name is the identity, like the reference's final else."""
    if act not in L.ACTS or act == "linear":
        return x
    x = L.dev_tensor(x)
    return bias_act(x.reshape(1, 1, 1, -1), None, act, alpha=alpha, gain=1.0).reshape(x.shape)


def upfirdn2d(x, f, up=1, down=1, padding=(0, 0, 0, 0), gain=1):
    x = L.dev_tensor(x)
    n, c, h, w = x.shape
    if f is None:
        f = torch.ones([1, 1], dtype=torch.float32)
    f = torch.as_tensor(f, dtype=torch.float32).cpu()
    g = _num(gain) ** (f.ndim / 2)
    if f.ndim == 1:  # separable: correlate with the outer product (same result as two 1-D passes)
        f2, g = torch.outer(f, f), g * g
    else:
        f2 = f
    up, down = _int(up), _int(down)
    pad = [_int(p) for p in (padding.tolist() if isinstance(padding, torch.Tensor) else padding)]
    if len(pad) == 2:
        pad = [pad[0], pad[0], pad[1], pad[1]]
    px0, px1, py0, py1 = pad
    fh, fw = f2.shape
    ho = (h * up + py0 + py1 - fh) // down + 1
    wo = (w * up + px0 + px1 - fw) // down + 1
    y = torch.empty((n, c, ho, wo), dtype=x.dtype, device=x.device)
    fd = L.dev_tensor(f2, torch.float32)
    L.check(L.lib().maua_upfirdn2d(L.ctx(x.device), L.ptr(x), L.ptr(fd), fh, fw, L.ptr(y), n, c, h, w, L.dtype_id(x),
                                   up, down, px0, px1, py0, py1, C.c_float(g)))
    return y


def _get_filter_size(f):
    return (1, 1) if f is None else (f.shape[-1], f.shape[0])


def upsample2d(x, f, up=2, padding=0, gain=1):
    up, padding = _int(up), _int(padding)
    fw, fh = _get_filter_size(f)
    p = (padding + (fw + up - 1) // 2, padding + (fw - up) // 2, padding + (fh + up - 1) // 2,
         padding + (fh - up) // 2)
    return upfirdn2d(x, f, up=up, padding=p, gain=_num(gain) * up * up)


def matmul_nt(a, b):
    """c [M, N] = a [M, K] @ b [N, K]^T in f32 on the HIP path (maua_matmul_nt) - F.linear without the bias.  Used for
    the small products of the path (mapping network, cosine bases) so that no BLAS library has to be initialised."""
    a = L.dev_tensor(a, torch.float32).contiguous()
    b = L.dev_tensor(b, torch.float32).contiguous()
    M, K = a.shape
    N = b.shape[0]
    assert b.shape[1] == K, (a.shape, b.shape)
    c = torch.empty((M, N), dtype=torch.float32, device=a.device)
    L.check(L.lib().maua_matmul_nt(L.ctx(a.device), L.ptr(a), L.ptr(b), L.ptr(c), M, N, K))
    return c


def add(a, b, out=None):
    """a + b on the HIP path (maua_add; f32 or bf16, summed in f32): the tensor additions of SynthesisBlock.forward
    (inference/stylegan2.py:360 "resnet" residual, :373 skip image)."""
    a = L.dev_tensor(a).contiguous()
    b = L.dev_tensor(b, a.dtype).contiguous()
    if a.shape != b.shape:
        raise ValueError(f"add: shapes differ, {tuple(a.shape)} vs {tuple(b.shape)}")
    out = torch.empty_like(a) if out is None else out
    L.check(L.lib().maua_add(L.ctx(a.device), L.ptr(a), L.ptr(b), L.ptr(out), C.c_long(a.numel()), L.dtype_id(a)))
    return out


def normalize_2nd_moment(x, dim=1, eps=1e-8):
    """ops.py:142-143 - the [P, D] prologue of the mapper (maua_normalize_2nd_moment: one wave per row)."""
    if x.dim() != 2 or dim not in (1, -1):
        return x / ((x * x).mean(dim=dim, keepdim=True) + eps).sqrt()
    x = L.dev_tensor(x, torch.float32)
    y = torch.empty_like(x)
    L.check(L.lib().maua_normalize_2nd_moment(L.ctx(x.device), L.ptr(x), x.shape[0], x.shape[1], C.c_float(eps), L.ptr(y)))
    return y


def repeat_rows(x, n):
    """x [P, D] -> [P, n, D] (w.unsqueeze(1).repeat(1, n, 1) of the mapper, inference/stylegan2.py:183)."""
    x = L.dev_tensor(x, torch.float32)
    out = torch.empty((x.shape[0], n, x.shape[1]), dtype=torch.float32, device=x.device)
    L.check(L.lib().maua_repeat_rows(L.ctx(x.device), L.ptr(x), x.shape[0], x.shape[1], n, L.ptr(out)))
    return out


def _is_render_filter(f):
    """True for setup_filter([1, 3, 3, 1]) - the one resample filter of the reference networks, which the fused up-layer kernels
    carry as constants."""
    if f is None:
        return False
    f = torch.as_tensor(f)
    return tuple(f.shape) == (4, 4) and torch.allclose(f.detach().cpu().float(), setup_filter([1, 3, 3, 1]), atol=1e-6)


def _up_paddings(k, fw, fh, up, padding):
    """ops.py:199-224's padding arithmetic for up > 1 (Q1: the scalar max(min(-p0, -p1), 0) it means):
    (transposed-conv padding x, y), (upfirdn2d padding x0, x1, y0, y1)."""
    px0 = px1 = py0 = py1 = padding
    px0 += (fw + up - 1) // 2 - (k - 1)
    px1 += (fw - up) // 2 - (k - up)
    py0 += (fh + up - 1) // 2 - (k - 1)
    py1 += (fh - up) // 2 - (k - up)
    pxt = max(min(-px0, -px1), 0)
    pyt = max(min(-py0, -py1), 0)
    return (pxt, pyt), (px0 + pxt, px1 + pxt, py0 + pyt, py1 + pyt)


def _modconv_up_composed(x, weight, styles, up, padding, f, demodulate, flip_weight):
    """ops.py:211-225 for ANY up factor / resample filter / padding, composed of the operators above (the fused up-layer kernels
    take the render path's up = 2, [1,3,3,1], 'same' case only):
      conv_transpose2d(x, w, stride=up, padding=pt)  ==  the valid correlation of the zero-stuffed input, padded by k - 1 - pt,
                                                         with the spatially flipped kernel (Q2: the reference does NOT flip
                                                         before its transposed convolution, so the flip happens here)
    then upfirdn2d(f, padding, gain = up^2).  up^2 the MACs of the minimal form - this is the off-path route."""
    k = weight.shape[2]
    fw, fh = _get_filter_size(f)
    (pxt, pyt), fpad = _up_paddings(k, fw, fh, up, padding)
    qx, qy = k - 1 - pxt, k - 1 - pyt
    zs = upfirdn2d(x, None, up=up, padding=(qx, qx - (up - 1), qy, qy - (up - 1)))
    wt = weight if flip_weight else weight.flip([2, 3])
    t = modulated_conv2d(zs, wt, styles, up=1, padding=k // 2, demodulate=demodulate)
    if k > 1:
        t = pad2d(t, (-(k // 2),) * 4)
    return upfirdn2d(t, f, padding=fpad, gain=up * up)


def modulated_conv2d(x, weight, styles, noise=None, up=1, down=1, padding=0, resample_filter=None, demodulate=True,
                     flip_weight=False, bias=None, act="linear", alpha=None, gain=None, clamp=None,
                     noise_strength=1.0):
    """ops.py:146-186.  Extra keyword arguments (bias/act/gain/clamp) fuse the bias_act that follows the
    convolution in every SynthesisLayer; leave them at their defaults for the bare reference op.
    The render path's cases (up = 1; up = 2 with the [1,3,3,1] filter) run as one fused launch; any other up factor or
    resample filter is composed of the same operators (_modconv_up_composed).  down != 1 and padding != k // 2 raise, as
    the reference does: its reshape to [B, Co, H * up, W * up] (ops.py:183) fits no other output size."""
    up, down, padding = _int(up), _int(down), _int(padding)
    x = L.dev_tensor(x)
    n, ci, h, w = x.shape
    co, wci, kh, kw = weight.shape
    if wci != ci or kh != kw or kh not in (1, 3):
        raise ValueError("weight must be [Co, Ci, k, k] with k in (1, 3)")
    if down != 1 or padding != kh // 2:
        raise ValueError(f"modulated_conv2d: down = {down}, padding = {padding} give an output that is not [B, Co, H * up, W * up] "
                         "(the reference's reshape at ops.py:183 raises for it too); conv2d_resample takes these")
    if up < 1:
        raise ValueError("up must be a positive integer")
    fused = up == 1 or (up == 2 and _is_render_filter(resample_filter))
    weight = L.dev_tensor(weight, torch.float32)
    styles = L.dev_tensor(styles, torch.float32)
    nstride = 0
    if noise is not None:
        noise = L.dev_tensor(noise, torch.float32)
        if noise.numel() == h * up * w * up:
            nstride = 0
        elif noise.numel() == n * h * up * w * up:
            nstride = h * up * w * up
        else:
            raise ValueError("noise must be [N|1, 1, H*up, W*up] (or [H*up, W*up])")
    if bias is not None:
        bias = L.dev_tensor(bias, torch.float32)
    if not fused:
        y = _modconv_up_composed(x, weight, styles, up, padding, resample_filter, demodulate, flip_weight)
        if noise is not None:
            nz = (noise.reshape(-1, 1, h * up, w * up) * _num(noise_strength)).to(y.dtype)
            y = add(y, nz.expand(n, co, h * up, w * up).contiguous())
        if bias is not None or act != "linear" or gain is not None or clamp is not None:
            y = bias_act(y, bias, act, alpha=alpha, gain=gain, clamp=clamp)
        return y
    da, dg = _DEFAULTS[act]
    alpha = da if alpha is None else _num(alpha)
    gain = dg if gain is None else _num(gain)
    clamp = -1.0 if clamp is None else _num(clamp)
    y = torch.empty((n, co, h * up, w * up), dtype=x.dtype, device=x.device)
    L.check(L.lib().maua_modconv2d(L.ctx(x.device), L.ptr(x), L.ptr(weight), L.ptr(styles), L.ptr(noise),
                                   C.c_long(nstride), C.c_float(_num(noise_strength)), L.ptr(bias), L.ptr(y), n, ci,
                                   co, h, w, kh, up, int(bool(demodulate)), int(bool(flip_weight)), L.ACTS[act],
                                   C.c_float(alpha), C.c_float(gain), C.c_float(clamp), L.dtype_id(x)))
    return y


def _conv2d_resample_one(x, w, f, up, down, padding, flip_weight):
    """conv2d_resample for one group (ops.py:189-233 with groups == 1)."""
    k = w.shape[2]
    styles = torch.ones((x.shape[0], x.shape[1]), dtype=torch.float32, device=x.device)
    if up > 1:
        if down == 1 and padding == k // 2 and up == 2 and _is_render_filter(f):
            return modulated_conv2d(x, w, styles, up=2, padding=padding, resample_filter=f, demodulate=False, flip_weight=flip_weight)
        y = _modconv_up_composed(x, L.dev_tensor(w, torch.float32), styles, up, padding, f, False, flip_weight)
        return upfirdn2d(y, f, down=down) if down > 1 else y        # ops.py:226-227
    if down != 1:
        raise NotImplementedError("conv2d_resample: down > 1 without up > 1 is the reference's 'Something weird is going on' "
                                  "assertion (ops.py:232)")
    if padding < 0:
        raise ValueError("padding must be >= 0 (ops.py:230)")
    if padding == k // 2:
        return modulated_conv2d(x, w, styles, up=1, padding=padding, demodulate=False)
    # F.conv2d(x, w, padding = p) = the 'same' convolution of the input padded by p, without its k // 2 border
    y = modulated_conv2d(pad2d(x, (padding,) * 4) if padding else x, w, styles, up=1, padding=k // 2, demodulate=False)
    return pad2d(y, (-(k // 2),) * 4) if k > 1 else y


def conv2d_resample(x, w, f=None, up=1, down=1, padding=0, groups=1, flip_weight=False):
    """ops.py:189-233: an un-modulated convolution = modulated_conv2d with unit styles and no demodulation.  up = 1 with any
    padding >= 0; up > 1 with any resample filter and padding, followed by the down-sampling FIR when down > 1 (:226-227);
    groups > 1 (the form modulated_conv2d hands over, x [1, G * Ci, H, W], w [G * Co, Ci, k, k]) runs group by group - inside
    this library the per-sample weights of that form never exist, modulated_conv2d is the fused operator."""
    up, down, padding, groups = _int(up), _int(down), _int(padding), _int(groups)
    x = L.dev_tensor(x)
    co, cig, kh, kw = w.shape
    if kh != kw or kh not in (1, 3):
        raise ValueError("weight must be [Co, Ci / groups, k, k] with k in (1, 3)")
    if groups == 1:
        return _conv2d_resample_one(x, w, f, up, down, padding, flip_weight)
    if x.shape[1] != groups * cig or co % groups:
        raise ValueError("x must be [N, groups * Ci, H, W] and w [groups * Co, Ci, k, k]")
    cog = co // groups
    outs = [_conv2d_resample_one(x[:, g * cig:(g + 1) * cig].contiguous(), w[g * cog:(g + 1) * cog].contiguous(), f, up, down, padding,
                                 flip_weight) for g in range(groups)]
    return torch.cat(outs, dim=1)


_PAD_HOW = {"circular": 0, "reflect": 1, "replicate": 2, "constant": 3}


def interpolate_bicubic(x, size):
    """torch.nn.functional.interpolate(x, size, mode="bicubic", align_corners=False) on NCHW f32 / bf16 — the
    "stretch" resize of wrappers/stylegan2.py:231,253."""
    x = L.dev_tensor(x)
    n, c, h, w = x.shape
    oh, ow = int(size[0]), int(size[1])
    y = torch.empty((n, c, oh, ow), dtype=x.dtype, device=x.device)
    L.check(L.lib().maua_resize2d(L.ctx(x.device), L.ptr(x), L.ptr(y), n, c, h, w, oh, ow, 0, 0, 0, 3, C.c_float(0.0),
                                  L.dtype_id(x)))
    return y


def pad2d(x, padding, mode="constant", value=0.0):
    """torch.nn.functional.pad(x, (left, right, top, bottom), mode, value) on NCHW (wrappers/stylegan2.py:294);
    negative entries crop (the pad strategies' inverse, :313-323)."""
    x = L.dev_tensor(x)
    n, c, h, w = x.shape
    pl, pr, pt, pb = (int(p) for p in padding)
    oh, ow = h + pt + pb, w + pl + pr
    y = torch.empty((n, c, oh, ow), dtype=x.dtype, device=x.device)
    L.check(L.lib().maua_resize2d(L.ctx(x.device), L.ptr(x), L.ptr(y), n, c, h, w, oh, ow, 1, pl, pt, _PAD_HOW[mode],
                                  C.c_float(float(value)), L.dtype_id(x)))
    return y


def _lanczos_taps(ratio, a=2):
    """maua/ops/image.py:198-211 lanczos(ramp(ratio, a), a), in the reference's float32 arithmetic."""
    import math
    n = math.ceil(a / ratio + 1)
    out = torch.empty([n])
    cur = 0
    for i in range(n):
        out[i] = cur
        cur += ratio
    x = torch.cat([-out[1:].flip([0]), out])[1:-1]
    sinc = lambda v: torch.where(v != 0, torch.sin(math.pi * v) / (math.pi * v), v.new_ones([]))
    k = torch.where(torch.logical_and(-a < x, x < a), sinc(x) * sinc(x / a), x.new_zeros([]))
    return k / k.sum()


def resample(input, size, align_corners=True):
    """maua/ops/image.py:214-240: lanczos pre-filter along every axis that shrinks (reflect-padded 1-D correlation),
    then bicubic to ``size`` = (h, w) (or the new length of the short side).  The post-render step of
    MauaPatch.force_output_size (patches/base/__init__.py:21-25)."""
    x = L.dev_tensor(input, torch.float32)
    n, c, h, w = x.shape
    if isinstance(size, (int, float)):
        short, long = (w, h) if w <= h else (h, w)
        new_short, new_long = round(size), round(size * long / short)
        dw, dh = (new_short, new_long) if w <= h else (new_long, new_short)
    else:
        dh, dw = (int(s) for s in size)
    lib, ctx = L.lib(), L.ctx(x.device)
    for axis, (d, s) in enumerate(((dh, h), (dw, w))):
        if d < s:
            taps = L.dev_tensor(_lanczos_taps(d / s, 2), torch.float32)
            y = torch.empty_like(x)
            L.check(lib.maua_conv1d_reflect(ctx, L.ptr(x), L.ptr(y), L.ptr(taps), (taps.numel() - 1) // 2, axis,
                                            C.c_long(n * c), h, w))
            x = y
    out = torch.empty((n, c, dh, dw), dtype=torch.float32, device=x.device)
    L.check(lib.maua_resize2d(ctx, L.ptr(x), L.ptr(out), n, c, h, w, dh, dw, 2 if align_corners else 0, 0, 0, 3,
                              C.c_float(0.0), L.dtype_id(x)))
    return out


def resample_size(h, w, size):
    """The (dh, dw) ``resample(x, size)`` produces for an h x w input (image.py:217-223)."""
    if isinstance(size, (int, float)):
        short, long = (w, h) if w <= h else (h, w)
        new_short, new_long = round(size), round(size * long / short)
        dw, dh = (new_short, new_long) if w <= h else (new_long, new_short)
        return dh, dw
    dh, dw = (int(s) for s in size)
    return dh, dw


def resample_vjp(grad_out, in_shape, align_corners=True):
    """(d resample(x, size) / d x)^T grad_out for x of shape ``in_shape`` = (n, c, h, w) and grad_out [n, c, dh, dw]: the bicubic
    interpolation's adjoint, then the adjoints of the lanczos pre-filters in reverse order - what ``torch.autograd.grad`` walks back
    through image.py:225-240 (LPIPSGrads, maua/grad.py:191-192).  There is no autograd here."""
    g = L.dev_tensor(grad_out, torch.float32)
    n, c, h, w = (int(v) for v in in_shape)
    dh, dw = g.shape[-2:]
    lib, ctx = L.lib(), L.ctx(g.device)
    x = torch.empty((n, c, h, w), dtype=torch.float32, device=g.device)
    L.check(lib.maua_resize2d_bicubic_vjp(ctx, L.ptr(g), L.ptr(x), n, c, h, w, dh, dw, int(bool(align_corners))))
    for axis, (d, s) in reversed(list(enumerate(((dh, h), (dw, w))))):
        if d < s:
            taps = L.dev_tensor(_lanczos_taps(d / s, 2), torch.float32)
            y = torch.empty_like(x)
            L.check(lib.maua_conv1d_reflect_vjp(ctx, L.ptr(x), L.ptr(y), L.ptr(taps), (taps.numel() - 1) // 2, axis, C.c_long(n * c), h, w))
            x = y
    return x
