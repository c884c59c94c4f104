"""Oracle restatement of the in-tree StyleGAN2 networks (test infrastructure only).

Follows maua/GAN/wrappers/inference/stylegan2.py.  Networks are plain
functions over a ``dict[str, Tensor]`` whose keys equal the reference modules'
``state_dict()`` keys (``bs.3.conv0.affine.weight`` ...), so a state dict
exported from the reference can be fed in directly (that is how the layer-level
goldens are checked).

``nv_compat=False`` (default) reproduces the in-tree behaviour, quirks included
(SURVEY.md Q2-Q4): no kernel flip on up-layers, ``x @ w`` in the non-linear FC
branch, unscaled noise.
"""
from math import sqrt

import numpy as np
import torch
import torch.nn.functional as F

from . import ops


# ----------------------------------------------------------------------------- structure
def block_resolutions(img_resolution):
    """stylegan2.py:399-402"""
    return [2 ** i for i in range(2, int(np.log2(img_resolution)) + 1)]


def channels_dict(img_resolution, channel_base=32768, channel_max=512):
    """stylegan2.py:403"""
    return {r: min(channel_base // r, channel_max) for r in block_resolutions(img_resolution)}


def num_ws(img_resolution):
    """stylegan2.py:406-426: one w per conv layer plus one for the last toRGB."""
    n = len(block_resolutions(img_resolution))
    return 2 * n  # (2n - 1) convs + 1


def layer_list(img_resolution, channel_base=32768, channel_max=512):
    """The 2n-1 synthesis layers in execution order: (prefix, c_in, c_out, res, up)."""
    ch = channels_dict(img_resolution, channel_base, channel_max)
    out = []
    for i, r in enumerate(block_resolutions(img_resolution)):
        if r > 4:
            out.append((f"bs.{i}.conv0", ch[r // 2], ch[r], r, 2))
        out.append((f"bs.{i}.conv1", ch[r], ch[r], r, 1))
    return out


# ----------------------------------------------------------------------------- random init
def init_synthesis_params(img_resolution, w_dim=512, img_channels=3, channel_base=32768, channel_max=512,
                          generator=None, architecture="skip"):
    """Draw parameters in the order the reference constructors draw them
    (stylegan2.py:296-337 block - toRGB only in the last block unless "skip", the 1 x 1 skip convolution only for
    "resnet" -, :221-227 layer, :263-265 toRGB, :43-44 FC, :87-88 Conv2dLayer), so that
    ``torch.manual_seed(s); SynthesisNetwork(...)`` and this function with
    ``torch.Generator().manual_seed(s)`` produce the same tensors."""
    g = generator
    ch = channels_dict(img_resolution, channel_base, channel_max)
    f = ops.setup_filter([1, 3, 3, 1])
    p = {}

    def fc(prefix, fin, fout, bias_init):
        p[prefix + ".weight"] = torch.randn([fout, fin], generator=g)
        p[prefix + ".bias"] = torch.full([fout], float(bias_init))

    def layer(prefix, cin, cout, res):
        p[prefix + ".resample_filter"] = f.clone()
        fc(prefix + ".affine", w_dim, cin, 1.0)
        p[prefix + ".weight"] = torch.randn([cout, cin, 3, 3], generator=g)
        p[prefix + ".noise_const"] = torch.randn([res, res], generator=g)
        p[prefix + ".bias"] = torch.zeros([cout])

    for i, r in enumerate(block_resolutions(img_resolution)):
        cin = ch[r // 2] if r > 4 else 0
        cout = ch[r]
        p[f"bs.{i}.resample_filter"] = f.clone()
        if cin == 0:
            p[f"bs.{i}.const"] = torch.randn([cout, r, r], generator=g)
        else:
            layer(f"bs.{i}.conv0", cin, cout, r)
        layer(f"bs.{i}.conv1", cout, cout, r)
        if architecture == "skip" or r == img_resolution:
            fc(f"bs.{i}.torgb.affine", w_dim, cout, 1.0)
            p[f"bs.{i}.torgb.weight"] = torch.randn([img_channels, cout, 1, 1], generator=g)
            p[f"bs.{i}.torgb.bias"] = torch.zeros([img_channels])
        if architecture == "resnet" and cin != 0:
            p[f"bs.{i}.skip.resample_filter"] = f.clone()
            p[f"bs.{i}.skip.weight"] = torch.randn([cout, cin, 1, 1], generator=g)
    return p


def init_mapping_params(z_dim=512, w_dim=512, num_layers=8, lr_multiplier=0.01, generator=None):
    """stylegan2.py:140-159 with FullyConnectedLayer :43-44 (weight = randn / lr_multiplier)."""
    p = {}
    feats = [z_dim] + [w_dim] * num_layers
    for i in range(num_layers):
        p[f"fcs.{i}.weight"] = torch.randn([feats[i + 1], feats[i]], generator=generator) / lr_multiplier
        p[f"fcs.{i}.bias"] = torch.zeros([feats[i + 1]])
    p["w_avg"] = torch.zeros([w_dim])
    return p


# ----------------------------------------------------------------------------- layers
def fully_connected(x, weight, bias, activation="linear", lr_multiplier=1.0, nv_compat=False):
    """stylegan2.py:48-58.  Non-linear branch uses x @ w in-tree (Q3)."""
    w = weight * (lr_multiplier / sqrt(weight.shape[1]))
    b = bias
    if b is not None and lr_multiplier != 1.0:
        b = b * lr_multiplier
    if activation == "linear":
        return F.linear(x, w, b)
    y = F.linear(x, w if nv_compat else w.T, None)
    return ops.bias_act(y, b, act=activation)


def mapping_network(p, z, truncation_psi=1.0, num_ws_=18, lr_multiplier=0.01, nv_compat=False,
                    truncation_cutoff=None):
    """stylegan2.py:161-192 (c_dim == 0)."""
    x = ops.normalize_2nd_moment(z)
    i = 0
    while f"fcs.{i}.weight" in p:
        x = fully_connected(x, p[f"fcs.{i}.weight"], p[f"fcs.{i}.bias"], "lrelu", lr_multiplier, nv_compat)
        i += 1
    x = x.unsqueeze(1).repeat(1, num_ws_, 1)
    if truncation_psi != 1:
        if truncation_cutoff is None:
            x = p["w_avg"].lerp(x, truncation_psi)
        else:
            x[:, :truncation_cutoff] = p["w_avg"].lerp(x[:, :truncation_cutoff], truncation_psi)
    return x


def synthesis_layer(p, prefix, x, w, up=1, noise=None, noise_strength=1.0, gain=1.0, conv_clamp=256.0,
                    nv_compat=False):
    """stylegan2.py:229-251.  ``noise`` defaults to the layer's noise_const (noise_mode='const')."""
    styles = fully_connected(w, p[prefix + ".affine.weight"], p[prefix + ".affine.bias"])
    if noise is None:
        noise = p[prefix + ".noise_const"]
    if nv_compat and (prefix + ".noise_strength") in p:  # upstream: learned per-layer strength
        noise_strength = float(p[prefix + ".noise_strength"].reshape(-1)[0])
    noise = noise * noise_strength
    x = ops.modulated_conv2d(x, p[prefix + ".weight"], styles, noise=noise, up=up, padding=1,
                             resample_filter=p[prefix + ".resample_filter"], flip_weight=nv_compat)
    return ops.bias_act(x, p[prefix + ".bias"], act="lrelu", gain=sqrt(2) * gain,
                        clamp=None if conv_clamp is None else conv_clamp * gain)


def torgb_layer(p, prefix, x, w, conv_clamp=256.0):
    """stylegan2.py:268-272"""
    cin = p[prefix + ".weight"].shape[1]
    styles = fully_connected(w, p[prefix + ".affine.weight"], p[prefix + ".affine.bias"]) * (1 / sqrt(cin))
    x = ops.modulated_conv2d(x, p[prefix + ".weight"], styles, demodulate=False)
    return ops.bias_act(x, p[prefix + ".bias"], clamp=conv_clamp)


def conv2d_layer(p, prefix, x, up=1, gain=1.0, act="linear", conv_clamp=None):
    """Conv2dLayer.forward, stylegan2.py:100-113 (the "resnet" blocks' 1 x 1 up-sampling skip convolution: bias=False,
    activation "linear" whose def_gain is 1)."""
    w = p[prefix + ".weight"]
    w = w * (1 / sqrt(w.shape[1] * w.shape[2] * w.shape[3]))
    x = ops.conv2d_resample(x, w, f=p[prefix + ".resample_filter"], up=up, padding=w.shape[-1] // 2)
    return ops.bias_act(x, p.get(prefix + ".bias"), act=act, gain=(sqrt(2) if act == "lrelu" else 1.0) * gain,
                        clamp=None if conv_clamp is None else conv_clamp * gain)


def _hook_resize(x, rs, feat):
    """get_hook's resize (wrappers/stylegan2.py:223-250 "stretch", :284-313 "pad-*"): the torch ops the hooks call."""
    if rs["mode"] == "stretch":
        x = F.interpolate(x, tuple(rs["target"]), mode="bicubic", align_corners=False)
    else:
        x = F.pad(x, tuple(rs["padding"]), mode=rs.get("pad_how", "constant"), value=rs.get("pad_value", 0.0))
    if feat and rs.get("fill") is not None:
        x = x + rs["fill"][None].to(x)
    return x


def _hook_inverse(x, rs, layer_size):
    """get_hook's inverse (:252-253 bicubic back to the layer size, :315-327 crop the padding)."""
    if rs["mode"] == "stretch":
        return F.interpolate(x, (layer_size, layer_size), mode="bicubic", align_corners=False)
    pl, pr, pt, pb = rs["padding"]
    return x[..., pt:x.shape[-2] - pb, pl:x.shape[-1] - pr]


def warp_affine(x, M):
    """kornia.geometry.transform.warp_affine(x, M, (h, w), mode="bilinear", padding_mode="reflection",
    align_corners=True) as kornia composes it (un-vendored; wrappers/stylegan2.py:153-194 reach it through
    kT.translate / rotate / scale): normalise the pixel-space affine, invert, F.affine_grid + F.grid_sample."""
    B, _, h, w = x.shape
    M3 = torch.eye(3).repeat(B, 1, 1)
    M3[:, :2] = M
    N = torch.tensor([[2.0 / max(w - 1, 1e-14), 0, -1], [0, 2.0 / max(h - 1, 1e-14), -1], [0, 0, 1]])
    theta = torch.linalg.inv(N @ M3 @ torch.linalg.inv(N))[:, :2]
    grid = F.affine_grid(theta, [B, x.shape[1], h, w], align_corners=True)
    return F.grid_sample(x, grid, mode="bilinear", padding_mode="reflection", align_corners=True)


def synthesis_network(p, ws, noise=None, noise_strength=1.0, nv_compat=False, return_features=False, resize=None,
                      warps=(), architecture="skip"):
    """stylegan2.py:429-436 + SynthesisBlock.forward :340-382 ("skip" - the reference default - here; "orig" and "resnet"
    in _synthesis_network_plain below).

    ``noise``: optional list, one [B|1,1,h,w] tensor per synthesis layer in
    execution order (what StyleGAN2Synthesizer.forward installs, wrappers/stylegan2.py:85-100).
    ``resize``: optional dict(layer, mode "stretch"|"pad", target (h, w), padding (l, r, t, b), pad_how, pad_value,
    fill [C, h, w]) — the forward (pre-)hooks of change_output_resolution (wrappers/stylegan2.py:104-151):
    layer 0 resizes the input of the first layer, layer L >= 1 the output of synthesis layer L-1, whose block also
    gets the rgb (inverse) and img (resize) hooks.  ``warps``: sequence of (layer, M [B,2,3]) forward hooks applied in
    order after the resize hook of their layer (the translate / zoom / rotate hooks, :153-194).
    """
    if architecture != "skip":
        if resize is not None or warps or return_features:
            raise NotImplementedError("hooks and feature capture are restated for the 'skip' architecture only")
        return _synthesis_network_plain(p, ws, noise, noise_strength, nv_compat, architecture)

    def hooks(x_, l1):
        for wl, M in warps:
            if wl == l1:
                x_ = warp_affine(x_, M)
        return x_
    nblocks = 0
    while f"bs.{nblocks}.conv1.weight" in p:
        nblocks += 1
    x = img = None
    w_idx = 0
    li = 0
    feats = []
    rs_layer = -1 if resize is None else resize["layer"]
    for i in range(nblocks):
        def nz():
            return None if noise is None or li >= len(noise) else noise[li]
        hooked_here = False
        if i == 0:
            x = p["bs.0.const"].unsqueeze(0).repeat(ws.shape[0], 1, 1, 1)
            if rs_layer == 0:
                x = _hook_resize(x, resize, True)
        else:
            x = synthesis_layer(p, f"bs.{i}.conv0", x, ws[:, w_idx], up=2, noise=nz(),
                                noise_strength=noise_strength, nv_compat=nv_compat)
            if rs_layer == li + 1:
                x, hooked_here = _hook_resize(x, resize, True), True
            x = hooks(x, li + 1)
            feats.append(x)
            w_idx += 1
            li += 1
        x = synthesis_layer(p, f"bs.{i}.conv1", x, ws[:, w_idx], up=1, noise=nz(),
                            noise_strength=noise_strength, nv_compat=nv_compat)
        if rs_layer == li + 1:
            x, hooked_here = _hook_resize(x, resize, True), True
        x = hooks(x, li + 1)
        feats.append(x)
        w_idx += 1
        li += 1
        if img is not None:
            img = ops.upsample2d(img, p[f"bs.{i}.resample_filter"])
        y = torgb_layer(p, f"bs.{i}.torgb", x, ws[:, w_idx])
        if hooked_here:
            y = _hook_inverse(y, resize, 4 * 2 ** i)
        img = y if img is None else img + y
        if hooked_here:
            img = _hook_resize(img, resize, False)
    if return_features:
        return img, feats
    return img


def _synthesis_network_plain(p, ws, noise, noise_strength, nv_compat, architecture):
    """SynthesisBlock.forward :340-382 for architecture "orig" (no skip images: only the last block has a toRGB layer) and
    "resnet" (y = skip(x, gain sqrt(.5)); x = conv1(conv0(x), gain sqrt(.5)); x = y + x).  The reference cannot execute its own
    up = 2 layers, so this restatement is pinned on the op / layer goldens only."""
    if architecture not in ("orig", "resnet"):
        raise ValueError(architecture)
    nblocks = 0
    while f"bs.{nblocks}.conv1.weight" in p:
        nblocks += 1
    x = img = None
    w_idx = li = 0

    def nz():
        return None if noise is None or li >= len(noise) else noise[li]
    for i in range(nblocks):
        g1 = 1.0
        if i == 0:
            x = p["bs.0.const"].unsqueeze(0).repeat(ws.shape[0], 1, 1, 1)
        else:
            y = conv2d_layer(p, f"bs.{i}.skip", x, up=2, gain=sqrt(0.5)) if architecture == "resnet" else None
            x = synthesis_layer(p, f"bs.{i}.conv0", x, ws[:, w_idx], up=2, noise=nz(), noise_strength=noise_strength,
                                nv_compat=nv_compat)
            w_idx, li = w_idx + 1, li + 1
            g1 = sqrt(0.5) if architecture == "resnet" else 1.0
        x = synthesis_layer(p, f"bs.{i}.conv1", x, ws[:, w_idx], up=1, noise=nz(), noise_strength=noise_strength,
                            gain=g1, nv_compat=nv_compat)
        w_idx, li = w_idx + 1, li + 1
        if i > 0 and architecture == "resnet":
            x = y + x
        if i == nblocks - 1:
            img = torgb_layer(p, f"bs.{i}.torgb", x, ws[:, w_idx])
    return img
