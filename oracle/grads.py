"""CPU oracle (TEST INFRASTRUCTURE, never imported by the product) of the reference's image-prompt grad modules.

Restates, in torch-CPU float32 with autograd:
  * maua/grad.py:27-70       differentiable_histogram / ColorMatchGrads (hue histogram of the saturation-weighted image, MSE to the style
                             image's histogram, ``torch.autograd.grad`` back to the image) - pinned by tests/golden/g34_grads.npz: the
                             reference's own functions run with this file's ``rgb_to_hsv`` standing in for the absent ``kornia``
  * maua/grad.py:73-93       VGGGrads around maua/perceptors/__init__.py:10-97 (Perceptor hooks) and vgg_kbc.py:10-71 (KBCPerceptor)
  * maua/grad.py:178-196     LPIPSGrads
  * maua/loss.py:33-80       scaled_mse_loss / feature_loss / gram_matrix - pinned by g34 (the reference's own functions)
  * maua/ops/cutouts.py:101-206  DangoCutouts with ``skip_augs=True`` (rectangle arithmetic, draws and their order, overview /
                             inner-crop / grey schedule) - pinned by g34 with stand-ins for the absent packages around it
and third-party pieces that are ABSENT from /root/reference and from this image - **parity unpinned**, restated from their published
form:
  * ``kornia.color.rgb_to_hsv`` (setup.py:59 "kornia", unpinned): h in [0, 2 pi), s = delta / (max + eps), v = max;
  * ``torchvision.models.vgg19 / vgg16 .features`` (3x3 convolutions + ReLU + MaxPool2d(2), the published configurations "E" / "D"),
    ``torchvision.transforms.Normalize`` / ``Grayscale`` (ITU-R 601-2 luma) / ``functional.hflip``;
  * ``lpips.LPIPS(net="vgg")`` (setup.py:61 "lpips", unpinned; Zhang et al. 2018, version 0.1): ScalingLayer, the five ReLU taps of
    VGG16, unit-normalised features, squared difference, non-negative 1x1 ``lin`` layers, spatial mean, sum over the taps.
The reference's ``Perceptor.forward`` returns ``torch.nested_tensor(...)``, an attribute this image's torch (2.10) no longer has, and its
``gram_matrix`` folds the batch into the channel axis ((B C) x (B C)); VGGGrads is therefore only well-defined for one image per call,
which is what the sampler hands it.  Here - and in the HIP module - a batch is B independent images against the same targets (identical
for B = 1).
"""
import math

import torch
import torch.nn.functional as F

IMAGENET_MEAN = (0.485, 0.456, 0.406)     # vgg_kbc.py:33
IMAGENET_STD = (0.229, 0.224, 0.225)
LPIPS_SHIFT = (-0.030, -0.088, -0.188)    # lpips ScalingLayer
LPIPS_SCALE = (0.458, 0.448, 0.450)
VGG19_CFG = (64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M")
VGG16_CFG = (64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M")
KBC_STYLE_LAYERS = (1, 6, 11, 20, 29)     # vgg_kbc.py:27: relu1_1 .. relu5_1 of vgg19.features
LPIPS_TAPS = (3, 8, 15, 22, 29)           # relu1_2, relu2_2, relu3_3, relu4_3, relu5_3 of vgg16.features
LPIPS_CHNS = (64, 128, 256, 512, 512)


# ------------------------------------------------------------------------------------------ kornia.color.rgb_to_hsv (published form)
def rgb_to_hsv(image, eps=1e-8):
    max_rgb, argmax_rgb = image.max(-3)
    min_rgb, _ = image.min(-3)
    deltac = max_rgb - min_rgb
    v = max_rgb
    s = deltac / (max_rgb + eps)
    deltac = torch.where(deltac == 0, torch.ones_like(deltac), deltac)
    rc, gc, bc = torch.unbind(max_rgb.unsqueeze(-3) - image, dim=-3)
    h1 = bc - gc
    h2 = (rc - bc) + 2.0 * deltac
    h3 = (gc - rc) + 4.0 * deltac
    h = torch.stack((h1, h2, h3), dim=-3) / deltac.unsqueeze(-3)
    h = torch.gather(h, dim=-3, index=argmax_rgb.unsqueeze(-3)).squeeze(-3)
    h = (h / 6.0) % 1.0
    h = 2.0 * math.pi * h
    return torch.stack((h, s, v), dim=-3)


# ------------------------------------------------------------------------------------------ grad.py:27-70
def histogram_bins(nbins=255):
    """grad.py:30-31: ``torch.arange(nbins + 1) * (1 / (nbins - 1))`` - float32 products of an integer ramp."""
    return torch.arange(nbins + 1) * (1 / (nbins - 1))


def differentiable_histogram(x, weighting=None, nbins=255):
    """grad.py:27-47 without the 255 passes: a value in [bins[k], bins[k + 1]) adds (bins[k + 1] - x) w to bin k (``mask_plus`` of
    dim = k) and (x - bins[k]) w to bin k + 1 (``mask_sub`` of dim = k + 1; for dim = 0 the reference's ``bins[dim - 1]`` is
    bins[-1] = the LAST edge, so that mask is empty).  Normalised to sum 1 per sample."""
    B = x.shape[0]
    bins = histogram_bins(nbins).to(x.device)
    if weighting is None:
        weighting = torch.ones_like(x)
    xf, wf = x.reshape(B, -1), weighting.reshape(B, -1)
    k = torch.bucketize(xf.detach().contiguous(), bins, right=True) - 1          # bins[k] <= x < bins[k + 1]; -1 below bins[0], nbins at / above the last edge
    plus_ok = (k >= 0) & (k <= nbins - 1)                           # dim = k exists (a value outside [bins[0], bins[nbins]) is in no mask)
    sub_ok = plus_ok & (k + 1 <= nbins - 1)                         # dim = k + 1 exists
    k = k.clamp(0, nbins)
    lo, hi = bins[k], bins[(k + 1).clamp(max=nbins)]
    hist = torch.zeros(B, nbins + 2, device=x.device, dtype=x.dtype)
    hist = hist.scatter_add(1, k, torch.where(plus_ok, (hi - xf) * wf, torch.zeros_like(xf)))
    hist = hist.scatter_add(1, k + 1, torch.where(sub_ok, (xf - lo) * wf, torch.zeros_like(xf)))
    hist = hist[:, :nbins]
    return hist / hist.sum(-1, keepdim=True)


def colormatch_histogram(img, saturation_weighting=True, bins=255):
    """ColorMatchGrads.histogram (grad.py:56-63).  kornia's hue is in radians, the reference clamps it to [0, 1]: every hue above
    one radian lands on the last edge."""
    hue, sat, val = rgb_to_hsv(img.add(1).div(2).clamp(1e-8, 1 - 1e-8)).clamp(0, 1).unbind(1)
    weighting = (sat * val).sqrt() if saturation_weighting else None
    return differentiable_histogram(hue, weighting, bins)


def colormatch_grads(img, target, scale=1.0, saturation_weighting=True, bins=255):
    """ColorMatchGrads.forward (grad.py:67-70) -> (grad, loss)."""
    with torch.enable_grad():
        x = img.clone().requires_grad_()
        loss = scale * F.mse_loss(colormatch_histogram(x, saturation_weighting, bins), target.expand(img.shape[0], -1))
        grad = torch.autograd.grad(loss, x)[0]
    return grad, loss.detach()


# ------------------------------------------------------------------------------------------ loss.py:33-80
def scaled_mse_loss(input, target, eps=1e-8):
    diff = input - target
    return diff.pow(2).sum() / diff.abs().sum().add(eps)


def feature_loss(input, target):
    """loss.py:40-54 at its defaults (norm_weights="elements", scaled=True)."""
    return scaled_mse_loss(input, target) / input.numel()


def gram_matrix(x):
    """loss.py:57-80 at its defaults for ONE image: [1, C, H, W] -> [C, C]."""
    B, C, H, W = x.shape
    f = x.reshape(B * C, H * W)
    return f @ f.T


# ------------------------------------------------------------------------------------------ torchvision VGG features (published form)
def vgg_plan(cfg, last_index):
    """[(kind, channels)] for ``features[: last_index + 1]`` + the features-index of every entry: conv and its ReLU are one entry
    (index of the ReLU), 'M' is MaxPool2d(2)."""
    ops, idx, i, cin = [], [], 0, 3
    for v in cfg:
        if v == "M":
            if i > last_index:
                break
            ops.append(("pool", cin)); idx.append(i); i += 1
        else:
            if i + 1 > last_index:
                break
            ops.append(("conv", v)); idx.append(i + 1); i += 2
            cin = v
    return ops, idx


def vgg_param_shapes(cfg, last_index):
    """torchvision key -> shape for ``features[: last_index + 1]`` ("0.weight", "0.bias", "2.weight", ...)."""
    shapes, i, cin = {}, 0, 3
    for v in cfg:
        if v == "M":
            i += 1
        else:
            if i + 1 > last_index:
                break
            shapes[f"{i}.weight"] = (v, cin, 3, 3)
            shapes[f"{i}.bias"] = (v,)
            cin = v
            i += 2
    return shapes


def init_vgg_params(cfg, last_index, generator=None, gain=1.0):
    """He-style random weights (no checkpoint in the image); biases small and positive so that ReLU keeps about half its inputs."""
    p = {}
    for k, s in vgg_param_shapes(cfg, last_index).items():
        if k.endswith("weight"):
            p[k] = torch.randn(s, generator=generator) * (gain * math.sqrt(2.0 / (s[1] * 9)))
        else:
            p[k] = torch.randn(s, generator=generator) * 0.05
    return p


def vgg_features(p, cfg, x, taps, first_padding="zeros"):
    """``features`` on a pre-processed image, returning the outputs of the ReLU modules listed in ``taps`` (features indices)."""
    feats, i, h, last = {}, 0, x, max(taps)
    for v in cfg:
        if i > last:
            break
        if v == "M":
            h = F.max_pool2d(h, 2)
            i += 1
        else:
            if i == 0 and first_padding == "replicate":                       # vgg_kbc.py:40 _change_padding_mode
                h = F.conv2d(F.pad(h, (1, 1, 1, 1), mode="replicate"), p[f"{i}.weight"], p[f"{i}.bias"])
            else:
                h = F.conv2d(h, p[f"{i}.weight"], p[f"{i}.bias"], padding=1)
            h = F.relu(h)
            if i + 1 in taps:
                feats[i + 1] = h
            i += 2
    return [feats[t] for t in taps]


def normalize_img(x, mean, std):
    m = torch.tensor(mean, dtype=x.dtype).reshape(1, 3, 1, 1)
    s = torch.tensor(std, dtype=x.dtype).reshape(1, 3, 1, 1)
    return (x - m) / s


# ------------------------------------------------------------------------------------------ grad.py:73-93 VGGGrads
def kbc_style_embeddings(p, img01, style_layers=KBC_STYLE_LAYERS):
    """Perceptor.get_target_embeddings(None, [img]) (perceptors/__init__.py:44-76) for one style image in [0, 1]: the Gram matrices
    of the style layers.  img01 [B, 3, H, W] -> per layer [B, C, C] (one Gram matrix per image)."""
    feats = vgg_features(p, VGG19_CFG, normalize_img(img01, IMAGENET_MEAN, IMAGENET_STD), style_layers, "replicate")
    return [torch.stack([gram_matrix(f[b:b + 1]) for b in range(f.shape[0])]) for f in feats]


def vgg_grads(p, img, targets, scale=1.0, style_layers=KBC_STYLE_LAYERS):
    """VGGGrads.forward (grad.py:90-93): loss = sum over the style layers of style_strength * feature_loss(gram, target)
    (perceptors/__init__.py:33-40), style_strength = scale (grad.py:76), per image; -> (grad [B, 3, H, W], losses [B])."""
    with torch.enable_grad():
        x = img.clone().requires_grad_()
        grams = kbc_style_embeddings(p, x.add(1).div(2), style_layers)
        losses = []
        for b in range(img.shape[0]):
            losses.append(sum(scale * feature_loss(g[b], t.reshape(t.shape[-2:])) for g, t in zip(grams, targets)))
        losses = torch.stack(losses)
        grad = torch.autograd.grad(losses.sum(), x)[0]
    return grad, losses.detach()


# ------------------------------------------------------------------------------------------ grad.py:178-196 LPIPSGrads
def lpips_normalize(f, eps=1e-10):
    return f / (f.pow(2).sum(1, keepdim=True).sqrt() + eps)


def init_lpips_lins(generator=None):
    """The ``lin`` layers' weights ([C] each, non-negative as in the released model)."""
    return [torch.rand(c, generator=generator) / c for c in LPIPS_CHNS]


def lpips_distance(p, lins, in0, in1):
    """lpips.LPIPS(net="vgg").forward(in0, in1) for inputs in [-1, 1] -> [B]."""
    f0 = vgg_features(p, VGG16_CFG, normalize_img(in0, LPIPS_SHIFT, LPIPS_SCALE), LPIPS_TAPS)
    f1 = vgg_features(p, VGG16_CFG, normalize_img(in1, LPIPS_SHIFT, LPIPS_SCALE), LPIPS_TAPS)
    val = 0
    for a, b, w in zip(f0, f1, lins):
        d = (lpips_normalize(a) - lpips_normalize(b)) ** 2
        val = val + (d * w.reshape(1, -1, 1, 1)).sum(1).mean((1, 2))
    return val


def lpips_grads(p, lins, img, target, scale=1.0):
    """LPIPSGrads.forward (grad.py:189-193) at 256 x 256, where ``resample(x, 256)`` is the identity (ops/image.py:214-240: no
    low-pass below the source size, bicubic interpolation with align_corners=True onto the same grid) -> (grad, distances [B])."""
    with torch.enable_grad():
        x = img.clone().requires_grad_()
        d = lpips_distance(p, lins, x, target.expand_as(x))
        grad = torch.autograd.grad(d.sum() * scale, x)[0]
    return grad, d.detach()


# ------------------------------------------------------------------------------------------ cutouts.py:101-206 DangoCutouts
def grayscale3(x):
    """torchvision.transforms.Grayscale(3) on a float tensor [.., 3, H, W]."""
    l = 0.2989 * x[..., 0:1, :, :] + 0.587 * x[..., 1:2, :, :] + 0.114 * x[..., 2:3, :, :]
    return l.expand(*x.shape[:-3], 3, *x.shape[-2:])


def dango_plan(sideY, sideX, cut_size, overview, inner_crop, ic_grey_p, cut_pow=1.0):
    """What DangoCutouts.forward (cutouts.py:154-206) draws for one call, as a list of (size, top, left, grey, flip) over the
    reflect-padded square for the overview cutouts (size = -1) and over the input for the inner crops - torch's GLOBAL generator,
    in the reference's order (one ``torch.rand([])`` and two ``torch.randint`` per inner crop)."""
    max_size = min(sideX, sideY)
    min_size = min(sideX, sideY, cut_size)
    plan = []
    if overview > 0:
        if overview <= 4:
            for k in range(overview):
                plan.append((-1, 0, 0, k in (1, 3), k in (2, 3)))
        else:
            plan += [(-1, 0, 0, False, False)] * overview
    for i in range(inner_crop):
        size = int(torch.rand([]) ** cut_pow * (max_size - min_size) + min_size)
        offsetx = int(torch.randint(0, sideX - size + 1, ()))
        offsety = int(torch.randint(0, sideY - size + 1, ()))
        plan.append((size, offsety, offsetx, i <= int(ic_grey_p * inner_crop), False))
    return plan


def dango_cutouts(input, plan, cut_size, resize):
    sideY, sideX = input.shape[2:4]
    max_size = min(sideX, sideY)
    pad = ((sideY - max_size) // 2, (sideY - max_size) // 2, (sideX - max_size) // 2, (sideX - max_size) // 2)
    pad_input = F.pad(input, pad, mode="reflect") if any(pad) else input
    outs = []
    for size, top, left, grey, flip in plan:
        src = pad_input if size < 0 else input[:, :, top:top + size, left:left + size]
        if size >= 0 and grey:
            src = grayscale3(src)
        c = resize(src, (cut_size, cut_size))       # (the reference's out_shape is [1, 3, cs, cs]: for one image per call, the spatial dimensions)
        if size < 0:
            if flip:
                c = c.flip(-1)
            if grey:
                c = grayscale3(c)
        outs.append(c)
    return torch.cat(outs)
