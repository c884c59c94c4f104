"""CPU oracle (TEST INFRASTRUCTURE, never imported by the product) of the reference's text-prompt guidance: ``CLIPGrads``.

Restates, in torch-CPU float32 with autograd:
  * maua/grad.py:96-165      CLIPGrads.set_targets / forward (cutout batches, spherical distance to the target embeddings, weights,
                             ``torch.autograd.grad`` back to the image, clamp_gradient)
  * maua/ops/cutouts.py:8-50 random_cutouts / MauaCutouts (pinned by tests/golden/g33_cutouts.npz: the reference's own function
                             run with this file's ``resize`` standing in for the absent ``resize_right`` package)
  * maua/loss.py:22-25       spherical_dist_loss (pinned by g33: the reference's own function)
and two third-party pieces that are ABSENT from /root/reference and from this image - **parity unpinned**, restated from their
published algorithms:
  * ``resize_right.resize`` (setup.py:88 "resize_right", unpinned version; Shocher, "ResizeRight", the algorithm of "From Discrete to
    Continuous Convolution Layers"): 1-D passes, cubic kernel (a = -0.5), antialiasing = kernel stretched by 1 / scale when shrinking,
    weights normalised per output sample, zero ("constant") padding - the defaults random_cutouts calls it with;
  * OpenAI CLIP's ``VisionTransformer`` (setup.py:37 "clip @ git+https://github.com/OpenAI/CLIP", unpinned; clip/model.py): patch
    convolution, class token, positional embedding, ln_pre, pre-LN residual attention blocks (nn.MultiheadAttention, QuickGELU MLP),
    ln_post on the class token, projection.  State-dict keys are CLIP's (``visual.*``), so a released checkpoint's image tower loads
    unchanged; ``tests/test_oracle_clip.py`` checks the attention restatement against ``torch.nn.MultiheadAttention``.
"""
import math

import torch
import torch.nn.functional as F

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)    # grad.py:110
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


# ------------------------------------------------------------------------------------------ resize_right (published algorithm)
def cubic(x):
    """resize_right/interp_methods.py ``cubic`` (support 4)."""
    absx = x.abs()
    absx2, absx3 = absx ** 2, absx ** 3
    return ((1.5 * absx3 - 2.5 * absx2 + 1.0) * (absx <= 1.0).to(x.dtype)
            + (-0.5 * absx3 + 2.5 * absx2 - 4.0 * absx + 2.0) * ((1.0 < absx) & (absx <= 2.0)).to(x.dtype))


def resize_tables(in_sz, out_sz):
    """One dimension of ``resize(..., out_shape=...)`` with the defaults (cubic, antialiasing, not by_convs):
    -> (left [out_sz] int64: first input index of each output sample's field of view - may be negative / run past in_sz: zero padding,
        weights [out_sz, taps] float32, normalised per output sample)."""
    eps = torch.finfo(torch.float32).eps
    scale = out_sz / in_sz
    out_coordinates = torch.arange(out_sz)
    projected = out_coordinates / float(scale) + (in_sz - 1) / 2 - (out_sz - 1) / (2 * float(scale))   # float32 tensor
    support = 4.0
    if scale < 1.0:
        method = lambda a: scale * cubic(scale * a)
        support = support / scale
    else:
        method = cubic
    left = torch.ceil(projected - support / 2 - eps).long()
    ordinal = torch.arange(math.ceil(support - eps))
    fov = left[:, None] + ordinal
    w = method(projected[:, None] - fov)
    s = w.sum(1, keepdim=True)
    s[s == 0] = 1
    return left, (w / s).float()


def resize(x, out_shape):
    """``resize_right.resize(x, out_shape=out_shape)`` on the last two dimensions: rows first, then columns (equal scale factors keep
    the dimension order); a dimension whose size does not change is skipped."""
    for dim, out_sz in ((-2, out_shape[0]), (-1, out_shape[1])):
        in_sz = x.shape[dim]
        if in_sz == out_sz:
            continue
        left, w = resize_tables(in_sz, out_sz)
        taps = w.shape[1]
        idx = left[:, None] + torch.arange(taps)                    # [out, taps]
        ok = ((idx >= 0) & (idx < in_sz)).to(x.dtype)
        nb = x.movedim(dim, -1)[..., idx.clamp(0, in_sz - 1)]       # [..., out, taps]
        x = (nb * (w.to(x.dtype) * ok)).sum(-1).movedim(-1, dim)
    return x


# ------------------------------------------------------------------------------------------ cutouts (maua/ops/cutouts.py:8-50)
def cutout_rects(sideY, sideX, cut_size, cutn, cut_pow):
    """The rectangles random_cutouts takes, in its order and with its draws from torch's GLOBAL generator (:30-33: one ``torch.rand([])``
    and one ``torch.randint`` per random cutout) -> list of (size, offsety, offsetx)."""
    max_size = min(sideX, sideY)
    min_size = min(sideX, sideY, cut_size)
    if sideY < sideX:
        size = sideY
        tops = torch.zeros(cutn // 4, dtype=int)
        lefts = torch.linspace(0, sideX - size, cutn // 4, dtype=int)
    else:
        size = sideX
        tops = torch.linspace(0, sideY - size, cutn // 4, dtype=int)
        lefts = torch.zeros(cutn // 4, dtype=int)
    rects = [(int(size), int(oy), int(ox)) for oy, ox in zip(tops, lefts)]
    for _ in range(cutn - len(rects)):
        size = (torch.rand([]) ** cut_pow * max_size).clamp(min_size, max_size).round().long().item()
        loc = torch.randint(0, (sideX - size + 1) * (sideY - size + 1), ())
        oy, ox = torch.div(loc, (sideX - size + 1), rounding_mode="floor"), loc % (sideX - size + 1)
        rects.append((int(size), int(oy), int(ox)))
    return rects


def cutouts_from_rects(input, rects, cut_size):
    """:24-38 given the rectangles: every cutout resized to cut_size^2, concatenated along the batch (cutout-major)."""
    return torch.cat([resize(input[:, :, oy:oy + s, ox:ox + s], (cut_size, cut_size)) for s, oy, ox in rects])


def maua_cutouts_pow(t, pow_gain=16.0):
    """MauaCutouts.forward :47 with ``t`` as CLIPGrads hands it over (grad.py:149 ``t[[0]].long()``: a one-element int64 tensor), so
    the schedule is float32 tensor arithmetic -> one-element float32 tensor."""
    t = torch.as_tensor(t).reshape(-1)[:1].long()
    return pow_gain ** ((500 - t) / 500)


def random_cutouts(input, cut_size=224, cutn=32, cut_pow=1.0):
    sideY, sideX = input.shape[-2:]
    return cutouts_from_rects(input, cutout_rects(sideY, sideX, cut_size, cutn, cut_pow), cut_size)


# ------------------------------------------------------------------------------------------ loss (maua/loss.py:22-25)
def spherical_dist_loss(x, y):
    x = F.normalize(x, dim=-1)
    y = F.normalize(y, dim=-1)
    return (x - y).norm(dim=-1).div(2).arcsin().pow(2).mul(2)


# ------------------------------------------------------------------------------------------ CLIP image tower (clip/model.py)
def vit_config(input_resolution=224, patch_size=16, width=768, layers=12, heads=12, output_dim=512):
    """ViT-B/16 by default (grad.py:100 perceptors=["ViT-B/16"])."""
    return dict(input_resolution=input_resolution, patch_size=patch_size, width=width, layers=layers, heads=heads, output_dim=output_dim)


def vit_param_shapes(cfg):
    w, p, L, E = cfg["width"], cfg["patch_size"], cfg["layers"], cfg["output_dim"]
    n_tok = (cfg["input_resolution"] // p) ** 2 + 1
    shapes = {"visual.conv1.weight": (w, 3, p, p), "visual.class_embedding": (w,), "visual.positional_embedding": (n_tok, w),
              "visual.ln_pre.weight": (w,), "visual.ln_pre.bias": (w,)}
    for i in range(L):
        b = f"visual.transformer.resblocks.{i}."
        shapes.update({b + "attn.in_proj_weight": (3 * w, w), b + "attn.in_proj_bias": (3 * w,), b + "attn.out_proj.weight": (w, w),
                       b + "attn.out_proj.bias": (w,), b + "ln_1.weight": (w,), b + "ln_1.bias": (w,),
                       b + "mlp.c_fc.weight": (4 * w, w), b + "mlp.c_fc.bias": (4 * w,), b + "mlp.c_proj.weight": (w, 4 * w),
                       b + "mlp.c_proj.bias": (w,), b + "ln_2.weight": (w,), b + "ln_2.bias": (w,)})
    shapes.update({"visual.ln_post.weight": (w,), "visual.ln_post.bias": (w,), "visual.proj": (w, E)})
    return shapes


def init_vit_params(cfg, generator=None):
    """Random parameters with the scales of clip/model.py (VisionTransformer.__init__, CLIP.initialize_parameters); LayerNorm gains
    and all biases get a perturbation so that tests see them."""
    g = generator or torch.Generator().manual_seed(0)
    w, L = cfg["width"], cfg["layers"]
    scale = w ** -0.5
    proj_std, attn_std, fc_std = scale * ((2 * L) ** -0.5), scale, (2 * w) ** -0.5
    p = {}
    for name, shape in vit_param_shapes(cfg).items():
        r = torch.randn(shape, generator=g)
        if name.endswith("conv1.weight"):
            p[name] = r / math.sqrt(3 * cfg["patch_size"] ** 2)
        elif name.endswith(("class_embedding", "positional_embedding", "visual.proj")):
            p[name] = scale * r
        elif ".ln_" in name or "ln_pre" in name or "ln_post" in name:
            p[name] = 1 + 0.1 * r if name.endswith("weight") else 0.1 * r
        elif name.endswith("bias"):
            p[name] = 0.02 * r
        elif name.endswith("in_proj_weight"):
            p[name] = attn_std * r
        elif name.endswith(("out_proj.weight", "c_proj.weight")):
            p[name] = proj_std * r
        else:   # c_fc.weight
            p[name] = fc_std * r
    return p


def layer_norm(x, w, b):
    """clip/model.py LayerNorm: float32 statistics, eps 1e-5."""
    return F.layer_norm(x.float(), (x.shape[-1],), w, b, 1e-5)


def attention(x, in_w, in_b, out_w, out_b, heads):
    """nn.MultiheadAttention(d, heads)(x, x, x, need_weights=False) for batch-first x [N, T, d]: packed in-projection [q | k | v],
    head h = channels [h * d / heads, (h + 1) * d / heads), q scaled by (d / heads) ** -0.5, softmax over keys, out-projection."""
    N, T, d = x.shape
    hd = d // heads
    qkv = x @ in_w.t() + in_b
    q, k, v = (t.reshape(N, T, heads, hd).transpose(1, 2) for t in qkv.split(d, dim=-1))
    a = torch.softmax((q * hd ** -0.5) @ k.transpose(-1, -2), dim=-1) @ v
    return a.transpose(1, 2).reshape(N, T, d) @ out_w.t() + out_b


def quick_gelu(x):
    return x * torch.sigmoid(1.702 * x)


def encode_image(p, cfg, x):
    """VisionTransformer.forward: x [N, 3, R, R] (already normalised) -> [N, output_dim]."""
    w = cfg["width"]
    x = F.conv2d(x, p["visual.conv1.weight"], stride=cfg["patch_size"])
    x = x.reshape(x.shape[0], w, -1).permute(0, 2, 1)
    x = torch.cat([p["visual.class_embedding"] + torch.zeros(x.shape[0], 1, w), x], dim=1)
    x = x + p["visual.positional_embedding"]
    x = layer_norm(x, p["visual.ln_pre.weight"], p["visual.ln_pre.bias"])
    for i in range(cfg["layers"]):
        b = f"visual.transformer.resblocks.{i}."
        x = x + attention(layer_norm(x, p[b + "ln_1.weight"], p[b + "ln_1.bias"]), p[b + "attn.in_proj_weight"], p[b + "attn.in_proj_bias"],
                          p[b + "attn.out_proj.weight"], p[b + "attn.out_proj.bias"], cfg["heads"])
        h = layer_norm(x, p[b + "ln_2.weight"], p[b + "ln_2.bias"]) @ p[b + "mlp.c_fc.weight"].t() + p[b + "mlp.c_fc.bias"]
        x = x + quick_gelu(h) @ p[b + "mlp.c_proj.weight"].t() + p[b + "mlp.c_proj.bias"]
    x = layer_norm(x[:, 0, :], p["visual.ln_post.weight"], p["visual.ln_post.bias"])
    return x @ p["visual.proj"]


def normalize(x):
    """torchvision Normalize(mean, std) of grad.py:110."""
    m = torch.tensor(CLIP_MEAN).reshape(1, 3, 1, 1)
    s = torch.tensor(CLIP_STD).reshape(1, 3, 1, 1)
    return (x - m) / s


# ------------------------------------------------------------------------------------------ CLIPGrads (maua/grad.py:96-165)
def normalise_weights(weights):
    """:139-143"""
    w = torch.as_tensor(weights, dtype=torch.float)
    if w.sum().abs() < 1e-3:
        raise RuntimeError("The weights must not sum to 0.")
    return w / w.sum().abs()


def clip_grads(p, cfg, img, rects_per_batch, target, weights, scale=1.0, clamp_gradient=None):
    """CLIPGrads.forward (:145-159) for one perceptor, with the cutout rectangles of every cutout batch handed in
    (``rects_per_batch``: list of lists of (size, oy, ox), what ``cutout_rects`` draws inside MauaCutouts).
    img [B, 3, H, W] in [-1, 1]; target [P, E]; weights [P] (already normalised).  -> d loss / d img."""
    cut_size = cfg["input_resolution"]
    n_batches = len(rects_per_batch)
    grad = torch.zeros_like(img)
    for rects in rects_per_batch:
        with torch.enable_grad():
            x = img.clone().requires_grad_()
            cuts = cutouts_from_rects(x.add(1).div(2), rects, cut_size)
            image_embeds = encode_image(p, cfg, normalize(cuts)).float()
            dists = spherical_dist_loss(image_embeds.unsqueeze(1), target.unsqueeze(0))
            loss = dists.view((-1, img.shape[0], dists.shape[-1])).mul(weights).sum(2).mean(0)
            grad += torch.autograd.grad(loss.sum() * scale, x)[0] / n_batches
    if clamp_gradient:
        magnitude = grad.square().mean().sqrt()
        grad = grad * (magnitude.clamp(max=clamp_gradient) / magnitude)
    return grad
