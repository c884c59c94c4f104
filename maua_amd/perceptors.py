"""VGG perceptors on the HIP device - the reference's ``maua/perceptors`` surface for the image-prompt grad modules.

Drop-in surface:
  * ``Perceptor`` / ``KBCPerceptor`` / ``load_perceptor``   <- maua/perceptors/__init__.py:10-101, vgg_kbc.py:10-71
    (``get_target_embeddings``, ``get_loss``; plus ``get_loss_grad``: the loss AND its gradient with respect to the image - what
    ``torch.autograd.grad(perceptor.get_loss(x, targets), x)`` gives the reference, maua/grad.py:90-93; there is no autograd here)
  * ``LPIPS``                                               <- ``lpips.LPIPS(net="vgg")`` as maua/grad.py:178-196 uses it
The networks run behind the C ABI (``maua_vgg_*``, csrc/perceptor.hip): forward, loss heads and the input gradient walked by hand.

torchvision and lpips are pip dependencies absent from /root/reference and from this image: vgg19 / vgg16 ``features`` (published
configurations "E" / "D"), LPIPS' ScalingLayer / taps / normalisation / ``lin`` layers are restated, **parity unpinned**; the
state-dict keys are torchvision's (``"<features index>.weight"``, also with a ``features.`` prefix) and lpips' (``lin<k>.model.1.weight``),
so released checkpoints load unchanged.  Without a checkpoint the networks are random-init (``allow_random_init=True``), like the UNet
and the CLIP tower of the bench.  "pgg" perceptors (vgg_pgg.py: caffe weights fetched with gdown) are not built; pooling other than
"max" raises.

One deviation, stated: the reference's ``gram_matrix`` folds the batch into the channel axis, so VGGGrads is only well-defined for
one image per call; here a batch is B independent images against the same targets (identical for B = 1).
"""
import ctypes as C
import math
import os

import numpy as np
import torch

from . import _lib as L

IMAGENET_MEAN = (0.485, 0.456, 0.406)     # vgg_kbc.py:33
IMAGENET_STD = (0.229, 0.224, 0.225)
LPIPS_SHIFT = (-0.030, -0.088, -0.188)    # lpips ScalingLayer: (x - shift) / scale
LPIPS_SCALE = (0.458, 0.448, 0.450)
VGG19_CFG = (64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M")
VGG16_CFG = (64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M")
LPIPS_TAPS = (3, 8, 15, 22, 29)           # relu1_2, relu2_2, relu3_3, relu4_3, relu5_3 of vgg16.features
LPIPS_CHNS = (64, 128, 256, 512, 512)


def features_plan(cfg, last_index):
    """torchvision's ``make_layers(cfg)[: last_index + 1]`` as the library's plan: ([channels | 0 = MaxPool2d(2)], features index of
    every plan entry - a convolution and its ReLU are one entry, indexed by the ReLU -, {weight key prefix: conv number})."""
    plan, idx, convs, i = [], [], {}, 0
    for v in cfg:
        if v == "M":
            if i > last_index:
                break
            plan.append(0); idx.append(i); i += 1
        else:
            if i + 1 > last_index:
                break
            convs[str(i)] = len(convs)
            plan.append(int(v)); idx.append(i + 1); i += 2
    return plan, idx, convs


class VGGFeatures(torch.nn.Module):
    """``torchvision.models.vgg*(...).features[: last_index + 1]`` behind ``maua_vgg_*``.  ``pre`` = (mul, add, mean, std): the network
    sees ((x * mul + add) - mean) / std."""

    def __init__(self, cfg, last_index, pre, replicate_first=False, dtype=torch.bfloat16, generator=None):
        super().__init__()
        if dtype not in (torch.bfloat16, torch.float32):
            raise ValueError("VGGFeatures: dtype must be torch.bfloat16 or torch.float32")
        self.cfg, self.last_index, self.pre, self.replicate_first, self.dtype = tuple(cfg), last_index, pre, bool(replicate_first), dtype
        self.plan, self.index, self.conv_of_key = features_plan(cfg, last_index)
        self._h = None
        self._dirty = True
        self._shape = None
        self._params = self._init_params(generator)

    def op_of(self, features_index):
        """Plan entry of a ``features`` index (a ReLU's or a pooling layer's)."""
        if features_index not in self.index:
            raise ValueError(f"features[{features_index}] is not a ReLU / pooling layer of this network (or lies beyond its last layer)")
        return self.index.index(features_index)

    def channels(self, features_index):
        op = self.op_of(features_index)
        while self.plan[op] == 0:
            op -= 1
        return self.plan[op]

    def _param_shapes(self):
        shapes, cin = {}, 3
        for key, _ in sorted(self.conv_of_key.items(), key=lambda kv: kv[1]):
            co = self.plan[self.index.index(int(key) + 1)]
            shapes[f"{key}.weight"] = (co, cin, 3, 3)
            shapes[f"{key}.bias"] = (co,)
            cin = co
        return shapes

    def _init_params(self, generator):
        g = generator or torch.Generator().manual_seed(0)
        p = {}
        with L.host_threads(1):
            for k, s in self._param_shapes().items():
                p[k] = torch.randn(s, generator=g) * (math.sqrt(2.0 / (s[1] * 9)) if k.endswith("weight") else 0.05)
        return p

    def state_dict(self, *a, **k):
        return {n: v.clone() for n, v in self._params.items()}

    def load_state_dict(self, sd, strict=True):
        """torchvision's keys ("0.weight" or "features.0.weight"); classifier / deeper feature layers are ignored."""
        sd = {(k[len("features."):] if k.startswith("features.") else k): v for k, v in sd.items()}
        shapes = self._param_shapes()
        missing = [k for k in shapes if k not in sd]
        if strict and missing:
            raise KeyError(f"VGGFeatures.load_state_dict: missing {missing[:4]}")
        for k, shape in shapes.items():
            if k in sd:
                v = torch.as_tensor(sd[k]).detach().float().cpu()
                if tuple(v.shape) != tuple(shape):
                    raise ValueError(f"VGGFeatures.load_state_dict: {k}: shape {tuple(v.shape)}, expected {tuple(shape)}")
                self._params[k] = v.contiguous()
        self._dirty = True
        return torch.nn.modules.module._IncompatibleKeys(missing, [])

    def eval(self):
        return self

    def requires_grad_(self, flag=True):
        return self

    def to(self, *a, **k):
        return self

    def _destroy(self):
        if self._h is not None:
            L.lib().maua_vgg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self._destroy()
        except Exception:
            pass

    def _handle(self):
        if self._h is None:
            h = C.c_void_p()
            plan = (C.c_int * len(self.plan))(*self.plan)
            mul, add, mean, std = self.pre
            L.check(L.lib().maua_vgg_create(L.ctx(), L.dtype_id(self.dtype), plan, len(self.plan), int(self.replicate_first), C.c_float(mul),
                                            C.c_float(add), (C.c_float * 3)(*mean), (C.c_float * 3)(*std), C.byref(h)))
            self._h = h
            self._dirty = True
        else:
            L.ctx()
        if self._dirty:
            for k, v in self._params.items():
                key, what = k.split(".")
                a = np.ascontiguousarray(v.numpy(), dtype=np.float32)
                L.check(L.lib().maua_vgg_load(self._h, self.conv_of_key[key], 0 if what == "weight" else 1, a.ctypes.data_as(C.c_void_p),
                                              C.c_size_t(a.size)))
            self._dirty = False
        return self._h

    # ------------------------------------------------------------------ forward and what it keeps
    def _check(self, x):
        x = L.dev_tensor(x, torch.float32)
        div = 1 << sum(1 for v in self.plan if v == 0)
        if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] % div or x.shape[3] % div:
            raise ValueError(f"VGGFeatures: expected [B, 3, H, W] with H, W multiples of {div}, got {tuple(x.shape)}")
        return x

    def forward(self, x, taps=None):
        """Runs the network (activations stay in the library); with ``taps`` (features indices) returns those activations [B, C, h, w]."""
        x = self._check(x)
        L.check(L.lib().maua_vgg_forward(self._handle(), L.ptr(x), x.shape[0], x.shape[2], x.shape[3]))
        self._shape = tuple(x.shape)
        if taps is None:
            return None
        return [self.features(t) for t in taps]

    def _grid(self, op):
        B, _, H, W = self._shape
        s = sum(1 for v in self.plan[:op + 1] if v == 0)
        return B, H >> s, W >> s

    def features(self, features_index):
        op = self.op_of(features_index)
        B, h, w = self._grid(op)
        out = torch.empty((B, self.channels(features_index), h, w), dtype=torch.float32, device="cuda")
        L.check(L.lib().maua_vgg_features(self._handle(), op, L.ptr(out)))
        return out

    def gram(self, features_index):
        op = self.op_of(features_index)
        c = self.channels(features_index)
        out = torch.empty((self._shape[0], c, c), dtype=torch.float32, device="cuda")
        L.check(L.lib().maua_vgg_gram(self._handle(), op, L.ptr(out)))
        return out

    def lpips_features(self, features_index):
        op = self.op_of(features_index)
        B, h, w = self._grid(op)
        out = torch.empty((B, h * w, self.channels(features_index)), dtype=torch.float32, device="cuda")
        L.check(L.lib().maua_vgg_lpips_features(self._handle(), op, L.ptr(out)))
        return out


def _ptr_array(tensors):
    return (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


# ------------------------------------------------------------------------------------------------ maua/perceptors/__init__.py
class Perceptor(torch.nn.Module):
    """perceptors/__init__.py:10-91.  Content layers (the feature maps themselves) are not built: VGGGrads constructs its perceptor
    with ``content_layers=[]`` and ``content_strength=0`` (grad.py:76)."""

    def __init__(self, content_strength, content_layers, style_strength, style_layers) -> None:
        super().__init__()
        if content_layers:
            raise NotImplementedError("Perceptor: content layers are not part of this build (VGGGrads uses style layers only, grad.py:76)")
        self.content_layers, self.style_layers = list(content_layers), list(style_layers)
        self.content_strength, self.style_strength = content_strength, style_strength

    def get_target_embeddings(self, contents=None, styles=None, content_weights=None, style_weights=None):
        """:44-76 for style images: the weighted mean of their Gram matrices, one [C, C] device tensor per style layer."""
        if contents is not None:
            raise NotImplementedError("Perceptor.get_target_embeddings: content targets are not part of this build")
        if styles is None:
            return None
        if isinstance(styles, torch.Tensor):
            styles = [styles]
        if style_weights is None:
            style_weights = torch.ones(len(styles))
        style_weights = torch.as_tensor(style_weights, dtype=torch.float32)
        style_weights = style_weights / style_weights.sum()
        out = None
        for style, sw in zip(styles, style_weights):
            style = L.dev_tensor(style, torch.float32)
            if style.dim() == 3:
                style = style.unsqueeze(0)
            self.net.forward(self._pre_undo(style))
            grams = [self.net.gram(l).mean(0) * float(sw) for l in self.style_layers]   # (one style image per entry: B = 1)
            out = grams if out is None else [a + b for a, b in zip(out, grams)]
        return out

    def _pre_undo(self, x01):
        """The library folds ``img.add(1).div(2)`` into the network's first kernel (VGGGrads hands it the sampler's image in [-1, 1]);
        ``get_target_embeddings`` / ``get_loss`` take images in [0, 1] like the reference's, so map them back."""
        return x01 * 2 - 1

    def forward(self, x):
        """x in [0, 1] -> the embeddings list (Gram matrices [B, C, C] of the style layers)."""
        self.net.forward(self._pre_undo(L.dev_tensor(x, torch.float32)))
        return [self.net.gram(l) for l in self.style_layers]

    def get_loss_grad(self, x, targets, from_unit_range=True):
        """(losses [B], d sum(losses) / d x): ``get_loss`` (:83-91) and the gradient VGGGrads takes of it, in one library call.
        ``from_unit_range``: x is in [0, 1] (the reference's ``get_loss`` argument) - the gradient is with respect to that x; False: x is
        the sampler's image in [-1, 1] and the gradient is with respect to it (VGGGrads.forward)."""
        x = L.dev_tensor(x, torch.float32)
        assert len(targets) == len(self.style_layers), \
            f"The target embeddings don't match this perceptor's embeddings: {len(targets)}. Expected: {len(self.style_layers)}"
        img = self._pre_undo(x) if from_unit_range else x
        img = self.net._check(img)
        B, _, H, W = img.shape
        tg = [L.dev_tensor(t, torch.float32).contiguous() for t in targets]
        strides = []
        for t, l in zip(tg, self.style_layers):
            c = self.net.channels(l)
            if tuple(t.shape) == (c, c):
                strides.append(0)
            elif tuple(t.shape) == (B, c, c):
                strides.append(c * c)
            else:
                raise ValueError(f"style target of features[{l}]: shape {tuple(t.shape)}, expected ({c}, {c}) or ({B}, {c}, {c})")
        taps = (C.c_int * len(tg))(*[self.net.op_of(l) for l in self.style_layers])
        grad = torch.empty_like(img)
        loss = torch.empty(B, dtype=torch.float32, device=img.device)
        L.check(L.lib().maua_vgg_style_grad(self.net._handle(), L.ptr(img), B, H, W, taps, len(tg), _ptr_array(tg),
                                            (C.c_long * len(tg))(*strides), C.c_float(float(self.style_strength)), L.ptr(grad), L.ptr(loss)))
        self.net._shape = tuple(img.shape)
        if from_unit_range:
            grad = grad * 2.0       # d img / d x = 2
        return loss, grad

    def get_loss(self, x, targets):
        """:83-91 -> the summed loss (a device scalar; no graph behind it - use ``get_loss_grad`` for the gradient)."""
        return self.get_loss_grad(x, targets)[0].sum()


class KBCPerceptor(Perceptor):
    """vgg_kbc.py:10-71 - VGG19 by Katherine Crowson: replicate padding on the first convolution, ImageNet normalisation."""

    pooling_scales = {"max": 1.0, "avg": 2.0, "l2": 0.78}

    def __init__(self, content_layers=None, style_layers=None, content_strength=1, style_strength=1, pooling="max", dtype=torch.bfloat16,
                 state_dict=None, allow_random_init=False, generator=None):
        if content_layers is None:
            content_layers = [22]
        if style_layers is None:
            style_layers = [1, 6, 11, 20, 29]
        if pooling != "max":
            raise NotImplementedError(f'KBCPerceptor(pooling="{pooling}"): only the default max pooling is built')
        if content_layers and content_strength == 0:
            content_layers = []           # (their loss would be multiplied by zero)
        super().__init__(content_strength, content_layers, style_strength, style_layers)
        self.net = VGGFeatures(VGG19_CFG, max(list(content_layers) + list(style_layers)), (0.5, 0.5, IMAGENET_MEAN, IMAGENET_STD),
                               replicate_first=True, dtype=dtype, generator=generator)
        _load_weights(self.net, state_dict, "vgg19", allow_random_init)


def _load_weights(net, state_dict, name, allow_random_init):
    if state_dict is None:
        hub = os.path.expanduser(os.path.join(os.environ.get("TORCH_HOME", "~/.cache/torch"), "hub", "checkpoints"))
        if os.path.isdir(hub):
            for f in sorted(os.listdir(hub)):
                if f.startswith(name + "-") and f.endswith(".pth"):      # torchvision's cache name, e.g. vgg19-dcbb9e9d.pth
                    state_dict = torch.load(os.path.join(hub, f), map_location="cpu", weights_only=True)
                    break
    if state_dict is not None:
        net.load_state_dict(state_dict, strict=True)
    elif not allow_random_init:
        raise FileNotFoundError(f"no weights for {name}: pass state_dict=..., place torchvision's checkpoint in ~/.cache/torch/hub/checkpoints, "
                                "or allow_random_init=True for a synthetic network")


def load_perceptor(name: str):
    """perceptors/__init__.py:97-101."""
    from functools import partial
    if name.startswith("pgg"):
        raise NotImplementedError('load_perceptor("pgg-..."): the caffe VGG perceptors (vgg_pgg.py, weights fetched with gdown) are not built')
    if name.startswith("kbc"):
        return KBCPerceptor
    raise Exception(f"Perceptor {name} not recognized!")


# ------------------------------------------------------------------------------------------------ lpips.LPIPS(net="vgg")
class LPIPS(torch.nn.Module):
    """``lpips.LPIPS(net="vgg", verbose=False)`` as LPIPSGrads uses it: ``__call__(in0, in1)`` -> distances [B, 1, 1, 1] for images in
    [-1, 1]; ``distance_grad``: the distances and d (scale * sum) / d in0."""

    def __init__(self, net="vgg", verbose=False, dtype=torch.bfloat16, state_dict=None, lin_state_dict=None, allow_random_init=False,
                 generator=None):
        super().__init__()
        if net != "vgg":
            raise NotImplementedError(f'LPIPS(net="{net}"): only the "vgg" variant (what LPIPSGrads constructs) is built')
        self.net = VGGFeatures(VGG16_CFG, LPIPS_TAPS[-1], (1.0, 0.0, LPIPS_SHIFT, LPIPS_SCALE), dtype=dtype, generator=generator)
        _load_weights(self.net, state_dict, "vgg16", allow_random_init)
        g = generator or torch.Generator().manual_seed(1)
        self.lins = [torch.rand(c, generator=g) / c for c in LPIPS_CHNS]
        if lin_state_dict is not None:
            for k in range(5):
                self.lins[k] = torch.as_tensor(lin_state_dict[f"lin{k}.model.1.weight"]).detach().float().reshape(-1).cpu()
        elif not allow_random_init:
            raise FileNotFoundError("no LPIPS lin weights: pass lin_state_dict=... (lpips' weights/v0.1/vgg.pth) or allow_random_init=True")
        self._lins_dev = None
        self._target = None

    def _lins(self):
        if self._lins_dev is None:
            self._lins_dev = [L.dev_tensor(w, torch.float32).contiguous() for w in self.lins]
        return self._lins_dev

    def embed(self, x):
        """The unit-normalised tap features of x [B, 3, H, W] in [-1, 1]: what a distance to x needs of it."""
        self.net.forward(x)
        return [self.net.lpips_features(t) for t in LPIPS_TAPS]

    def distance_grad(self, in0, target_feats, scale=1.0):
        x = self.net._check(in0)
        B, _, H, W = x.shape
        strides = []
        for f in target_feats:
            if f.shape[0] not in (1, B):
                raise ValueError("LPIPS: the target is one image or one per sample")
            strides.append(0 if f.shape[0] == 1 else f.shape[1] * f.shape[2])
        taps = (C.c_int * 5)(*[self.net.op_of(t) for t in LPIPS_TAPS])
        grad = torch.empty_like(x)
        dist = torch.empty(B, dtype=torch.float32, device=x.device)
        L.check(L.lib().maua_vgg_lpips_grad(self.net._handle(), L.ptr(x), B, H, W, taps, 5, _ptr_array(target_feats), (C.c_long * 5)(*strides),
                                            _ptr_array(self._lins()), C.c_float(float(scale)), L.ptr(grad), L.ptr(dist)))
        self.net._shape = tuple(x.shape)
        return dist, grad

    def forward(self, in0, in1):
        feats = self.embed(in1)
        return self.distance_grad(in0, feats)[0].reshape(-1, 1, 1, 1)
