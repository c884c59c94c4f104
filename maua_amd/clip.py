"""CLIP's image tower on the HIP device: ``VisionTransformer`` with OpenAI CLIP's constructor arguments and state-dict keys.

Drop-in for what maua/grad.py:96-165 (``CLIPGrads``) needs of ``clip.load(name)[0]``: ``visual.input_resolution`` and
``encode_image``.  ``clip`` (setup.py:37, "clip @ git+https://github.com/OpenAI/CLIP") is a pip dependency that is absent from
/root/reference and from this image: the published architecture (clip/model.py ``VisionTransformer`` / ``ResidualAttentionBlock`` /
``QuickGELU`` / fp32 ``LayerNorm``) is restated, **parity unpinned**; the keys are CLIP's, so the ``visual.*`` half of a released
checkpoint loads unchanged.  The network runs behind the C ABI (``maua_clip_*``, csrc/clip.hip); ``vjp`` is the input gradient the
library evaluates by walking the network backwards (what ``torch.autograd.grad`` gives the reference).

The text tower is not here (no tokenizer vocabulary, no weights in the image): text prompts enter as precomputed embeddings
(``maua_amd.grad.EmbeddingPrompt``) or through a caller-supplied ``text_encoder``.
"""
import ctypes as C
import math
import os

import numpy as np
import torch

from . import _lib as L

# name -> (input_resolution, patch_size, width, layers, heads, output_dim): clip/model.py build_model on the released checkpoints
VISION_CONFIGS = {
    "ViT-B/32": (224, 32, 768, 12, 12, 512),
    "ViT-B/16": (224, 16, 768, 12, 12, 512),
}
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)   # grad.py:110
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


class VisionTransformer(torch.nn.Module):
    """clip/model.py VisionTransformer(input_resolution, patch_size, width, layers, heads, output_dim).
    ``dtype``: torch.bfloat16 (default; the reference runs the perceptor in fp16) or torch.float32 (exact-f32 MFMA parity mode)."""

    def __init__(self, input_resolution, patch_size, width, layers, heads, output_dim, dtype=torch.bfloat16, generator=None):
        super().__init__()
        if dtype not in (torch.bfloat16, torch.float32):
            raise ValueError("VisionTransformer: dtype must be torch.bfloat16 or torch.float32")
        self.input_resolution, self.patch_size, self.width, self.layers, self.heads, self.output_dim = \
            input_resolution, patch_size, width, layers, heads, output_dim
        self.dtype = dtype
        self._h = None
        self._dirty = True
        self._kept = 0
        self._params = self._init_params(generator)

    # ------------------------------------------------------------------ parameters (CLIP's keys below "visual.")
    def _param_shapes(self):
        w, p, E = self.width, self.patch_size, self.output_dim
        n_tok = (self.input_resolution // p) ** 2 + 1
        shapes = {"conv1.weight": (w, 3, p, p), "class_embedding": (w,), "positional_embedding": (n_tok, w),
                  "ln_pre.weight": (w,), "ln_pre.bias": (w,)}
        for i in range(self.layers):
            b = f"transformer.resblocks.{i}."
            shapes.update({b + "attn.in_proj_weight": (3 * w, w), b + "attn.in_proj_bias": (3 * w,),
                           b + "attn.out_proj.weight": (w, w), b + "attn.out_proj.bias": (w,), b + "ln_1.weight": (w,), b + "ln_1.bias": (w,),
                           b + "mlp.c_fc.weight": (4 * w, w), b + "mlp.c_fc.bias": (4 * w,), b + "mlp.c_proj.weight": (w, 4 * w),
                           b + "mlp.c_proj.bias": (w,), b + "ln_2.weight": (w,), b + "ln_2.bias": (w,)})
        shapes.update({"ln_post.weight": (w,), "ln_post.bias": (w,), "proj": (w, E)})
        return shapes

    def _init_params(self, generator):
        """clip/model.py: VisionTransformer.__init__ (scale = width ** -0.5) and CLIP.initialize_parameters' stds for the blocks."""
        g = generator or torch.Generator().manual_seed(0)
        w, Ls = self.width, self.layers
        scale = w ** -0.5
        proj_std, attn_std, fc_std = scale * ((2 * Ls) ** -0.5), scale, (2 * w) ** -0.5
        p = {}
        with L.host_threads(1):
            for name, shape in self._param_shapes().items():
                if ".ln_" in name or name.startswith("ln_"):
                    p[name] = torch.ones(shape) if name.endswith("weight") else torch.zeros(shape)
                elif name.endswith("bias"):
                    p[name] = torch.zeros(shape)
                else:
                    r = torch.randn(shape, generator=g)
                    if name == "conv1.weight":
                        p[name] = r / math.sqrt(3 * self.patch_size ** 2)    # nn.Conv2d's default scale
                    elif name in ("class_embedding", "positional_embedding", "proj"):
                        p[name] = scale * r
                    elif name.endswith("in_proj_weight"):
                        p[name] = attn_std * r
                    elif name.endswith(("out_proj.weight", "c_proj.weight")):
                        p[name] = proj_std * r
                    else:
                        p[name] = fc_std * r
        return p

    def state_dict(self, *a, **k):
        return {n: v.clone() for n, v in self._params.items()}

    def load_state_dict(self, sd, strict=True):
        """Takes the tower's own keys or a whole CLIP state dict (keys below ``visual.`` are used, the text tower's ignored)."""
        if any(k.startswith("visual.") for k in sd):
            sd = {k[len("visual."):]: v for k, v in sd.items() if k.startswith("visual.")}
        shapes = self._param_shapes()
        missing = [k for k in shapes if k not in sd]
        unexpected = [k for k in sd if k not in shapes]
        if strict and (missing or unexpected):
            raise KeyError(f"VisionTransformer.load_state_dict: missing {missing[:4]}, unexpected {unexpected[:4]}")
        for k, shape in shapes.items():
            if k in sd:
                v = torch.as_tensor(sd[k]).detach().float().cpu()
                if tuple(v.shape) != tuple(shape):
                    raise ValueError(f"VisionTransformer.load_state_dict: {k}: shape {tuple(v.shape)}, expected {tuple(shape)}")
                self._params[k] = v.contiguous()
        self._dirty = True
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    def eval(self):
        return self

    def requires_grad_(self, flag=True):
        return self

    def to(self, *a, **k):
        return self

    def float(self):
        return self

    def _destroy(self):
        if self._h is not None:
            L.lib().maua_clip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self._destroy()
        except Exception:
            pass

    def _handle(self):
        if self._h is None:
            h = C.c_void_p()
            L.check(L.lib().maua_clip_create(L.ctx(), self.input_resolution, self.patch_size, self.width, self.layers, self.heads,
                                             self.output_dim, L.dtype_id(self.dtype), C.byref(h)))
            self._h = h
            self._dirty = True
        else:
            L.ctx()   # (rebinds the context to torch's current stream)
        if self._dirty:
            for k, v in self._params.items():
                a = np.ascontiguousarray(v.numpy(), dtype=np.float32)
                L.check(L.lib().maua_clip_load(self._h, k.encode(), a.ctypes.data_as(C.c_void_p), C.c_size_t(a.size)))
            self._dirty = False
        return self._h

    # ------------------------------------------------------------------ forward / input gradient
    def forward(self, x, keep=False):
        """x [N, 3, R, R] (already normalised) -> [N, output_dim] float32 on the device."""
        x = L.dev_tensor(x, torch.float32)
        R = self.input_resolution
        if x.dim() != 4 or tuple(x.shape[1:]) != (3, R, R):
            raise ValueError(f"VisionTransformer: expected [N, 3, {R}, {R}], got {tuple(x.shape)}")
        out = torch.empty((x.shape[0], self.output_dim), dtype=torch.float32, device=x.device)
        L.check(L.lib().maua_clip_encode_image(self._handle(), L.ptr(x), x.shape[0], int(bool(keep)), L.ptr(out)))
        self._kept = x.shape[0] if keep else 0
        return out

    def vjp(self, d_embeds):
        """(d embeds / d x)^T d_embeds for the images of the last ``forward(x, keep=True)`` -> [N, 3, R, R] float32."""
        g = L.dev_tensor(d_embeds, torch.float32)
        if self._kept == 0 or tuple(g.shape) != (self._kept, self.output_dim):
            raise ValueError("VisionTransformer.vjp: call forward(x, keep=True) first; d_embeds is [N, output_dim]")
        R = self.input_resolution
        out = torch.empty((self._kept, 3, R, R), dtype=torch.float32, device=g.device)
        L.check(L.lib().maua_clip_encode_image_vjp(self._handle(), L.ptr(g), self._kept, L.ptr(out)))
        return out


class CLIPImageModel(torch.nn.Module):
    """What ``clip.load(name, jit=False)[0]`` is to CLIPGrads: ``.visual`` (with ``input_resolution``) and ``encode_image``.
    ``encode_text`` exists only when a ``text_encoder`` (tokens or strings -> [n, output_dim]) is supplied."""

    def __init__(self, visual, text_encoder=None):
        super().__init__()
        self.visual = visual
        self.text_encoder = text_encoder

    def encode_image(self, image):
        return self.visual(image)

    def encode_text(self, text):
        if self.text_encoder is None:
            raise NotImplementedError(
                "CLIPImageModel.encode_text: the text tower is not part of this build (no tokenizer vocabulary or weights in the "
                "image); pass text prompts as maua_amd.grad.EmbeddingPrompt, or give CLIPImageModel a text_encoder")
        return self.text_encoder(text)

    def eval(self):
        return self

    def requires_grad_(self, flag=True):
        return self


def load(name, jit=False, dtype=torch.bfloat16, state_dict=None, allow_random_init=False, generator=None, text_encoder=None):
    """``clip.load(name, jit=False)`` for the image towers this build has -> (model, preprocess=None).  Weights: ``state_dict`` (a CLIP
    state dict or its ``visual.*`` half), else the file CLIP's own loader caches (~/.cache/clip/<name>.pt, TorchScript archive or state
    dict), else - only with ``allow_random_init`` - CLIP's own initialisation (benchmarks: there is no network for checkpoints)."""
    if name not in VISION_CONFIGS:
        raise NotImplementedError(f"perceptor {name!r}: this build has the ViT image towers {sorted(VISION_CONFIGS)} "
                                  "(ResNet towers and ViT-L/14's 14-pixel patches are not built)")
    vt = VisionTransformer(*VISION_CONFIGS[name], dtype=dtype, generator=generator)
    if state_dict is None:
        path = os.path.expanduser(f"~/.cache/clip/{name.replace('/', '-')}.pt")
        if os.path.exists(path):
            try:
                state_dict = torch.jit.load(path, map_location="cpu").state_dict()
            except RuntimeError:
                state_dict = torch.load(path, map_location="cpu", weights_only=True)
    if state_dict is not None:
        vt.load_state_dict(state_dict, strict=False)
    elif not allow_random_init:
        raise FileNotFoundError(f"no weights for {name}: pass state_dict=..., place CLIP's checkpoint in ~/.cache/clip/, or "
                                "allow_random_init=True for a synthetic tower")
    return CLIPImageModel(vt, text_encoder), None
