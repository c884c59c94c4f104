"""Checkpoint import for the render path — mirrors maua/GAN/load.py.

    load_network(path, for_inference=False) -> Generator           (maua/GAN/load.py:191-207)
    load_rosinality2ada(path, blur_scale=4.0, for_inference=False)  (:18-127)
    load_nvidia_pt(path, z_dim=512, ..., for_inference=False)       (:167-189)
    load_nvidia(path, for_inference=None)                           (:130-164)

What the reference's ``for_inference`` flag selects is which network class runs the weights: ``False`` = the
un-vendored NVIDIA training network (upstream semantics: kernel flipped before the transposed convolution,
``x @ w.T`` in the mapping MLP, noise scaled by the learned ``noise_strength``), ``True`` = the in-tree inference
network (no flip, ``x @ w``, unscaled noise).  Here both run on the same HIP kernels and the
flag becomes ``nv_compat = not for_inference``.

Pure host-side key/layout work: the converters produce the flat state dict of the target layout
("synthesis.b16.conv0.weight" / "mapping.fc0.weight" for the training layout, "synthesis.bs.2.conv0.weight" /
"mapping.fcs.0.weight" for the inference layout), and ``Generator.load_state_dict`` accepts either.
Resample filters travel with the checkpoints (rosinality stores the blur kernel x 4); the kernels implement the
[1,3,3,1] filter every public StyleGAN2 checkpoint uses, so other filters are rejected instead of being ignored.
"""
import re
import traceback
from functools import partial

import numpy as np
import torch

from .stylegan2 import MappingNetwork, SynthesisNetwork, channels_dict

_DEFAULT_FILTER = np.outer([1.0, 3.0, 3.0, 1.0], [1.0, 3.0, 3.0, 1.0]) / 64.0


# ------------------------------------------------------------------------------------------------ generator
class Generator(torch.nn.Module):
    """inference/stylegan2.py:439-470: ``mapping`` + ``synthesis`` with the reference's constructor and forward."""

    def __init__(self, z_dim, c_dim, w_dim, img_resolution, img_channels, mapping_kwargs={}, **synthesis_kwargs):
        super().__init__()
        self.z_dim, self.c_dim, self.w_dim = z_dim, c_dim, w_dim
        self.img_resolution, self.img_channels = img_resolution, img_channels
        nv_compat = bool(synthesis_kwargs.pop("nv_compat", False))
        self.synthesis = SynthesisNetwork(w_dim=w_dim, img_resolution=img_resolution, img_channels=img_channels,
                                          nv_compat=nv_compat, **synthesis_kwargs)
        self.num_ws = self.synthesis.num_ws
        self.mapping = MappingNetwork(z_dim=z_dim, c_dim=c_dim, w_dim=w_dim, num_ws=self.num_ws, nv_compat=nv_compat,
                                      **mapping_kwargs)

    def forward(self, z, c=None, truncation_psi=1.0, truncation_cutoff=None, noise_mode="const"):
        ws = self.mapping(z, c, truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff)
        return self.synthesis(ws, noise_mode)

    def state_dict(self, *a, **k):
        out = {f"synthesis.{k}": v for k, v in self.synthesis.state_dict().items()}
        out.update({f"mapping.{k}": v for k, v in self.mapping.state_dict().items()})
        return out

    def load_state_dict(self, sd, strict=True):
        """Accepts the training layout (b{res} / fc{i}) and the inference layout (bs.{i} / fcs.{i})."""
        syn, mp = split_generator_state(sd, self.img_resolution)
        self.synthesis.load_state_dict(syn, strict=strict)
        self.mapping.load_state_dict(mp, strict=strict)


def _check_filter(key, value):
    f = np.asarray(value.detach().double().cpu().numpy() if isinstance(value, torch.Tensor) else value, dtype=np.float64)
    if f.ndim == 1:
        f = np.outer(f, f)
    if f.shape != (4, 4) or not np.allclose(f / f.sum(), _DEFAULT_FILTER, rtol=1e-4, atol=1e-6):
        raise ValueError(f"{key}: only the [1,3,3,1] resample filter is implemented by the HIP kernels")


def split_generator_state(sd, img_resolution):
    """Flat generator state dict (either layout) -> (synthesis dict in 'bs.{i}.*' keys, mapping dict in 'fcs.{i}.*'
    keys).  ``resample_filter`` buffers are validated (only [1,3,3,1] is implemented); unknown keys are kept so that strict loading
    reports them."""
    log2 = int(np.log2(img_resolution))
    syn, mp = {}, {}
    for key, val in sd.items():
        if key.startswith("synthesis."):
            rest = key[len("synthesis."):]
            m = re.match(r"b(\d+)\.(.*)$", rest)
            if m:  # training layout: block named by its resolution
                res = int(m.group(1))
                idx = int(np.log2(res)) - 2
                if 4 * 2 ** idx != res or not (2 <= int(np.log2(res)) <= log2):
                    raise KeyError(f"{key}: block resolution {res} is not part of a {img_resolution} network")
                rest = f"bs.{idx}.{m.group(2)}"
            if rest.endswith("resample_filter"):
                _check_filter(key, val)
            syn[rest] = val
        elif key.startswith("mapping."):
            rest = key[len("mapping."):]
            m = re.match(r"fc(\d+)\.(weight|bias)$", rest)
            if m:
                rest = f"fcs.{m.group(1)}.{m.group(2)}"
            mp[rest] = val
        else:
            raise KeyError(f"{key}: expected 'synthesis.*' or 'mapping.*'")
    return syn, mp


def _infer_shape(sd):
    """(img_resolution, mapping layers, w_dim, z_dim) from a flat generator state dict in either layout."""
    res, n_map = 4, 0
    for key in sd:
        m = re.match(r"synthesis\.b(\d+)\.", key)
        if m:
            res = max(res, int(m.group(1)))
        m = re.match(r"synthesis\.bs\.(\d+)\.", key)
        if m:
            res = max(res, 4 * 2 ** int(m.group(1)))
        m = re.match(r"mapping\.fcs?\.?(\d+)\.weight$", key)
        if m:
            n_map = max(n_map, int(m.group(1)) + 1)
    first = sd.get("mapping.fc0.weight", sd.get("mapping.fcs.0.weight"))
    w_dim = int(first.shape[0]) if first is not None else 512
    z_dim = int(first.shape[1]) if first is not None else 512
    return res, n_map, w_dim, z_dim


# ------------------------------------------------------------------------------------------------ rosinality
# rosinality/stylegan2-pytorch "g_ema" layout -> NVIDIA ADA layout.  Block i (resolution 4 * 2**i, i >= 1) owns
# convs.{2i-2} (upsampling, -> conv0) and convs.{2i-1} (-> conv1), to_rgbs.{i-1} and noises.noise_{2i-1}, noise_{2i};
# the 4x4 block is input / conv1 / to_rgb1 / noise_0.  style.{n} (n >= 1; style.0 is the parameter-free PixelNorm)
# is mapping layer n-1.
_ROS_LAYER_FIELDS = {  # rosinality suffix -> (ADA suffix, squeeze leading singleton dims?)
    "conv.weight": ("weight", True),
    "activate.bias": ("bias", False),
    "conv.modulation.weight": ("affine.weight", False),
    "conv.modulation.bias": ("affine.bias", False),
}
_ROS_RGB_FIELDS = {
    "conv.weight": ("weight", True),
    "conv.modulation.weight": ("affine.weight", False),
    "conv.modulation.bias": ("affine.bias", False),
}


def rosinality_to_nvidia(checkpoint, blur_scale=4.0, for_inference=False):
    """checkpoint = {"g_ema": state dict[, "latent_avg": [w_dim]]} -> (flat ADA-layout state dict, meta).
    meta = {"img_resolution", "mapping_layers", "use_const"}.  Key for key what maua/GAN/load.py:18-116 builds
    (pinned by tests/golden/g15_load_keymap.json, generated by running the reference on a synthetic checkpoint)."""
    ros = checkpoint["g_ema"]
    out = {}

    def block(i):
        return f"synthesis.bs.{i}" if for_inference else f"synthesis.b{4 * 2 ** i}"

    def put_layer(dst, src, with_strength=True):
        for suffix, (name, squeeze) in _ROS_LAYER_FIELDS.items():
            v = ros[f"{src}.{suffix}"]
            out[f"{dst}.{name}"] = v.squeeze(0) if squeeze else v
        if with_strength and not for_inference:  # the inference layout has no learned noise strength (Q4)
            out[f"{dst}.noise_strength"] = ros[f"{src}.noise.weight"].squeeze(0)

    def put_rgb(dst, src):
        for suffix, (name, squeeze) in _ROS_RGB_FIELDS.items():
            v = ros[f"{src}.{suffix}"]
            out[f"{dst}.{name}"] = v.squeeze(0) if squeeze else v
        out[f"{dst}.bias"] = ros[f"{src}.bias"].reshape(-1)

    # ---- 4x4 block
    b0 = block(0)
    use_const = tuple(ros["input.input"].shape) != (1,)
    if use_const:
        out[f"{b0}.const"] = ros["input.input"].squeeze(0)
    else:  # learned-affine input variant
        out[f"{b0}.const.affine.weight"] = ros["input.linear.weight"].squeeze(0)
        out[f"{b0}.const.affine.bias"] = ros["input.linear.bias"].squeeze(0)
    out[f"{b0}.conv1.noise_const"] = ros["noises.noise_0"].reshape(ros["noises.noise_0"].shape[-2:])
    put_layer(f"{b0}.conv1", "conv1")
    put_rgb(f"{b0}.torgb", "to_rgb1")
    first_blur = ros["convs.0.conv.blur.kernel"] / blur_scale
    out[f"{b0}.resample_filter"] = first_blur
    out[f"{b0}.conv1.resample_filter"] = first_blur

    # ---- the rest, in the order the keys appear (the reference iterates the dict once)
    n_blocks, n_map = 1, 1
    for key in ros:
        head, _, tail = key.partition(".")
        if head == "style":
            num, field = tail.split(".")
            name = f"mapping.fcs.{int(num) - 1}.{field}" if for_inference else f"mapping.fc{int(num) - 1}.{field}"
            out[name] = ros[key]
            n_map = max(n_map, int(num))
        elif head == "noises":
            n = int(tail.split("_")[1])
            if n == 0:
                continue
            i = (n - 1) // 2 + 1
            out[f"{block(i)}.conv{(n - 1) % 2}.noise_const"] = ros[key].reshape(ros[key].shape[-2:])
        elif head == "convs":
            num, _, field = tail.partition(".")
            n = int(num)
            i, which = n // 2 + 1, n % 2
            dst = f"{block(i)}.conv{which}"
            if field in _ROS_LAYER_FIELDS:
                name, squeeze = _ROS_LAYER_FIELDS[field]
                out[f"{dst}.{name}"] = ros[key].squeeze(0) if squeeze else ros[key]
            elif field == "noise.weight":
                if not for_inference:
                    out[f"{dst}.noise_strength"] = ros[key].squeeze(0)
            elif field == "conv.blur.kernel":  # stored on the upsampling conv; both layers of the block get it
                out[f"{block(i)}.conv0.resample_filter"] = ros[key] / blur_scale
                out[f"{block(i)}.conv1.resample_filter"] = ros[key] / blur_scale
            else:
                raise KeyError(f"Key {key} not recognized!")
            n_blocks = max(n_blocks, i + 1)
        elif head == "to_rgbs":
            num, _, field = tail.partition(".")
            i = int(num) + 1
            dst = f"{block(i)}.torgb"
            if field in _ROS_RGB_FIELDS:
                name, squeeze = _ROS_RGB_FIELDS[field]
                out[f"{dst}.{name}"] = ros[key].squeeze(0) if squeeze else ros[key]
            elif field == "bias":
                out[f"{dst}.bias"] = ros[key].reshape(-1)
            elif field == "upsample.kernel":
                out[f"{block(i)}.resample_filter"] = ros[key] / blur_scale
            else:
                raise KeyError(f"Key {key} not recognized!")
        # (anything else at the top level is ignored, as in the reference)
    out["mapping.w_avg"] = checkpoint["latent_avg"] if "latent_avg" in checkpoint else torch.zeros(512)
    meta = {"img_resolution": 4 * 2 ** (n_blocks - 1), "mapping_layers": n_map, "use_const": use_const}
    return out, meta


def synthetic_rosinality_checkpoint(res=16, n_map=2, seed=7, const_input=True):
    """A random checkpoint with the key/shape structure of a rosinality StyleGAN2 generator (test input: the
    loader golden is the reference's conversion of exactly this object)."""
    g = torch.Generator().manual_seed(seed)
    ch = channels_dict(res)

    def rn(*shape):
        return torch.randn(*shape, generator=g)

    blur = torch.tensor(_DEFAULT_FILTER * 4.0, dtype=torch.float32)  # make_kernel([1,3,3,1]) * factor**2
    s = {}
    c4 = ch[4]
    if const_input:
        s["input.input"] = rn(1, c4, 4, 4)
    else:
        s["input.input"] = rn(1)
        s["input.linear.weight"] = rn(1, c4 * 16, 512)
        s["input.linear.bias"] = rn(1, c4 * 16)

    def layer(prefix, co, ci):
        s[f"{prefix}.conv.weight"] = rn(1, co, ci, 3, 3)
        s[f"{prefix}.conv.modulation.weight"] = rn(ci, 512)
        s[f"{prefix}.conv.modulation.bias"] = rn(ci)
        s[f"{prefix}.noise.weight"] = rn(1)
        s[f"{prefix}.activate.bias"] = rn(co)

    def rgb(prefix, ci):
        s[f"{prefix}.conv.weight"] = rn(1, 3, ci, 1, 1)
        s[f"{prefix}.conv.modulation.weight"] = rn(ci, 512)
        s[f"{prefix}.conv.modulation.bias"] = rn(ci)
        s[f"{prefix}.bias"] = rn(1, 3, 1, 1)

    layer("conv1", c4, c4)
    rgb("to_rgb1", c4)
    n_blocks = int(np.log2(res)) - 1
    for i in range(1, n_blocks):
        r = 4 * 2 ** i
        layer(f"convs.{2 * i - 2}", ch[r], ch[r // 2])
        s[f"convs.{2 * i - 2}.conv.blur.kernel"] = blur.clone()
        layer(f"convs.{2 * i - 1}", ch[r], ch[r])
        rgb(f"to_rgbs.{i - 1}", ch[r])
        s[f"to_rgbs.{i - 1}.upsample.kernel"] = blur.clone()
    s["noises.noise_0"] = rn(1, 1, 4, 4)
    for i in range(1, n_blocks):
        r = 4 * 2 ** i
        s[f"noises.noise_{2 * i - 1}"] = rn(1, 1, r, r)
        s[f"noises.noise_{2 * i}"] = rn(1, 1, r, r)
    for n in range(1, n_map + 1):
        s[f"style.{n}.weight"] = rn(512, 512)
        s[f"style.{n}.bias"] = rn(512)
    return {"g_ema": s, "latent_avg": rn(512)}


# ------------------------------------------------------------------------------------------------ loaders
def _build(sd, img_resolution, n_map, for_inference, z_dim=512, c_dim=0, w_dim=512, img_channels=3, dtype=torch.bfloat16):
    G = Generator(z_dim, c_dim, w_dim, img_resolution, img_channels, mapping_kwargs=dict(num_layers=n_map),
                  nv_compat=not for_inference, dtype=dtype)
    G.load_state_dict(sd)
    return G


def load_rosinality2ada(path, blur_scale=4.0, for_inference=False, dtype=torch.bfloat16):
    ck = torch.load(path, map_location="cpu")
    sd, meta = rosinality_to_nvidia(ck, blur_scale, for_inference)
    if not meta["use_const"]:
        raise NotImplementedError("rosinality checkpoints with a learned-affine input are not supported by the "
                                  "reference's generator either (maua/GAN/load.py:124 leaves use_const commented out)")
    return _build(sd, meta["img_resolution"], meta["mapping_layers"], for_inference, dtype=dtype)


def load_nvidia_pt(path, z_dim=512, c_dim=0, w_dim=512, img_resolution=1024, img_channels=3, map_layers=8,
                   for_inference=False, dtype=torch.bfloat16):
    """{"G_ema": state dict} saved without NVIDIA's persistence wrapper.  The reference takes the shape arguments on
    trust; here they are read from the state dict when they disagree with it (a 256 checkpoint loads without
    having to pass img_resolution)."""
    sd = torch.load(path, map_location="cpu")["G_ema"]
    if any(k.startswith("synthesis.input.") or ".affine.weight" in k and "L0_" in k for k in sd):
        raise NotImplementedError("StyleGAN3 checkpoints are outside the StyleGAN2 render path")
    res, n_map, w, z = _infer_shape(sd)
    return _build(sd, res, n_map or map_layers, for_inference, z_dim=z, c_dim=c_dim, w_dim=w, img_channels=img_channels,
                  dtype=dtype)


class _PickledObject:
    """Stand-in for any class of NVIDIA's code base met inside a network pickle: keeps the instance's state, runs nothing."""

    def __init__(self, *a, **k):
        self._ctor_args = (a, k)

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)
        else:
            self._state = state


def _persistent_obj(meta):
    """torch_utils.persistence._reconstruct_persistent_obj without importing the pickled module source: a persistent
    object's meta carries type='class', class_name, module_src and ``state`` = the instance's __dict__ (for an nn.Module:
    _parameters / _buffers / _modules ...).  Only the state is kept."""
    obj = _PickledObject()
    meta = dict(meta)
    obj.__dict__.update(meta.get("state") or {})
    obj._class_name = meta.get("class_name")
    return obj


def _safe_storage_from_bytes(b):
    """torch.storage._load_from_bytes with the tensors-only unpickler (the stock one calls torch.load(weights_only=False) on
    the embedded bytes, i.e. runs whatever pickle they hold)."""
    import io
    return torch.load(io.BytesIO(b), weights_only=True)


class _NetworkUnpickler(__import__("pickle").Unpickler):
    """Reads NVIDIA's network pickles (legacy.load_network_pkl's format, StyleGAN2-ADA-PyTorch / StyleGAN3: a dict with
    G / D / G_ema persistent objects) WITHOUT NVIDIA's dnnlib / torch_utils / legacy packages and without executing anything
    the file names: an explicit allow-list rebuilds tensors, parameters, numpy arrays and plain containers (a tensor's
    storage through torch's tensors-only loader); every other global - NVIDIA's classes, torch.nn modules, os.system -
    resolves to an inert stand-in that only keeps the state it is handed."""
    ALLOWED = {
        ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_tensor"), ("torch._utils", "_rebuild_parameter"),
        ("torch._utils", "_rebuild_parameter_with_state"), ("torch", "Size"), ("torch", "device"),
        ("collections", "OrderedDict"), ("_codecs", "encode"), ("copyreg", "_reconstructor"),
        ("numpy", "ndarray"), ("numpy", "dtype"), ("numpy.core.multiarray", "_reconstruct"), ("numpy.core.multiarray", "scalar"),
        ("numpy._core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "scalar"),
    }
    BUILTINS = ("dict", "list", "tuple", "set", "frozenset", "int", "float", "bool", "str", "bytes", "bytearray", "complex",
                "slice", "range", "object")

    def find_class(self, module, name):
        if module == "torch_utils.persistence" and name == "_reconstruct_persistent_obj":
            return _persistent_obj
        if (module, name) == ("torch.storage", "_load_from_bytes"):
            return _safe_storage_from_bytes
        if (module, name) in self.ALLOWED or (module == "builtins" and name in self.BUILTINS):
            return super().find_class(module, name)
        if module == "torch" and (name.endswith("Storage") or isinstance(getattr(torch, name, None), torch.dtype)):
            return getattr(torch, name)
        if name == "EasyDict":            # dnnlib.EasyDict: a dict with attribute access
            return dict
        return _PickledObject


def _module_state_dict(obj, prefix=""):
    """nn.Module.state_dict() over the raw __dict__ states of a pickled module tree."""
    sd = {}
    for name, t in (getattr(obj, "_parameters", None) or {}).items():
        if t is not None:
            sd[prefix + name] = t.detach() if hasattr(t, "detach") else t
    non_persistent = getattr(obj, "_non_persistent_buffers_set", None) or set()
    for name, t in (getattr(obj, "_buffers", None) or {}).items():
        if t is not None and name not in non_persistent:
            sd[prefix + name] = t
    for name, child in (getattr(obj, "_modules", None) or {}).items():
        if child is not None:
            sd.update(_module_state_dict(child, prefix + name + "."))
    return sd


def nvidia_pkl_state_dict(path, key="G_ema"):
    """The ``state_dict()`` of a network inside an NVIDIA ``.pkl`` (default: the EMA generator), read with the restricted
    unpickler above."""
    with open(path, "rb") as f:
        data = _NetworkUnpickler(f).load()
    if not isinstance(data, dict) or key not in data:
        raise ValueError(f"{path}: not an NVIDIA network pickle (no {key!r} entry)")
    sd = _module_state_dict(data[key])
    if not sd:
        raise ValueError(f"{path}: {key} carries no parameters")
    return {k: torch.as_tensor(v).float() for k, v in sd.items()}


def load_nvidia(path, for_inference=None, dtype=torch.bfloat16):
    """maua/GAN/load.py:130-164: NVIDIA network pickles.  The reference unpickles them with the un-vendored ``nv`` package
    (dnnlib, torch_utils.persistence, legacy) - class pickles that embed and exec their own module source.  Here the
    pickle is read by a restricted unpickler that rebuilds only tensors / arrays / containers and keeps every persistent
    object's state: G_ema's parameter tree becomes the state dict the NVIDIA ``.pt`` route takes."""
    if not str(path).endswith(".pkl"):
        raise ValueError("not a .pkl file")
    sd = nvidia_pkl_state_dict(path)
    if any(k.startswith("synthesis.input.") for k in sd):
        raise NotImplementedError("StyleGAN3 checkpoints are outside the StyleGAN2 render path")
    res, n_map, w, z = _infer_shape(sd)
    return _build(sd, res, n_map, bool(for_inference), z_dim=z, w_dim=w, dtype=dtype)


def load_flat_state_dict(path, for_inference=False, dtype=torch.bfloat16):
    """A plain ``torch.save(generator.state_dict())`` file (optionally under a "state_dict" key) in either layout —
    what this package's own Generator.state_dict() produces."""
    sd = torch.load(path, map_location="cpu")
    if isinstance(sd, dict) and "state_dict" in sd:
        sd = sd["state_dict"]
    if not isinstance(sd, dict) or not any(str(k).startswith("synthesis.") for k in sd):
        raise ValueError(f"{path}: not a flat generator state dict")
    res, n_map, w, z = _infer_shape(sd)
    return _build(sd, res, n_map, for_inference, z_dim=z, w_dim=w, dtype=dtype)


_CACHE = {}


def load_network_cached(path, for_inference=False, dtype=torch.bfloat16):
    """load_network, memoised on (path, mtime, flag, dtype): the reference's mapper and synthesizer wrappers each call
    load_network(model_file) and keep one half (wrappers/stylegan.py:21, wrappers/stylegan2.py:38)."""
    import os
    key = (os.path.abspath(path), os.path.getmtime(path), bool(for_inference), str(dtype))
    if key not in _CACHE:
        _CACHE.clear()
        _CACHE[key] = load_network(path, for_inference, dtype)
    return _CACHE[key]


def load_network(path, for_inference=False, dtype=torch.bfloat16):
    """Try the converters in the reference's order (maua/GAN/load.py:191-207) and return the first Generator."""
    errors = {}
    for name, loader in [
        ("NVIDIA StyleGAN3 loader", load_nvidia),
        ("NVIDIA non-persistence loader", load_nvidia_pt),
        ("Rosinality StyleGAN2 to ADA-PT converter", load_rosinality2ada),
        ("Rosinality StyleGAN2 to Inference converter", partial(load_rosinality2ada, for_inference=True)),
        ("flat generator state dict", load_flat_state_dict),
    ]:
        try:
            return loader(path, for_inference=for_inference, dtype=dtype)
        except Exception:
            errors[name] = traceback.format_exc()
    error_str = "\n".join(f"\n{k}:\n{e}\n" for k, e in errors.items())
    raise Exception(f"Error loading checkpoint! None of the converters succeeded:\n{error_str}")
