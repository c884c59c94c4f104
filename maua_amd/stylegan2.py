"""StyleGAN2 networks and maua's generator wrappers, executed by libmaua_hip.so.

Module API drop-in (same class names, constructor arguments, attributes and forward kwargs):
  * SynthesisNetwork / MappingNetwork   <- maua/GAN/wrappers/inference/stylegan2.py:116-192, :385-436
  * StyleGAN2Mapper / StyleGAN2Synthesizer / StyleGAN2 <- maua/GAN/wrappers/stylegan.py:11-77,
    maua/GAN/wrappers/stylegan2.py:22-102, :196-213; MauaGenerator.render <- maua/GAN/wrappers/__init__.py:52-99

Parameters keep the reference's state_dict names, so a reference state dict loads unchanged.  Random
initialisation draws from torch's CPU generator in the reference constructors' order, so
``torch.manual_seed(s)`` gives the same weights as the reference's ``SynthesisNetwork`` under that seed.
The forward pass is one C-ABI call per batch (maua_synth_forward); nothing here falls back to PyTorch math.
"""
import ctypes as C
from math import sqrt
from typing import Dict, Optional

import warnings

import numpy as np
import torch

from . import _lib as L
from . import ops


# ------------------------------------------------------------------------------------------- structure / init
def block_resolutions(img_resolution):
    return [2 ** i for i in range(2, int(np.log2(img_resolution)) + 1)]


def channels_dict(img_resolution, channel_base=32768, channel_max=512):
    return {r: min(channel_base // r, channel_max) for r in block_resolutions(img_resolution)}


def _randn(shape, generator):
    if callable(generator):   # (init_synthesis_params_parallel: the caller fills the tensors itself)
        return generator(shape)
    return torch.randn(shape, generator=generator) if generator is not None else torch.randn(shape)


def init_synthesis_params(img_resolution, w_dim=512, img_channels=3, channel_base=32768, channel_max=512,
                          generator=None, architecture="skip") -> Dict[str, torch.Tensor]:
    """Same tensors, same draw order as the reference constructors (stylegan2.py:296-337, :221-227, :263-265; "orig" /
    "resnet": a toRGB layer in the last block only, "resnet": the blocks' 1 x 1 skip convolution :331-337 drawn last)."""
    ch = channels_dict(img_resolution, channel_base, channel_max)
    f = ops.setup_filter([1, 3, 3, 1])
    p = {}

    def layer(prefix, cin, cout, res):
        p[prefix + ".resample_filter"] = f.clone()
        p[prefix + ".affine.weight"] = _randn([cin, w_dim], generator)
        p[prefix + ".affine.bias"] = torch.ones([cin])
        p[prefix + ".weight"] = _randn([cout, cin, 3, 3], generator)
        p[prefix + ".noise_const"] = _randn([res, res], generator)
        p[prefix + ".bias"] = torch.zeros([cout])

    for i, r in enumerate(block_resolutions(img_resolution)):
        cin = ch[r // 2] if r > 4 else 0
        cout = ch[r]
        p[f"bs.{i}.resample_filter"] = f.clone()
        if cin == 0:
            p[f"bs.{i}.const"] = _randn([cout, r, r], generator)
        else:
            layer(f"bs.{i}.conv0", cin, cout, r)
        layer(f"bs.{i}.conv1", cout, cout, r)
        if architecture == "skip" or r == img_resolution:
            p[f"bs.{i}.torgb.affine.weight"] = _randn([cout, w_dim], generator)
            p[f"bs.{i}.torgb.affine.bias"] = torch.ones([cout])
            p[f"bs.{i}.torgb.weight"] = _randn([img_channels, cout, 1, 1], generator)
            p[f"bs.{i}.torgb.bias"] = torch.zeros([img_channels])
        if architecture == "resnet" and cin != 0:
            p[f"bs.{i}.skip.resample_filter"] = f.clone()
            p[f"bs.{i}.skip.weight"] = _randn([cout, cin, 1, 1], generator)
    return p


def init_synthesis_params_parallel(img_resolution, w_dim=512, img_channels=3, channel_base=32768, channel_max=512, seed=0,
                                   workers=8) -> Dict[str, torch.Tensor]:
    """The same parameter set with every random tensor drawn from its OWN generator (seeded from ``seed`` and the tensor's
    position) on a small thread pool - torch's CPU normal sampler is single-threaded and releases the GIL, and the 23.6 M
    draws of a 1024^2 network are the longest serial item of a clip's set-up (0.09 s -> ~0.02 s).  Deterministic in ``seed``;
    NOT the values ``init_synthesis_params(generator=manual_seed(seed))`` gives (one generator, reference draw order)."""
    from concurrent.futures import ThreadPoolExecutor
    todo = []

    def defer(shape):
        t = torch.empty(shape)
        todo.append(t)
        return t
    p = init_synthesis_params(img_resolution, w_dim, img_channels, channel_base, channel_max, generator=defer)

    def fill(i):
        todo[i].normal_(generator=torch.Generator().manual_seed((seed * 1000003 + i) & 0x7fffffff))
    order = sorted(range(len(todo)), key=lambda i: -todo[i].numel())    # largest first: the pool stays balanced
    with ThreadPoolExecutor(max_workers=workers) as ex:
        list(ex.map(fill, order))
    return p


def init_synthesis_params_device(img_resolution, w_dim=512, img_channels=3, channel_base=32768, channel_max=512, seed=0,
                                 device=None) -> Dict[str, torch.Tensor]:
    """The same parameter set with every random tensor drawn ON THE DEVICE from the build-owned counter RNG (rng.PhiloxStreams:
    the k-th random tensor of the construction order is Philox stream k of ``seed``) - no host draws, no upload: what the benchmark's
    synthetic generator uses.  Identical on every rank / device and reproducible on the host through oracle/rng.py;
    NOT the values of torch's generator."""
    from .rng import PhiloxStreams
    return init_synthesis_params(img_resolution, w_dim, img_channels, channel_base, channel_max,
                                 generator=PhiloxStreams(seed, device=device))


def parse_seeds(seeds):
    """wrappers/stylegan.py:59-65: "a-b,c" -> ints (ranges end-exclusive)."""
    out = []
    for s in str(seeds).split(","):
        if "-" in s:
            a, b = s.split("-")
            out += list(range(int(a), int(b)))
        else:
            out.append(int(s))
    return out


@L.host_threads(1)
def get_z_latents(seeds, z_dim=512):
    """wrappers/stylegan.py:58-69: float64 [P, z_dim] from numpy's MT19937 (host)."""
    return torch.cat([torch.from_numpy(np.random.RandomState(s).randn(1, z_dim)) for s in parse_seeds(seeds)])


class SynthesisNetwork(torch.nn.Module):
    """inference/stylegan2.py:385-436.  ``dtype``: torch.bfloat16 (default, MFMA bf16 operands / f32 accumulate: every fast
    kernel), torch.float32 (exact-f32 MFMA, parity mode) or torch.float16 (IEEE half operands / f32 accumulate with ops.py:161-165's
    pre-normalisation of weights and styles - the reference's half type - on the generic MFMA kernels)."""

    def __init__(self, w_dim, img_resolution, img_channels=3, channel_base=32768, channel_max=512, num_fp16_res=0,
                 dtype=torch.bfloat16, nv_compat=False, generator=None, _params=None, **block_kwargs):
        super().__init__()
        if img_channels != 3:
            raise NotImplementedError("img_channels must be 3")
        self.architecture = block_kwargs.pop("architecture", "skip")
        if self.architecture not in ("orig", "skip", "resnet"):
            raise ValueError(f"architecture must be 'orig', 'skip' or 'resnet', got {self.architecture!r}")
        self.w_dim, self.img_resolution, self.img_channels = w_dim, img_resolution, img_channels
        self.img_resolution_log2 = int(np.log2(img_resolution))
        self.channel_base, self.channel_max = channel_base, channel_max
        self.block_resolutions = block_resolutions(img_resolution)
        self.num_ws = 2 * len(self.block_resolutions)
        self.num_layers = 2 * len(self.block_resolutions) - 1
        self.dtype, self.nv_compat = dtype, bool(nv_compat)
        # (_params: clone() hands over an existing parameter dict - no random init, the RNG is not touched)
        self._params = dict(_params) if _params is not None else \
            init_synthesis_params(img_resolution, w_dim, img_channels, channel_base, channel_max, generator,
                                  architecture=self.architecture)
        self._dev_params = None  # "orig" / "resnet": device copies of the parameters for the layer-at-a-time forward
        self._net = None  # device handle, created on first use
        self._net_device = None
        self._keep_features = False
        self._resize = None  # feature-space resize spec (set_resize), re-applied when the device object is rebuilt
        # noise buffers of the layers the resize re-sized: kept apart from _params, which always hold the network's
        # own (native-size) noise_const, so that set_resize(None) / a second resize / a rebuilt device object all
        # find the right tensors to upload
        self._resized_noise = {}

    # -- parameters ------------------------------------------------------------------------------------
    def state_dict(self, *a, **k):
        """Parameters as the device object holds them: after a feature-space resize the later layers' noise_const
        entries are the re-sized buffers, as in the reference (wrappers/stylegan2.py:141-146 replaces the modules'
        buffers); the network's own buffers come back with set_resize(None)."""
        return {**self._params, **self._resized_noise}

    def load_state_dict(self, sd, strict=True):
        missing = [k for k in self._params if k not in sd]
        unexpected = [k for k in sd if k not in self._params and not k.endswith("noise_strength")]
        if strict and (missing or unexpected):
            raise KeyError(f"missing {missing[:4]}..., unexpected {unexpected[:4]}...")
        for k, v in sd.items():
            if k in self._params:
                if tuple(v.shape) != tuple(self._params[k].shape):
                    raise ValueError(f"{k}: shape {tuple(v.shape)} != {tuple(self._params[k].shape)}")
                self._params[k] = v.detach().float().cpu().contiguous()
            elif k.endswith("noise_strength"):
                self._params[k] = v.detach().float().cpu().reshape(1)
        self._dev_params = None
        self._destroy()

    def layer_shapes(self):
        """[(prefix, c_in, c_out, resolution, up)] for the synthesis layers in execution order."""
        ch = channels_dict(self.img_resolution, self.channel_base, self.channel_max)
        out = []
        for i, r in enumerate(self.block_resolutions):
            if r > 4:
                out.append((f"bs.{i}.conv0", ch[r // 2], ch[r], r, 2))
            out.append((f"bs.{i}.conv1", ch[r], ch[r], r, 1))
        return out

    # -- device object ---------------------------------------------------------------------------------
    def _destroy(self):
        if self._net is not None:
            L.lib().maua_synth_destroy(self._net)
            self._net = None

    def __del__(self):
        try:
            self._destroy()
        except Exception:
            pass

    def _handle(self, upload_noise=True):
        """The device object (built on first use).  ``upload_noise=False``: set_resize builds the object and then does
        the one noise upload itself with the caller's generator (a draw here would come from the global RNG and be
        cached, so the caller's seed would never be used)."""
        L.require_device()
        if self.architecture != "skip":
            raise NotImplementedError(f"architecture {self.architecture!r} runs a layer at a time (forward only): the one-call "
                                      "device network, feature capture and the resize / warp hooks are the 'skip' networks'")
        dev = torch.cuda.current_device()
        if self._net is None or self._net_device != dev:
            self._destroy()
            lib = L.lib()
            net = C.c_void_p()
            flags = 3 if self.nv_compat else 0
            L.check(lib.maua_synth_create(L.ctx(dev), self.img_resolution, self.w_dim, self.channel_base,
                                          self.channel_max, L.dtype_id(self.dtype), flags, C.byref(net)))
            for k, v in self._params.items():
                if v.is_cuda:   # drawn on the device (rng.PhiloxStreams): no host round trip
                    v = v.to(device=dev, dtype=torch.float32).contiguous()
                    L.check(lib.maua_synth_load_device(net, k.encode(), L.ptr(v), C.c_size_t(v.numel())))
                    continue
                a = np.ascontiguousarray(v.numpy(), dtype=np.float32)
                L.check(lib.maua_synth_load(net, k.encode(), a.ctypes.data_as(C.c_void_p), C.c_size_t(a.size)))
            L.check(lib.maua_synth_set_option(net, b"keep_features", int(self._keep_features)))
            self._net, self._net_device = net, dev
            if self._resize is not None:
                self._apply_resize()
                if upload_noise:
                    self._upload_noise(None)  # (rebuilt object: the resized buffers already exist in _resized_noise)
        else:
            L.ctx(dev)  # re-bind torch's current stream
        return self._net

    # -- arbitrary output size ---------------------------------------------------------
    def set_resize(self, layer, mode="stretch", target=None, padding=(0, 0, 0, 0), pad_how="constant", pad_value=0.0,
                   fill_noise=None, noise_generator=None):
        """Resize the feature map at ``layer`` (index into the reference's layer_names: 0 = input of bs.0.conv1,
        L >= 1 = output of synthesis layer L-1) to ``target`` = (h, w); every later layer runs at the scaled size.
        ``mode``: "stretch" (bicubic) or "pad" with ``padding`` = (left, right, top, bottom), ``pad_how`` in
        {"constant", "reflect", "replicate", "circular"}.  ``fill_noise``: optional [C, h, w] tensor added to the
        resized features.  Layers after the resize get fresh N(0,1) noise_const of their new size from
        ``noise_generator`` (wrappers/stylegan2.py:139-150).  ``layer=None`` removes the resize."""
        if layer is None:
            self._resize = None
            self._resized_noise = {}
            if self._net is not None:
                L.check(L.lib().maua_synth_set_resize(self._net, -1, 0, 0, 0, 0, 0, 0, 0, 3, C.c_float(0.0), None))
                self._upload_noise(None)  # the device re-allocated (zeroed) the buffers whose size changed back
            return
        fn = None if fill_noise is None else np.ascontiguousarray(fill_noise.detach().float().cpu().numpy())
        self._resize = dict(layer=int(layer), mode={"stretch": 0, "pad": 1}[mode], th=int(target[0]), tw=int(target[1]),
                            padding=tuple(int(p) for p in padding),
                            how={"circular": 0, "reflect": 1, "replicate": 2, "constant": 3}[pad_how],
                            value=float(pad_value), fill=fn)
        if self._net is not None or torch.cuda.is_available():
            if self._net is None:
                self._handle(upload_noise=False)  # builds the object and applies the resize
            else:
                self._apply_resize()
            self._upload_noise(noise_generator)  # the single upload: fresh buffers come from the caller's generator

    def _upload_noise(self, noise_generator):
        """After every maua_synth_set_resize: (re-)upload the noise buffer of every layer - the network's own
        noise_const where the layer runs at its native size, a resized layer's buffer otherwise (drawn fresh from
        ``noise_generator`` when there is none of the right size yet, wrappers/stylegan2.py:139-150)."""
        for l, (pfx, _, _, res, _) in enumerate(self.layer_shapes()):
            h, w = self.layer_size(l)
            key = pfx + ".noise_const"
            if (h, w) == (res, res):
                nz = self._params[key]
                self._resized_noise.pop(key, None)
            else:
                nz = self._resized_noise.get(key)
                if nz is None or tuple(nz.shape) != (h, w):
                    nz = torch.randn((h, w), generator=noise_generator)
                    self._resized_noise[key] = nz
            if nz.is_cuda:
                nz = nz.to(dtype=torch.float32).contiguous()
                L.check(L.lib().maua_synth_load_device(self._net, key.encode(), L.ptr(nz), C.c_size_t(nz.numel())))
                continue
            a = np.ascontiguousarray(nz.numpy(), dtype=np.float32)
            L.check(L.lib().maua_synth_load(self._net, key.encode(), a.ctypes.data_as(C.c_void_p), C.c_size_t(a.size)))

    def clone(self):
        """An independent network with the same parameters (own device object, own resize state): what every wrapper
        built from one cached checkpoint gets."""
        return SynthesisNetwork(self.w_dim, self.img_resolution, self.img_channels, self.channel_base, self.channel_max,
                                dtype=self.dtype, nv_compat=self.nv_compat, _params=self._params,
                                architecture=self.architecture)

    def _apply_resize(self):
        r = self._resize
        fill = None if r["fill"] is None else r["fill"].ctypes.data_as(C.c_void_p)
        pl, pr, pt, pb = r["padding"]
        L.check(L.lib().maua_synth_set_resize(self._net, r["layer"], r["mode"], r["th"], r["tw"], pl, pr, pt, pb,
                                              r["how"], C.c_float(r["value"]), fill))

    def layer_size(self, layer):
        """(h, w) of synthesis layer ``layer``'s output (= its noise size); layer -1: the final image."""
        if self._net is None and self._resize is None:
            r = self.img_resolution if layer < 0 else self.layer_shapes()[layer][3]
            return r, r
        h, w = C.c_int(), C.c_int()
        L.check(L.lib().maua_synth_layer_size(self._handle(), layer, C.byref(h), C.byref(w)))
        return h.value, w.value

    @property
    def output_hw(self):
        return self.layer_size(-1)

    def keep_features(self, flag=True):
        self._keep_features = bool(flag)
        if self._net is not None:
            L.check(L.lib().maua_synth_set_option(self._net, b"keep_features", int(flag)))

    # -- forward ---------------------------------------------------------------------------------------
    def _noise_args(self, noise, B):
        if noise is None:
            return None, None, []
        n = self.num_layers
        ptrs = (C.c_void_p * n)()
        strides = (C.c_long * n)()
        keep = []
        shapes = self.layer_shapes()
        for l in range(n):
            t = noise[l] if l < len(noise) else None
            if t is None:
                ptrs[l], strides[l] = None, 0
                continue
            rh, rw = self.layer_size(l)
            t = L.dev_tensor(t, torch.float32)
            if tuple(t.shape[-2:]) != (rh, rw):  # wrappers/stylegan2.py:92-98: resize mismatching noise
                t = ops.interpolate_bicubic(t.reshape(-1, 1, *t.shape[-2:]).contiguous(), (rh, rw))
            nb = t.numel() // (rh * rw)
            if nb not in (1, B):
                raise ValueError(f"noise{l}: batch {nb} does not match latents batch {B}")
            keep.append(t)
            ptrs[l], strides[l] = t.data_ptr(), (0 if nb == 1 else rh * rw)
        return ptrs, strides, keep

    def forward(self, ws, noise_mode="const", noise=None, out=None, rgb8_out=None):
        """ws [B, num_ws, w_dim] -> img f32 [B, 3, R, R].  ``noise``: optional list of per-layer tensors
        [B|1, 1, h, w] replacing the layers' noise_const for this call (what the reference's
        StyleGAN2Synthesizer installs per batch).  ``rgb8_out``: optional uint8 [B, R, R, 3] buffer that
        receives the packed frame (render/ffmpeg.py:72 + ops/io.py:47-70) in the same call."""
        if noise_mode != "const":
            raise NotImplementedError("noise_mode must be 'const' (the render path never uses 'random')")
        if self.architecture != "skip":
            return self._forward_layerwise(ws, noise, out, rgb8_out)
        net = self._handle()
        ws = L.dev_tensor(ws, torch.float32)
        B = ws.shape[0]
        if tuple(ws.shape[1:]) != (self.num_ws, self.w_dim):
            raise ValueError(f"ws must be [B, {self.num_ws}, {self.w_dim}], got {tuple(ws.shape)}")
        if out is None and rgb8_out is None:
            oh, ow = self.output_hw
            out = torch.empty((B, 3, oh, ow), dtype=torch.float32, device=ws.device)
        # un-normalised Loop maps (noise.loop_batch(..., raw=True) -> RawNoise) bring their per-(layer, sample) factors 1 / (rms + eps)
        # along: the convolution epilogues multiply them into the noise strength (the maps are written once instead of twice).  Any
        # other sequence - also a RawNoise somebody iterated or sliced - holds normalised maps
        from .noise import RawNoise
        sc = None
        if isinstance(noise, RawNoise):
            noise, sc = noise.maps, noise.scales
            if sc is None or sc.shape[0] != self.num_layers or sc.shape[1] < B:
                raise ValueError(f"noise scales must be [{self.num_layers}, >= {B}], got {None if sc is None else tuple(sc.shape)}")
        ptrs, strides, keep = self._noise_args(noise, B)
        # (the factors apply to the NEXT render call only: the library forgets them when that call returns)
        L.check(L.lib().maua_synth_set_noise_scale(net, L.ptr(sc), C.c_long(0 if sc is None else sc.stride(0))))
        L.check(L.lib().maua_synth_render_rgb8(net, L.ptr(ws), ptrs, strides, B, L.ptr(out), L.ptr(rgb8_out)))
        del keep
        return out if out is not None else rgb8_out

    def _forward_layerwise(self, ws, noise, out, rgb8_out):
        """SynthesisBlock.forward, inference/stylegan2.py:340-382, for the "orig" and "resnet" architectures: one library call
        per layer (maua_modconv2d with its fused bias_act, maua_upfirdn2d, maua_add) instead of the one-call network - these
        architectures are not the reference's default and none of its shipped networks uses them, so they get the operator
        path, not fused kernels.  "orig": no skip images, the last block's toRGB is the image; "resnet": y = skip(x) * sqrt(.5)
        (1 x 1, up-sampled), x = conv1(conv0(x), gain sqrt(.5)), x = y + x."""
        L.require_device()
        ws = L.dev_tensor(ws, torch.float32)
        B = ws.shape[0]
        if tuple(ws.shape[1:]) != (self.num_ws, self.w_dim):
            raise ValueError(f"ws must be [B, {self.num_ws}, {self.w_dim}], got {tuple(ws.shape)}")
        from .noise import RawNoise
        if isinstance(noise, RawNoise):
            noise = noise.normalised()
        if self._dev_params is None or next(iter(self._dev_params.values())).device != ws.device:
            self._dev_params = {k: v.to(ws.device) for k, v in self._params.items()}
        p = self._dev_params
        resnet = self.architecture == "resnet"

        def affine(prefix, w):    # FullyConnectedLayer (linear, bias_init 1): F.linear(w, W / sqrt(fan_in), b)
            W = p[prefix + ".affine.weight"]
            y = ops.matmul_nt(w.contiguous(), W * (1.0 / sqrt(W.shape[1])))
            return ops.bias_act(y[:, :, None, None], p[prefix + ".affine.bias"])[:, :, 0, 0]

        def layer(prefix, x, w, up, li, gain):
            nz = noise[li] if noise is not None and li < len(noise) and noise[li] is not None else p[prefix + ".noise_const"]
            strength = float(p[prefix + ".noise_strength"][0]) if self.nv_compat and (prefix + ".noise_strength") in p else 1.0
            return ops.modulated_conv2d(x, p[prefix + ".weight"], affine(prefix, w), noise=nz, up=up, padding=1,
                                        resample_filter=p[prefix + ".resample_filter"], flip_weight=self.nv_compat,
                                        bias=p[prefix + ".bias"], act="lrelu", gain=sqrt(2.0) * gain, clamp=256.0 * gain,
                                        noise_strength=strength)
        x = None
        w_idx = li = 0
        for i, r in enumerate(self.block_resolutions):
            g1 = 1.0
            if i == 0:
                x = p["bs.0.const"].to(self.dtype).unsqueeze(0).expand(B, -1, -1, -1).contiguous()
            else:
                y = None
                if resnet:   # Conv2dLayer(1 x 1, bias=False, up=2), gain sqrt(.5): the 1 x 1 kernel commutes with the up-sampling
                    wk = p[f"bs.{i}.skip.weight"]
                    y = ops.conv2d_resample(x, wk * (sqrt(0.5) / sqrt(wk.shape[1])), padding=0)
                    y = ops.upsample2d(y, p[f"bs.{i}.skip.resample_filter"], up=2)
                    g1 = sqrt(0.5)
                x = layer(f"bs.{i}.conv0", x, ws[:, w_idx], 2, li, 1.0)
                w_idx, li = w_idx + 1, li + 1
            x = layer(f"bs.{i}.conv1", x, ws[:, w_idx], 1, li, g1)
            w_idx, li = w_idx + 1, li + 1
            if i > 0 and resnet:
                x = ops.add(y, x)
        last = f"bs.{len(self.block_resolutions) - 1}.torgb"
        cin = p[last + ".weight"].shape[1]
        img = ops.modulated_conv2d(x, p[last + ".weight"], affine(last, ws[:, w_idx]) * (1.0 / sqrt(cin)), demodulate=False,
                                   padding=0, bias=p[last + ".bias"], clamp=256.0).float()
        if rgb8_out is not None:
            # (render/ffmpeg.py:72 + ops/io.py:47-70: (img + 1) / 2 -> clamp -> * 255 -> round half even -> u8 HWC)
            L.check(L.lib().maua_pack_rgb8(L.ctx(img.device), L.ptr(img), L.ptr(rgb8_out), B, img.shape[2], img.shape[3]))
        if out is not None:
            out.copy_(img)
            return out
        return img if rgb8_out is None else rgb8_out

    def get_feature(self, layer, B):
        shp = self.layer_shapes()[layer]
        h, w = self.layer_size(layer)
        if self._resize is not None and self._resize["layer"] == layer + 1:  # the hook's output replaces the layer's
            h, w = self._resize["th"], self._resize["tw"]
        out = torch.empty((B, shp[2], h, w), dtype=torch.float32, device="cuda")
        L.check(L.lib().maua_synth_get_feature(self._handle(), layer, B, L.ptr(out)))
        return out


class MappingNetwork(torch.nn.Module):
    """inference/stylegan2.py:116-192 (c_dim == 0).  A [P,512] x 8 small f32 GEMM chain executed once per clip through
    the library's own `maua_matmul_nt` (initialising a BLAS library for it would cost the clip 0.5 s); in-tree semantics
    incl. the x @ w quirk."""

    @L.host_threads(1)
    def __init__(self, z_dim, c_dim, w_dim, num_ws, num_layers=8, embed_features=None, layer_features=None, activation="lrelu",
                 lr_multiplier=0.01, w_avg_beta=0.998, nv_compat=False, generator=None):
        super().__init__()
        if c_dim != 0:
            raise NotImplementedError("class conditioning is not on the render path")
        if layer_features not in (None, w_dim) or activation != "lrelu":
            raise NotImplementedError("the reference networks' mapper: w_dim-wide lrelu layers")
        self.w_avg_beta = w_avg_beta   # (a training-time decay: unused at inference, kept for the constructor's signature)
        self.z_dim, self.c_dim, self.w_dim, self.num_ws, self.num_layers = z_dim, c_dim, w_dim, num_ws, num_layers
        self.lr_multiplier, self.nv_compat = lr_multiplier, nv_compat
        feats = [z_dim] + [w_dim] * num_layers
        self._params = {}
        for i in range(num_layers):
            self._params[f"fcs.{i}.weight"] = _randn([feats[i + 1], feats[i]], generator) / lr_multiplier
            # (a device generator - rng.PhiloxStreams - leaves the whole parameter set on its device: nothing is uploaded at forward time)
            self._params[f"fcs.{i}.bias"] = torch.zeros([feats[i + 1]], device=self._params[f"fcs.{i}.weight"].device)
        self._params["w_avg"] = torch.zeros([w_dim], device=self._params["fcs.0.weight"].device)

    def state_dict(self, *a, **k):
        return dict(self._params)

    def load_state_dict(self, sd, strict=True):
        # strict like torch's: a conditional checkpoint (embed.*, fc0 fed with z_dim + embedding) must not load into
        # this unconditional network silently
        unexpected = [k for k in sd if k not in self._params]
        if strict and unexpected:
            raise KeyError(f"unexpected mapping-network keys {unexpected[:4]} (conditional networks are not supported)")
        for k in self._params:
            if k in sd:
                v = sd[k].detach().float().cpu()
                if tuple(v.shape) != tuple(self._params[k].shape):
                    raise ValueError(f"mapping.{k}: shape {tuple(v.shape)} != {tuple(self._params[k].shape)}")
                self._params[k] = v
            elif strict:
                raise KeyError(k)

    @L.host_threads(1)
    def forward(self, z, c=None, truncation_psi=1.0, truncation_cutoff=None):
        L.require_device()
        x = L.dev_tensor(z, torch.float32)
        x = ops.normalize_2nd_moment(x)
        for i in range(self.num_layers):
            # (gain and layout of the 512 x 512 matrices are prepared on the host: the once-per-clip mapper then runs only
            #  library kernels - the first elementwise PyTorch kernel of a process costs ~0.3 s of code-object loading)
            wh = self._params[f"fcs.{i}.weight"]
            w = wh * (self.lr_multiplier / sqrt(wh.shape[1]))
            b = (self._params[f"fcs.{i}.bias"] * self.lr_multiplier).to(x.device)
            # F.linear(x, w) upstream; the in-tree layer multiplies by the un-transposed matrix
            y = ops.matmul_nt(x, (w if self.nv_compat else w.T.contiguous()).to(x.device))
            x = ops.bias_act(y[:, :, None, None], b, act="lrelu")[:, :, 0, 0]
        x = ops.repeat_rows(x, self.num_ws)
        if truncation_psi != 1:  # stylegan2.py:185-190: lerp towards w_avg, all ws or only the first `cutoff`
            w_avg = self._params["w_avg"].to(x.device)
            if truncation_cutoff is None:
                x = w_avg.lerp(x, truncation_psi)
            else:
                x[:, :truncation_cutoff] = w_avg.lerp(x[:, :truncation_cutoff], truncation_psi)
        return x


# =================================================================================================== wrappers
class MauaMapper(torch.nn.Module):
    """maua/GAN/wrappers/__init__.py:20-22"""

    def forward(self):
        raise NotImplementedError()


class MauaSynthesizer(torch.nn.Module):
    """maua/GAN/wrappers/__init__.py:25-38"""
    _hook_handles = []

    def forward(self):
        raise NotImplementedError()

    def change_output_resolution(self):
        raise NotImplementedError()

    def refresh_model_hooks(self):
        self._hook_handles = []


def _load_generator(model_file, inference, dtype=torch.bfloat16):
    """maua/GAN/load.py:191-207 via maua_amd.load (rosinality / NVIDIA state-dict / flat state-dict checkpoints)."""
    from .load import load_network_cached
    return load_network_cached(model_file, inference, dtype)


class StyleGAN2Mapper(MauaMapper):
    """maua/GAN/wrappers/stylegan.py:11-32 (StyleGANMapper) with the StyleGAN2 mapping network."""

    def __init__(self, model_file=None, inference=False, generator=None):
        super().__init__()
        if model_file is None or model_file == "None":
            self.G_map = MappingNetwork(z_dim=512, c_dim=0, w_dim=512, num_ws=18, generator=generator)
        else:
            self.G_map = _load_generator(model_file, inference).mapping
        self.z_dim, self.c_dim = self.G_map.z_dim, self.G_map.c_dim
        self.modulation_targets = {"latent_z": (self.z_dim,), "truncation": (1,)}

    def forward(self, latent_z, class_conditioning=None, truncation=1.0):
        return self.G_map.forward(latent_z, class_conditioning, truncation_psi=truncation)


def resize_strategy(layer_size, target_hw, strategy):
    """The strategy strings of get_hook (wrappers/stylegan2.py:216-290) -> SynthesisNetwork.set_resize kwargs."""
    th, tw = target_hw
    if strategy == "stretch":
        return dict(mode="stretch")
    if strategy.startswith("pad"):
        parts = strategy.split("-")
        if len(parts) != 3:  # (the reference's CLI default "pad-zero" unpacks into 3 names and fails)
            raise ValueError(f"Resize strategy must be 'pad-<how>-<where>': {strategy}")
        _, how, where = parts
        pad_h, pad_w = th - layer_size, tw - layer_size
        # (negative values crop, as F.pad does at :294; defined at layer 0 only - maua_synth_set_resize says why)
        half = lambda p: (p // 2, round(1e-16 + p / 2))
        if where == "out":
            padding = (*half(pad_w), *half(pad_h))
        elif where == "left":
            padding = (pad_w, 0, *half(pad_h))
        elif where == "right":
            padding = (0, pad_w, *half(pad_h))
        elif where == "top":
            padding = (*half(pad_w), pad_h, 0)
        elif where == "bottom":
            padding = (*half(pad_w), 0, pad_h)
        else:
            raise ValueError(f"Resize strategy not found: {strategy}")
        if how in ("reflect", "replicate", "circular"):
            return dict(mode="pad", padding=padding, pad_how=how, pad_value=0.0)
        return dict(mode="pad", padding=padding, pad_how="constant", pad_value=float(how))
    raise Exception(f"Resize strategy not found: {strategy}")


def _num_ws_from_sd(sd):
    n = 0
    while f"bs.{n}.conv1.weight" in sd:
        n += 1
    return 2 * n


class StyleGAN2Synthesizer(MauaSynthesizer):
    """maua/GAN/wrappers/stylegan2.py:22-151: same constructor, attributes (w_dim, num_ws, layer_names,
    modulation_targets, output_size, G_synth) and forward kwargs, incl. arbitrary output sizes through the
    feature-space resize of change_output_resolution ("stretch" / "pad-<how>-<where>" at ``layer``).  The geometric
    transform hooks (kornia translate / rotate / zoom, :153-194) are not implemented."""

    def __init__(self, model_file=None, inference=False, output_size=None, strategy="stretch", layer=0,
                 img_resolution=1024, dtype=torch.bfloat16, generator=None):
        super().__init__()
        if model_file is None or model_file == "None":
            # The reference always builds the 1024 net and reaches other sizes through feature-space resize hooks.
            # For a random-init net a square power-of-two output_size selects the NATIVE net of that size instead
            # (SURVEY config C0); every other size goes through change_output_resolution like the reference.
            if output_size is not None and output_size[0] == output_size[1] and output_size[0] >= 8 \
                    and (output_size[0] & (output_size[0] - 1)) == 0:
                img_resolution = int(output_size[0])
            self.G_synth = SynthesisNetwork(w_dim=512, img_resolution=img_resolution, img_channels=3, dtype=dtype,
                                            generator=generator)
        else:
            self.G_synth = _load_generator(model_file, inference, dtype).synthesis.clone()
        R = self.G_synth.img_resolution
        if output_size is None:
            output_size = (R, R)
        self.w_dim, self.num_ws = self.G_synth.w_dim, self.G_synth.num_ws
        self.layer_names = [f"bs.{c // 2}.conv{1 if bs == 4 else c % 2}"
                            for c, bs in enumerate(sorted(self.G_synth.block_resolutions * 2))]
        self.modulation_targets = {"latent_w": (self.w_dim,), "latent_w_plus": (self.num_ws, self.w_dim),
                                   "translation": (2,), "rotation": (1,)}
        self.output_size = (R, R)
        self._generator = generator
        self.change_output_resolution(tuple(output_size), strategy, layer)

    def change_output_resolution(self, output_size, strategy, layer, add_noise=True):
        """wrappers/stylegan2.py:104-151 + get_hook :216-340.  ``output_size`` is (width, height) like the reference
        (:217 flips it); it is rounded to a multiple of img_resolution // layer_size.  The per-channel fill noise of
        the resized features (drawn by the reference from the statistics of one random forward, :233-248) and the new
        noise buffers of the later layers (:141-146) come from this object's generator."""
        G = self.G_synth
        R = G.img_resolution
        G.set_resize(None)
        if tuple(output_size) != (R, R):
            _, block, _conv = self.layer_names[layer].split(".")
            layer_size = 4 * 2 ** int(block)
            lay_mult = R // layer_size
            unrounded = np.array(output_size) / lay_mult
            target = np.round(unrounded).astype(int)                 # (W, H)
            if sum(abs(unrounded - target)) > 1e-10:
                warnings.warn(f"Layer {layer} resizes to multiples of {lay_mult}. --output-size rounded to "
                              f"{lay_mult * target}")
            tw, th = int(target[0]), int(target[1])
            kw = resize_strategy(layer_size, (th, tw), strategy)
            gen = self._generator
            G.set_resize(layer, target=(th, tw), noise_generator=gen, **kw)
            if add_noise:
                # statistics of the resized features of one random forward (the reference's warm-up call, :149)
                if layer == 0:
                    const = G.state_dict()["bs.0.const"][None]
                    x = ops.interpolate_bicubic(const, (th, tw)) if kw["mode"] == "stretch" else \
                        ops.pad2d(const, kw["padding"], kw["pad_how"], kw["pad_value"])
                else:
                    keep = G._keep_features
                    G.keep_features(True)
                    G.forward(torch.randn((1, self.num_ws, self.w_dim), generator=gen))
                    x = G.get_feature(layer - 1, 1)
                    G.keep_features(keep)
                x = x.float().cpu()
                mean, std = x.mean((0, 2, 3)), x.transpose(0, 1).reshape(x.shape[1], -1).std(1)
                fill = torch.randn(x.shape[1:], generator=gen) * std[:, None, None] + mean[:, None, None]
                G.set_resize(layer, target=(th, tw), fill_noise=fill, noise_generator=gen, **kw)
        self.output_size = tuple(output_size)

    # -- geometric transform hooks (wrappers/stylegan2.py:153-194) ----------------------------------------
    def _set_warp(self, slot, layer, matrix):
        """matrix [B, 2, 3]: kornia's forward (source -> destination, pixel coordinates) affine of translate / scale /
        rotate; the kernel samples the source at inverse(matrix) @ destination.  Like the reference's hooks, a warp
        stays installed until a later call replaces it."""
        M = torch.as_tensor(matrix, dtype=torch.float32).reshape(-1, 2, 3).cpu()
        A, t = M[:, :, :2], M[:, :, 2:]
        Ainv = torch.linalg.inv(A)
        minv = torch.cat([Ainv, -Ainv @ t], dim=2).reshape(-1, 6).contiguous()
        if not hasattr(self, "_warps"):
            self._warps = {}
        # layer_names[0] and [1] both name bs.0.conv1 (the reference lists every block twice): a hook on either lands
        # on synthesis layer 1
        self._warps[slot] = dict(layer=max(1, int(layer)), host=minv, dev=None)
        self._install_warp(slot, len(minv))

    def _install_warp(self, slot, B):
        """Upload slot's matrices for a batch of B frames: one row per frame; a single row is broadcast (the kernel
        reads B rows); any other count is an error instead of an out-of-bounds read."""
        w = self._warps[slot]
        n = len(w["host"])
        if n != B and n != 1:
            raise ValueError(f"transform hook holds {n} matrices, the batch has {B} frames")
        rows = w["host"] if n == B else w["host"].expand(B, 6).contiguous()
        if w["dev"] is None or len(w["dev"]) != B:
            w["dev"] = L.dev_tensor(rows, torch.float32)  # (kept alive here: the library stores the pointer)
            L.check(L.lib().maua_synth_set_warp(self.G_synth._handle(), slot, w["layer"], L.ptr(w["dev"])))

    def _layer_hw(self, layer):
        h, w = self.G_synth.layer_size(layer - 1)
        r = self.G_synth._resize
        if r is not None and r["layer"] == layer:  # the resize hook runs first: warps see the resized grid
            h, w = r["th"], r["tw"]
        return h, w

    def apply_translation(self, layer, translation):
        """:153-166 kT.translate(output, translation * [[h, w]], padding_mode="reflection")"""
        h, w = self._layer_hw(layer)
        t = torch.as_tensor(translation, dtype=torch.float32).reshape(-1, 2).cpu() * torch.tensor([[h, w]], dtype=torch.float32)
        M = torch.eye(2, 3).repeat(len(t), 1, 1)
        M[:, 0, 2], M[:, 1, 2] = t[:, 0], t[:, 1]
        self._set_warp(0, layer, M)

    @staticmethod
    def _rotation_scale_matrix(angle_deg, scale, center, h, w, n):
        """kornia get_rotation_matrix2d(center, angle, scale): [[a, b, (1-a) cx - b cy], [-b, a, b cx + (1-a) cy]] with
        a = s cos, b = s sin; the default centre is the image centre ((w-1)/2, (h-1)/2)."""
        if center is None:
            center = torch.tensor([[(w - 1) / 2, (h - 1) / 2]], dtype=torch.float32).repeat(n, 1)
        center = torch.as_tensor(center, dtype=torch.float32).reshape(-1, 2).cpu().expand(n, 2)
        rad = torch.deg2rad(angle_deg)
        a, b = scale * torch.cos(rad), scale * torch.sin(rad)
        cx, cy = center[:, 0], center[:, 1]
        return torch.stack([torch.stack([a, b, (1 - a) * cx - b * cy], 1),
                            torch.stack([-b, a, b * cx + (1 - a) * cy], 1)], 1)

    def apply_rotation(self, layer, angle, center):
        """:168-180 kT.rotate(output, angle.squeeze(), center, padding_mode="reflection") (degrees, anti-clockwise)"""
        h, w = self._layer_hw(layer)
        ang = torch.as_tensor(angle, dtype=torch.float32).reshape(-1).cpu()
        self._set_warp(2, layer, self._rotation_scale_matrix(ang, torch.ones_like(ang), center, h, w, len(ang)))

    def apply_zoom(self, layer, zoom, center):
        """:182-194 kT.scale(output, zoom.squeeze(), center, padding_mode="reflection")"""
        h, w = self._layer_hw(layer)
        z = torch.as_tensor(zoom, dtype=torch.float32).reshape(-1).cpu()
        self._set_warp(1, layer, self._rotation_scale_matrix(torch.zeros_like(z), z, center, h, w, len(z)))

    def forward(self, latents, translation=None, translation_layer=7, zoom=None, zoom_layer=7, zoom_center=None,
                rotation=None, rotation_layer=7, rotation_center=None, rgb8_out=None, **noise):
        if translation is not None:
            self.apply_translation(translation_layer, translation)
        if zoom is not None:
            self.apply_zoom(zoom_layer, zoom, zoom_center)
        if rotation is not None:
            self.apply_rotation(rotation_layer, rotation, rotation_center)
        for slot in getattr(self, "_warps", {}):  # hooks persist across calls: re-check them against this batch
            self._install_warp(slot, len(latents))
        # noise kwargs are consumed in dict order as layer 0..16 (wrappers/stylegan2.py:86-100)
        nz = list(noise.values()) if noise else None
        return self.G_synth.forward(latents, noise_mode="const", noise=nz, rgb8_out=rgb8_out)

    def make_noise_pyramid(self, noise, layer_limit=8):
        """wrappers/stylegan2.py:196-213: bicubic resize of a [T,1,h,w] base to each layer's size, / per-frame std
        (a clip-level pre-pass; torch device ops)."""
        noises = {}
        shapes = self.G_synth.layer_shapes()
        noise = L.dev_tensor(noise, torch.float32)
        for l, layer in enumerate(self.layer_names[1:]):
            if l > layer_limit:
                continue
            h, w = self.G_synth.layer_size(l)
            n = ops.interpolate_bicubic(noise, (h, w))  # maua_resize2d: F.interpolate(bicubic, align_corners=False)
            noises[f"noise{l}"] = n / n.std((1, 2, 3), keepdim=True)
        return noises


_fp16_warned = set()


def _warn_fp16_flag(synthesizer, fp16):
    """``fp16=False`` cannot turn a network whose weights were prepared in a 16-bit type into a float32 one: say so once."""
    dt = getattr(getattr(synthesizer, "G_synth", synthesizer), "dtype", None)
    if not fp16 and dt in (torch.bfloat16, torch.float16) and dt not in _fp16_warned:
        import warnings
        _fp16_warned.add(dt)
        warnings.warn(f"fp16=False requested, but the synthesizer was built with dtype={dt}: it renders in that type "
                      "(float32 accumulation); build it with dtype=torch.float32 for float32 compute")


class MauaGenerator(torch.nn.Module):
    """maua/GAN/wrappers/__init__.py:41-99"""
    MapperCls = None
    SynthesizerCls = None

    def __init__(self, mapper_kwargs={}, synthesizer_kwargs={}):
        super().__init__()
        self.mapper = self.__class__.MapperCls(**mapper_kwargs)
        self.synthesizer = self.__class__.SynthesizerCls(**synthesizer_kwargs)

    def render(self, inputs, batch_size=32, postprocess_fn=lambda x: x, device=None, fp16=True, batched=True,
               verbose=False):
        """Generator over frame batches in [0,1] (wrappers/__init__.py:52-99).  ``fp16``: the reference (:62-82, render/ffmpeg.py:
        41-60) casts the INPUTS to float16 and sets ``use_fp16`` on the synthesis blocks - a flag its in-tree inference network
        stores (inference/stylegan2.py:297) and never reads, while F.linear / conv2d refuse the float16-input x float32-weight mix
        that results.  What the flag asks for - 16-bit compute - is here a property of the synthesizer: built with the default
        ``dtype=torch.bfloat16`` or with ``torch.float16`` (IEEE half operands, ops.py:161-165's pre-normalisation) it renders in
        that type with float32 accumulation, inputs kept in float32; ``fp16=False`` on such a synthesizer warns once (build it with
        ``dtype=torch.float32`` for the exact-f32 path) instead of silently computing in 16 bits."""
        _warn_fp16_flag(self.synthesizer, fp16)
        keys = list(inputs.keys())
        T = len(inputs[keys[0]])
        for i in range(0, T, batch_size):
            batch = {k: inputs[k][i:i + batch_size] for k in keys}
            frames = self.synthesizer.forward(**batch).add(1).div(2).clamp(0, 1)
            frames = postprocess_fn(frames)
            if batched:
                yield frames
            else:
                for f in frames:
                    yield f[None]


class StyleGAN2(MauaGenerator):
    """maua/GAN/wrappers/stylegan.py:39-77 + stylegan2.py StyleGAN2."""
    MapperCls = StyleGAN2Mapper
    SynthesizerCls = StyleGAN2Synthesizer

    def __init__(self, model_file=None, inference=False, output_size=None, strategy="stretch", layer=0, **kw):
        # construction order (mapper, then synthesizer) and the shared generator mirror MauaGenerator.__init__
        super().__init__(mapper_kwargs=dict(model_file=model_file, inference=inference, generator=kw.get("generator")),
                         synthesizer_kwargs=dict(model_file=model_file, inference=inference, output_size=output_size,
                                                 strategy=strategy, layer=layer, **kw))
        self.z_dim, self.c_dim = self.mapper.G_map.z_dim, self.mapper.G_map.c_dim
        self.w_dim, self.num_ws = self.mapper.G_map.w_dim, self.synthesizer.num_ws
        self.mapper.G_map.num_ws = self.num_ws
        self.res = self.synthesizer.G_synth.img_resolution
        self.model_file = model_file

    def get_z_latents(self, seeds):
        return get_z_latents(seeds, self.mapper.z_dim)

    def get_w_latents(self, seeds, truncation=1):
        return self.mapper(self.get_z_latents(seeds).float(), truncation=truncation)

    def forward(self, z, *args, c=None, **kwargs):
        return self.synthesizer(self.mapper(z, c))


def get_generator_class(architecture):
    """maua/GAN/wrappers/__init__.py:102-112"""
    if architecture == "stylegan2":
        return StyleGAN2
    raise Exception(f"Architecture not found: {architecture}")
