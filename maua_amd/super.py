"""RealESRGAN x4 up-scaling on the HIP device (BASELINE configs[4] "StyleGAN2 render -> RealESRGAN 4x").
Drop-in for maua/super/image/models/realesrgan.py:22-49 (``load_model`` / ``upscale``): same names and call shapes for all
five models - the RRDBNet ones ("x4plus", "pbaylies-*": 23 blocks; "x4plus-anime": 6 blocks) and "xsx4-animevideo"
(SRVGGNetCompact) - and the slice of realesrgan.RealESRGANer the reference drives (``enhance`` with pre_pad, optional tiling
with tile_pad, the BGR<->RGB flips).  The networks (basicsr / realesrgan, un-vendored: published architectures, parity
unpinned) run behind the C ABI (maua_rrdb_* / maua_srvgg_*, csrc/super.hip).  There is no network access for the published
checkpoints: ``load_model`` loads ``modelzoo/RealESRGAN_<name>.pth`` and raises FileNotFoundError when it is missing, unless
``allow_random_init=True`` asks for a seeded random-init network of the right shape (benchmarks, tests)."""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib as L

BLOCKS = {"x4plus": 23, "x4plus-anime": 6, "pbaylies-wikiart": 23, "pbaylies-hr-paintings": 23}


def init_rrdb_params(num_in_ch=3, num_out_ch=3, num_feat=64, num_block=23, num_grow_ch=32, generator=None, scale=0.1):
    """A state dict with basicsr's RRDBNet key names; kaiming-normal weights (x ``scale`` inside the dense blocks, as
    basicsr's default_init_weights does), zero biases."""
    g = generator or torch.Generator().manual_seed(0)

    def w(co, ci, s=1.0):
        return torch.randn(co, ci, 3, 3, generator=g) * (s * (2.0 / (ci * 9)) ** 0.5)
    p = {"conv_first.weight": w(num_feat, num_in_ch), "conv_first.bias": torch.zeros(num_feat)}
    for i in range(num_block):
        for r in (1, 2, 3):
            for k in range(5):
                co = num_feat if k == 4 else num_grow_ch
                p[f"body.{i}.rdb{r}.conv{k + 1}.weight"] = w(co, num_feat + k * num_grow_ch, scale)
                p[f"body.{i}.rdb{r}.conv{k + 1}.bias"] = torch.zeros(co)
    for name, co in (("conv_body", num_feat), ("conv_up1", num_feat), ("conv_up2", num_feat), ("conv_hr", num_feat),
                     ("conv_last", num_out_ch)):
        p[name + ".weight"] = w(co, num_feat)
        p[name + ".bias"] = torch.zeros(co)
    return p


class RRDBNet(torch.nn.Module):
    """basicsr.archs.rrdbnet_arch.RRDBNet(num_in_ch=3, num_out_ch=3, scale=4, ...) - forward on the HIP device."""

    scale = 4

    def __init__(self, num_in_ch=3, num_out_ch=3, scale=4, num_feat=64, num_block=23, num_grow_ch=32,
                 dtype=torch.bfloat16, generator=None):
        super().__init__()
        if (num_in_ch, num_out_ch, scale) != (3, 3, 4):
            raise NotImplementedError("only the 3 -> 3 channel, x4 RRDBNet of RealESRGAN is implemented")
        self.num_feat, self.num_block, self.num_grow_ch, self.dtype = num_feat, num_block, num_grow_ch, dtype
        self._params = init_rrdb_params(3, 3, num_feat, num_block, num_grow_ch, generator)
        self._net = None
        self._flip = False   # RealESRGANer's BGR <-> RGB flips, folded into the first / last convolution (set_channel_flip)

    def state_dict(self, *a, **k):
        return dict(self._params)

    def load_state_dict(self, sd, strict=True):
        sd = sd.get("params_ema", sd.get("params", sd)) if isinstance(sd, dict) else sd   # RealESRGANer's checkpoint keys
        missing = [k for k in self._params if k not in sd]
        unexpected = [k for k in sd if k not in self._params]
        if strict and (missing or unexpected):
            raise KeyError(f"missing {missing[:4]}..., unexpected {unexpected[:4]}...")
        for k in self._params:
            if k in sd:
                if tuple(sd[k].shape) != tuple(self._params[k].shape):
                    raise ValueError(f"{k}: shape {tuple(sd[k].shape)} != {tuple(self._params[k].shape)}")
                self._params[k] = sd[k].detach().float().cpu().contiguous()
        self._destroy()

    def _destroy(self):
        if self._net is not None:
            L.lib().maua_rrdb_destroy(self._net)
            self._net = None

    def __del__(self):
        try:
            self._destroy()
        except Exception:
            pass

    def _handle(self):
        L.require_device()
        if self._net is None:
            net = C.c_void_p()
            L.check(L.lib().maua_rrdb_create(L.ctx(), self.num_feat, self.num_block, self.num_grow_ch,
                                             L.dtype_id(self.dtype), C.byref(net)))
            for k, v in _flipped(self._params, "conv_first", "conv_last", self._flip).items():
                a = np.ascontiguousarray(v.numpy(), dtype=np.float32)
                L.check(L.lib().maua_rrdb_load(net, k.encode(), a.ctypes.data_as(C.c_void_p), C.c_size_t(a.size)))
            self._net = net
        else:
            L.ctx()
        return self._net

    def set_channel_flip(self, flag):
        """y = flip(net(flip(x))) over the colour channels without a copy: the first convolution's input channels and the
        last convolution's output channels are loaded in reversed order."""
        if bool(flag) != self._flip:
            self._flip = bool(flag)
            self._destroy()

    def forward(self, x, out=None, rgb8_out=None, clamp=True):
        """x f32 [B, 3, H, W] in [0, 1] -> f32 [B, 3, 4H, 4W], clamped to [0, 1] unless ``clamp=False`` (RealESRGANer clamps
        after its own stitching); ``rgb8_out``: optional uint8 [B, 4H, 4W, 3] receiving round(255 * clamp(y)) in the same call."""
        x = L.dev_tensor(x, torch.float32)
        b, c, h, w = x.shape
        if c != 3:
            raise ValueError("RRDBNet expects 3-channel images")
        if out is None and rgb8_out is None:
            out = torch.empty((b, 3, 4 * h, 4 * w), dtype=torch.float32, device=x.device)
        L.check(L.lib().maua_rrdb_forward_ex(self._handle(), L.ptr(x), b, h, w, int(bool(clamp)), L.ptr(out), L.ptr(rgb8_out)))
        return out if out is not None else rgb8_out

    def enhance_u8(self, frames_u8, pre_pad=0, out=None):
        """RealESRGANer.enhance's arithmetic for device-resident uint8 frames [B, h, w, 3] in one library call (maua_rrdb_enhance_u8:
        x / 255 and the reflect pre_pad in the first convolution's input staging, clamp / round(255 y) / crop in the last one's
        store) -> uint8 [B, 4h, 4w, 3]."""
        f = frames_u8.cuda().contiguous()
        b, h, w, _ = f.shape
        if out is None:
            out = torch.empty((b, 4 * h, 4 * w, 3), dtype=torch.uint8, device=f.device)
        L.check(L.lib().maua_rrdb_enhance_u8(self._handle(), L.ptr(f), b, h, w, int(pre_pad), L.ptr(out)))
        return out


def _flipped(params, first, last, flip):
    """The parameter dict with the colour channels of the first convolution's input and the last convolution's output
    reversed (when ``flip``): net'(x) = net(x[:, [2, 1, 0]])[:, [2, 1, 0]]."""
    if not flip:
        return params
    p = dict(params)
    p[first + ".weight"] = params[first + ".weight"][:, [2, 1, 0]].contiguous()
    w, b = params[last + ".weight"], params[last + ".bias"]
    co = w.shape[0]
    perm = torch.arange(co).reshape(3, co // 3).flip(0).reshape(-1)    # channel c of a pixel-shuffled output: c * s^2 + ...
    p[last + ".weight"], p[last + ".bias"] = w[perm].contiguous(), b[perm].contiguous()
    return p


def init_srvgg_params(num_feat=64, num_conv=16, upscale=4, generator=None):
    """A state dict with SRVGGNetCompact's key names (body.<2k> convolutions, body.<2k+1> PReLU slopes)."""
    g = generator or torch.Generator().manual_seed(0)
    p = {}
    chans = [3] + [num_feat] * (num_conv + 1) + [3 * upscale * upscale]
    for k in range(num_conv + 2):
        ci, co = chans[k], chans[k + 1]
        p[f"body.{2 * k}.weight"] = torch.randn(co, ci, 3, 3, generator=g) * (1.0 / (ci * 9)) ** 0.5
        p[f"body.{2 * k}.bias"] = 0.05 * torch.randn(co, generator=g)
        if k <= num_conv:
            p[f"body.{2 * k + 1}.weight"] = 0.25 + 0.1 * torch.randn(num_feat, generator=g)
    return p


class SRVGGNetCompact(torch.nn.Module):
    """realesrgan.archs.srvgg_arch.SRVGGNetCompact(num_in_ch=3, num_out_ch=3, num_feat, num_conv, upscale, act_type) -
    forward on the HIP device (maua_srvgg_*): convolutions + PReLU, PixelShuffle, + the nearest-upsampled input."""

    def __init__(self, num_in_ch=3, num_out_ch=3, num_feat=64, num_conv=16, upscale=4, act_type="prelu",
                 dtype=torch.bfloat16, generator=None):
        super().__init__()
        if (num_in_ch, num_out_ch) != (3, 3) or act_type not in ("prelu", "relu", "leakyrelu"):
            raise NotImplementedError("3 -> 3 channels, act_type prelu / relu / leakyrelu")
        self.num_feat, self.num_conv, self.upscale, self.act_type, self.dtype = num_feat, num_conv, upscale, act_type, dtype
        self.scale = upscale
        self._params = init_srvgg_params(num_feat, num_conv, upscale, generator)
        if act_type != "prelu":
            self._params = {k: v for k, v in self._params.items() if int(k.split(".")[1]) % 2 == 0}
        self._net, self._flip = None, False

    def state_dict(self, *a, **k):
        return dict(self._params)

    load_state_dict = RRDBNet.load_state_dict
    set_channel_flip = RRDBNet.set_channel_flip

    def _destroy(self):
        if self._net is not None:
            L.lib().maua_srvgg_destroy(self._net)
            self._net = None

    def __del__(self):
        try:
            self._destroy()
        except Exception:
            pass

    def _handle(self):
        L.require_device()
        if self._net is None:
            net = C.c_void_p()
            act = {"prelu": 0, "relu": 1, "leakyrelu": 2}[self.act_type]
            L.check(L.lib().maua_srvgg_create(L.ctx(), self.num_feat, self.num_conv, self.upscale, act, L.dtype_id(self.dtype),
                                              C.byref(net)))
            last = f"body.{2 * (self.num_conv + 1)}"
            for k, v in _flipped(self._params, "body.0", last, self._flip).items():
                a = np.ascontiguousarray(v.numpy(), dtype=np.float32)
                L.check(L.lib().maua_srvgg_load(net, k.encode(), a.ctypes.data_as(C.c_void_p), C.c_size_t(a.size)))
            self._net = net
        else:
            L.ctx()
        return self._net

    def forward(self, x, out=None, rgb8_out=None, clamp=True):
        x = L.dev_tensor(x, torch.float32)
        b, c, h, w = x.shape
        if c != 3:
            raise ValueError("SRVGGNetCompact expects 3-channel images")
        s = self.upscale
        if out is None and rgb8_out is None:
            out = torch.empty((b, 3, s * h, s * w), dtype=torch.float32, device=x.device)
        L.check(L.lib().maua_srvgg_forward(self._handle(), L.ptr(x), b, h, w, int(bool(clamp)), L.ptr(out), L.ptr(rgb8_out)))
        return out if out is not None else rgb8_out



class RealESRGANer:
    """realesrgan.utils.RealESRGANer as the reference constructs and calls it (realesrgan.py:40 ``RealESRGANer(scale=4,
    model_path=..., model=..., tile=0, half=True)``; :46 ``model.enhance(input)[0]``), restated from its published code
    (un-vendored): ``enhance(img)`` on an HWC image in [0, 255] -> (HWC uint8 image ``scale`` x the size, "RGB").

      pre_process : [0, 1] planar image, reflect-padded by ``pre_pad`` (default 10) on the right and bottom
      process     : the network on the whole image, or (``tile`` > 0) on tile x tile crops padded by ``tile_pad`` input
                    pixels on every side that has neighbours, the centre of each result pasted into the output
      post_process: crop the pre_pad * scale border, clamp to [0, 1], round(255 x)
    enhance() treats its input as BGR (cv2) and flips to RGB before the network and back after it; the reference hands it
    RGB, so its network sees channel-swapped images - kept (the flips are folded into the first / last convolution)."""

    def __init__(self, scale=4, model_path=None, dni_weight=None, model=None, tile=0, tile_pad=10, pre_pad=10, half=False,
                 device=None, gpu_id=None, allow_random_init=False):
        if dni_weight is not None:
            raise NotImplementedError("deep network interpolation of two checkpoints (not used by the reference)")
        self.scale, self.tile_size, self.tile_pad, self.pre_pad, self.mod_scale, self.half = scale, tile, tile_pad, pre_pad, None, half
        self.model = model
        if getattr(model, "scale", scale) != scale:
            raise ValueError(f"the model up-scales x{model.scale}, RealESRGANer was asked for x{scale}")
        if model_path is not None and os.path.exists(model_path):
            loadnet = torch.load(model_path, map_location="cpu")
            key = "params_ema" if "params_ema" in loadnet else ("params" if "params" in loadnet else None)
            self.model.load_state_dict(loadnet[key] if key else loadnet, strict=True)
        elif model_path is not None and not allow_random_init:
            raise FileNotFoundError(f"{model_path} not found (the reference downloads it; this box has no network): place "
                                    "the checkpoint there, or pass allow_random_init=True for a seeded random-init network")
        self.model.set_channel_flip(True)

    def _net(self, x):
        return self.model(x, clamp=False)

    @torch.inference_mode()
    def enhance(self, img, outscale=None, alpha_upsampler="realesrgan"):
        from . import ops
        img = np.asarray(img, dtype=np.float32)
        if img.ndim != 3 or img.shape[2] != 3:
            raise NotImplementedError("3-channel images (gray / RGBA inputs are not on the reference's call path)")
        max_range = 65535 if np.max(img) > 256 else 255
        x = L.dev_tensor(torch.from_numpy(np.ascontiguousarray(img / max_range)).permute(2, 0, 1)[None], torch.float32)
        if self.pre_pad:
            x = ops.pad2d(x, (0, self.pre_pad, 0, self.pre_pad), "reflect")
        if self.tile_size > 0:
            out = self._tile_process(x)
        else:
            out = self._net(x)
        if self.pre_pad:
            out = out[:, :, : out.shape[2] - self.pre_pad * self.scale, : out.shape[3] - self.pre_pad * self.scale]
        out = out[0].clamp(0, 1).permute(1, 2, 0)
        if max_range == 65535:
            res = (out * 65535.0).round().cpu().numpy().astype(np.uint16)
        else:
            res = (out * 255.0).round().byte().cpu().numpy()
        if outscale is not None and outscale != float(self.scale):
            raise NotImplementedError("outscale != scale (a cv2 Lanczos resize in realesrgan; not used by the reference)")
        return res, "RGB"

    @torch.inference_mode()
    def enhance_frames(self, frames_u8):
        """``enhance`` for a BATCH of frames that never leave the device (BASELINE configs[4]: render -> x4 per frame): uint8
        [B, H, W, 3] RGB as the renderer packs them -> uint8 [B, scale H, scale W, 3].  Per frame exactly enhance()'s arithmetic
        (x / 255, reflect pre_pad on the right / bottom, the network, crop, clamp to [0, 1], round(255 x)); the reference reaches
        the same frames through a video file (super/video/frame_by_frame.py:22-33: decoded frame / 255 -> upscale -> writer)."""
        from . import ops
        f = torch.as_tensor(frames_u8)
        if f.dtype != torch.uint8 or f.ndim != 4 or f.shape[-1] != 3:
            raise ValueError("enhance_frames expects uint8 [B, H, W, 3] frames")
        f = f.cuda()
        b, h, w, _ = f.shape
        if self.tile_size <= 0 and hasattr(self.model, "enhance_u8"):
            # (round 6) the whole of enhance() inside the library: no torch convert / pad / crop passes around the network
            return self.model.enhance_u8(f, self.pre_pad)
        x = f.permute(0, 3, 1, 2).float().div_(255.0).contiguous()
        if self.pre_pad:
            x = ops.pad2d(x, (0, self.pre_pad, 0, self.pre_pad), "reflect")
        s = self.scale
        if self.tile_size > 0:
            out = self._tile_process(x)[:, :, : h * s, : w * s]
            return out.clamp_(0, 1).mul_(255.0).round_().byte().permute(0, 2, 3, 1).contiguous()
        # the network packs round(255 clamp(y)) itself (same rounding as enhance's .round()); the pre_pad border is cut from the u8 frames
        u8 = torch.empty((b, x.shape[2] * s, x.shape[3] * s, 3), dtype=torch.uint8, device=x.device)
        self.model(x, rgb8_out=u8)
        return u8[:, : h * s, : w * s].contiguous() if self.pre_pad else u8

    def _tile_process(self, x):
        """RealESRGANer.tile_process: ceil(H / tile) x ceil(W / tile) tiles."""
        import math
        b, c, h, w = x.shape
        s, T, P = self.scale, self.tile_size, self.tile_pad
        out = torch.zeros((b, c, h * s, w * s), dtype=torch.float32, device=x.device)
        for ty in range(math.ceil(h / T)):
            for tx in range(math.ceil(w / T)):
                x0, y0 = tx * T, ty * T
                x1, y1 = min(x0 + T, w), min(y0 + T, h)
                px0, px1, py0, py1 = max(x0 - P, 0), min(x1 + P, w), max(y0 - P, 0), min(y1 + P, h)
                tile = self._net(x[:, :, py0:py1, px0:px1].contiguous())
                ox, oy = (x0 - px0) * s, (y0 - py0) * s
                out[:, :, y0 * s:y1 * s, x0 * s:x1 * s] = tile[:, :, oy:oy + (y1 - y0) * s, ox:ox + (x1 - x0) * s]
        return out


def load_model(model_name="pbaylies-hr-paintings", device=None, dtype=torch.bfloat16, allow_random_init=False, tile=0):
    """realesrgan.py:22-40."""
    if model_name == "xsx4-animevideo":
        model = SRVGGNetCompact(num_in_ch=3, num_out_ch=3, num_feat=64, num_conv=16, upscale=4, act_type="prelu", dtype=dtype)
    elif model_name in BLOCKS:
        model = RRDBNet(num_in_ch=3, num_out_ch=3, num_feat=64, num_block=BLOCKS[model_name], num_grow_ch=32, scale=4, dtype=dtype)
    else:
        raise KeyError(f"{model_name}: the reference's models are {sorted(BLOCKS) + ['xsx4-animevideo']}")
    return RealESRGANer(scale=4, model_path=f"modelzoo/RealESRGAN_{model_name}.pth", model=model, tile=tile, half=True,
                        allow_random_init=allow_random_init)


@torch.inference_mode()
def upscale(images, model):
    """realesrgan.py:43-49: images = iterable of [1, 3, H, W] tensors in [0, 1]; yields [1, 3, 4H, 4W] in [0, 1]."""
    for img in images:
        x = torch.as_tensor(img).detach().squeeze().permute(1, 2, 0).mul(255).cpu().numpy()
        large = model.enhance(x)[0]
        yield torch.from_numpy(large).permute(2, 0, 1).unsqueeze(0).float().div(255)
