"""RealESRGAN x4 up-scaling on the HIP device (first slice; BASELINE configs[4] "StyleGAN2 render ->
RealESRGAN 4x").  Drop-in for maua/super/image/models/realesrgan.py:22-49 (``load_model`` / ``upscale``) for the RRDBNet
models ("x4plus", "pbaylies-*": 23 blocks; "x4plus-anime": 6 blocks): same names and call shapes, the network itself
(basicsr's RRDBNet, un-vendored) runs behind the C ABI (maua_rrdb_*, csrc/super.hip).  "xsx4-animevideo"
(SRVGGNetCompact) is not implemented.  There is no network access for the published checkpoints: ``load_model`` loads
``modelzoo/RealESRGAN_<name>.pth`` when it exists and otherwise builds a seeded random-init network of the right shape."""
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

    def __init__(self, num_in_ch=3, num_out_ch=3, scale=4, num_feat=64, num_block=23, num_grow_ch=32,
                 dtype=torch.bfloat16, generator=None):
        super().__init__()
        if (num_in_ch, num_out_ch, scale) != (3, 3, 4):
            raise NotImplementedError("only the 3 -> 3 channel, x4 RRDBNet of RealESRGAN is implemented")
        self.num_feat, self.num_block, self.num_grow_ch, self.dtype = num_feat, num_block, num_grow_ch, dtype
        self._params = init_rrdb_params(3, 3, num_feat, num_block, num_grow_ch, generator)
        self._net = None

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
            for k, v in self._params.items():
                a = np.ascontiguousarray(v.numpy(), dtype=np.float32)
                L.check(L.lib().maua_rrdb_load(net, k.encode(), a.ctypes.data_as(C.c_void_p), C.c_size_t(a.size)))
            self._net = net
        else:
            L.ctx()
        return self._net

    def forward(self, x, out=None, rgb8_out=None):
        """x f32 [B, 3, H, W] in [0, 1] -> f32 [B, 3, 4H, 4W] clamped to [0, 1] (RealESRGANer.enhance's post-process);
        ``rgb8_out``: optional uint8 [B, 4H, 4W, 3] receiving round(255 * y) in the same call."""
        x = L.dev_tensor(x, torch.float32)
        b, c, h, w = x.shape
        if c != 3:
            raise ValueError("RRDBNet expects 3-channel images")
        if out is None and rgb8_out is None:
            out = torch.empty((b, 3, 4 * h, 4 * w), dtype=torch.float32, device=x.device)
        L.check(L.lib().maua_rrdb_forward(self._handle(), L.ptr(x), b, h, w, L.ptr(out), L.ptr(rgb8_out)))
        return out if out is not None else rgb8_out


class RealESRGANer:
    """The slice of realesrgan.RealESRGANer the reference uses (scale 4, tile 0): ``enhance(img)`` on an HWC image in
    [0, 255] (numpy) -> (HWC uint8 image 4x the size, None)."""

    def __init__(self, scale=4, model_path=None, model=None, tile=0, half=True):
        if scale != 4 or tile != 0:
            raise NotImplementedError("scale 4 without tiling (what the reference constructs)")
        self.scale, self.model = scale, model
        if model_path is not None and os.path.exists(model_path):
            self.model.load_state_dict(torch.load(model_path, map_location="cpu"), strict=True)

    @torch.inference_mode()
    def enhance(self, img, outscale=None):
        x = torch.from_numpy(np.ascontiguousarray(img, dtype=np.float32) / 255.0).permute(2, 0, 1)[None]
        u8 = torch.empty((1, 4 * x.shape[2], 4 * x.shape[3], 3), dtype=torch.uint8, device="cuda")
        self.model(x, rgb8_out=u8)
        return u8[0].cpu().numpy(), None


def load_model(model_name="pbaylies-hr-paintings", device=None, dtype=torch.bfloat16):
    """realesrgan.py:22-40."""
    if model_name not in BLOCKS:
        raise NotImplementedError(f"{model_name}: only the RRDBNet models {sorted(BLOCKS)} are implemented")
    model = RRDBNet(num_in_ch=3, num_out_ch=3, num_feat=64, num_block=BLOCKS[model_name], num_grow_ch=32, scale=4, dtype=dtype)
    return RealESRGANer(scale=4, model_path=f"modelzoo/RealESRGAN_{model_name}.pth", model=model, tile=0, half=True)


@torch.inference_mode()
def upscale(images, model):
    """realesrgan.py:43-49: images = iterable of [1, 3, H, W] tensors in [0, 1]; yields [1, 3, 4H, 4W] in [0, 1]."""
    for img in images:
        x = torch.as_tensor(img).detach().squeeze().permute(1, 2, 0).mul(255).cpu().numpy()
        large = model.enhance(x)[0]
        yield torch.from_numpy(large).permute(2, 0, 1).unsqueeze(0).float().div(255)
