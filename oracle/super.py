"""Oracle restatement of the RealESRGAN x4 generator (test infrastructure only): basicsr's RRDBNet as published
(ESRGAN / Real-ESRGAN; basicsr and realesrgan are un-vendored, unpinned dependencies of the reference, setup.py:32,85 -
PARITY UNPINNED), run the way maua/super/image/models/realesrgan.py:22-49 runs it through RealESRGANer.enhance:
[0,1] image -> network -> clamp(0,1) (-> round(255 x) for the u8 frame).  Plain PyTorch-CPU fp32."""
import torch
import torch.nn.functional as F


def _conv(p, name, x):
    return F.conv2d(x, p[name + ".weight"], p[name + ".bias"], padding=1)


def rdb(p, pfx, x):
    """ResidualDenseBlock: 5 convs, growth by concatenation, lrelu(0.2), out * 0.2 + x."""
    feats = [x]
    for k in range(1, 5):
        feats.append(F.leaky_relu(_conv(p, f"{pfx}.conv{k}", torch.cat(feats, 1)), 0.2))
    return _conv(p, f"{pfx}.conv5", torch.cat(feats, 1)) * 0.2 + x


def rrdb(p, pfx, x):
    out = x
    for r in (1, 2, 3):
        out = rdb(p, f"{pfx}.rdb{r}", out)
    return out * 0.2 + x


def rrdbnet(p, x, num_block):
    """RRDBNet.forward for scale 4; x [B,3,H,W] in [0,1] -> [B,3,4H,4W] clamped to [0,1]."""
    feat = _conv(p, "conv_first", x)
    body = feat
    for i in range(num_block):
        body = rrdb(p, f"body.{i}", body)
    feat = feat + _conv(p, "conv_body", body)
    feat = F.leaky_relu(_conv(p, "conv_up1", F.interpolate(feat, scale_factor=2, mode="nearest")), 0.2)
    feat = F.leaky_relu(_conv(p, "conv_up2", F.interpolate(feat, scale_factor=2, mode="nearest")), 0.2)
    out = _conv(p, "conv_last", F.leaky_relu(_conv(p, "conv_hr", feat), 0.2))
    return out.clamp(0, 1)


def srvgg_compact(p, x, num_conv=16, upscale=4, act_type="prelu"):
    """realesrgan.archs.srvgg_arch.SRVGGNetCompact.forward (un-vendored; published architecture): conv + act, num_conv x
    (conv + act), conv -> PixelShuffle(upscale) -> + F.interpolate(x, scale_factor=upscale, mode="nearest")."""
    out = x
    for k in range(num_conv + 2):
        out = F.conv2d(out, p[f"body.{2 * k}.weight"], p[f"body.{2 * k}.bias"], padding=1)
        if k <= num_conv:
            if act_type == "prelu":
                out = F.prelu(out, p[f"body.{2 * k + 1}.weight"])
            elif act_type == "relu":
                out = F.relu(out)
            else:
                out = F.leaky_relu(out, 0.1)
    return F.pixel_shuffle(out, upscale) + F.interpolate(x, scale_factor=upscale, mode="nearest")


def realesrganer_enhance(net, img, scale=4, tile=0, tile_pad=10, pre_pad=10):
    """realesrgan.utils.RealESRGANer.enhance for a 3-channel 8-bit-range HWC image (un-vendored; published code), as
    maua/super/image/models/realesrgan.py:40,46 drives it: /255 -> BGR2RGB flip -> reflect pre_pad (right, bottom) ->
    network (whole image or tile_process) -> crop the padding -> clamp(0, 1) -> flip back -> round(255 x) u8.
    ``net``: callable [1, 3, h, w] -> [1, 3, scale h, scale w] WITHOUT a clamp."""
    import math
    import numpy as np
    x = torch.from_numpy(np.ascontiguousarray(np.asarray(img, dtype=np.float32) / 255.0))
    x = x.permute(2, 0, 1)[None][:, [2, 1, 0]]                       # cv2.COLOR_BGR2RGB
    if pre_pad:
        x = F.pad(x, (0, pre_pad, 0, pre_pad), "reflect")
    if tile > 0:
        b, c, h, w = x.shape
        out = x.new_zeros((b, c, h * scale, w * scale))
        for ty in range(math.ceil(h / tile)):
            for tx in range(math.ceil(w / tile)):
                x0, y0 = tx * tile, ty * tile
                x1, y1 = min(x0 + tile, w), min(y0 + tile, h)
                px0, px1, py0, py1 = max(x0 - tile_pad, 0), min(x1 + tile_pad, w), max(y0 - tile_pad, 0), min(y1 + tile_pad, h)
                t = net(x[:, :, py0:py1, px0:px1])
                ox, oy = (x0 - px0) * scale, (y0 - py0) * scale
                out[:, :, y0 * scale:y1 * scale, x0 * scale:x1 * scale] = \
                    t[:, :, oy:oy + (y1 - y0) * scale, ox:ox + (x1 - x0) * scale]
    else:
        out = net(x)
    if pre_pad:
        out = out[:, :, : out.shape[2] - pre_pad * scale, : out.shape[3] - pre_pad * scale]
    out = out[0].clamp(0, 1)[[2, 1, 0]].permute(1, 2, 0)
    return (out * 255.0).round().numpy().astype(np.uint8)


def rrdbnet_raw(p, x, num_block):
    """RRDBNet.forward without the final clamp (what RealESRGANer stitches)."""
    feat = _conv(p, "conv_first", x)
    body = feat
    for i in range(num_block):
        body = rrdb(p, f"body.{i}", body)
    feat = feat + _conv(p, "conv_body", body)
    feat = F.leaky_relu(_conv(p, "conv_up1", F.interpolate(feat, scale_factor=2, mode="nearest")), 0.2)
    feat = F.leaky_relu(_conv(p, "conv_up2", F.interpolate(feat, scale_factor=2, mode="nearest")), 0.2)
    return _conv(p, "conv_last", F.leaky_relu(_conv(p, "conv_hr", feat), 0.2))
