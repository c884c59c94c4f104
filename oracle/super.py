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
