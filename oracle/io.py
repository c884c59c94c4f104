"""Oracle restatement of the output/seed boundary (test infrastructure only)."""
import numpy as np
import torch


def tensor2bytes(t, value_range=(0, 1)):
    """maua/ops/io.py:47-70: [1,C,H,W] -> uint8 [H,W,C]; clamp, shift, true division by the range width, * 255, round half to
    even - every step rounded to float32 (the values of a host tensor; pinned by g14 / g26)."""
    mn, mx = value_range
    return t.squeeze(0).permute(1, 2, 0).clamp(mn, mx).sub(mn).div(mx - mn).mul(255).round().byte().cpu().numpy()


def frames_to_u8(img):
    """render/ffmpeg.py:72 (.add(1).div(2)) followed by tensor2bytes per frame: [B,3,H,W] -> [B,H,W,3] u8."""
    return np.stack([tensor2bytes(f[None].add(1).div(2)) for f in img])


def parse_seeds(seeds):
    """maua/GAN/wrappers/stylegan.py:59-65: "a-b,c" -> ints, ranges end-exclusive."""
    out = []
    for s in seeds.split(","):
        if "-" in s:
            a, b = s.split("-")
            out += list(range(int(a), int(b)))
        else:
            out.append(int(s))
    return out


def get_z_latents(seeds, z_dim=512):
    """stylegan.py:66-69 — float64 [P, z_dim] from numpy's MT19937 RandomState(seed).randn."""
    return torch.cat([torch.from_numpy(np.random.RandomState(s).randn(1, z_dim)) for s in parse_seeds(seeds)])
