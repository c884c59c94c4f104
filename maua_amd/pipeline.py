"""Clip-level plumbing shared by bench.py, the entry points and the tests: frame-range sharding and the synthetic
BASELINE clip."""
import math

import torch


def frame_range(n_frames, rank, world):
    """Contiguous frame range [lo, hi) of `rank` (frames are independent after the clip-global pre-pass;
    SURVEY 8e).  Ranges differ by at most one frame and cover [0, n_frames) exactly."""
    base, rem = divmod(n_frames, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def synthetic_audio(n_samples, sr, seed=1234):
    """SURVEY 8(d) synthetic clip: 220 Hz tone + 2 Hz click train + noise, float32 mono."""
    g = torch.Generator().manual_seed(seed)
    t = torch.arange(n_samples, dtype=torch.float64).div_(sr)
    u = torch.rand(n_samples, generator=g, dtype=torch.float64)
    nz = torch.randn(n_samples, generator=g, dtype=torch.float64)
    # 0.3 sin(2 pi 220 t) + 0.2 (u - 0.5) [2t mod 1 < 0.05] + 0.01 nz, in place (eight 29 MB temporaries cost the
    # 3600-frame clip 0.25 s of page faults)
    u.sub_(0.5).mul_(0.2).mul_(t.mul(2).remainder_(1).lt_(0.05))
    t.mul_(2 * math.pi * 220).sin_().mul_(0.3).add_(u).add_(nz.mul_(0.01))
    return t.float()


def synthetic_clip_latents(n_frames, fps, num_ws, w_dim, seeds="0-60", n_loops=4):
    """Audio-reactive latent schedule of the BASELINE clip: onset envelope of the synthetic audio blends two
    spline-loop schedules (latent.py:12-18 single_weighted over latent.py:83-92 spline_loops), sigma=2 smoothing.
    Returns ([T, num_ws, w_dim] f32 on the HIP device, description)."""
    from . import audio, latent
    from .stylegan2 import MappingNetwork, get_z_latents
    sr = 1024 * fps
    wav = synthetic_audio(n_frames * 1024, sr)
    env = audio.onsets(wav, sr).squeeze(-1)                      # [T], on device
    mapper = MappingNetwork(w_dim, 0, w_dim, num_ws, generator=torch.Generator().manual_seed(0))
    palette = mapper(get_z_latents(seeds, w_dim).float())        # [P, num_ws, w_dim]
    half = palette.shape[0] // 2
    low = latent.spline_loops(palette[:half], n_frames, n_loops)
    high = latent.spline_loops(palette[half:2 * half], n_frames, n_loops)
    lat = latent.sequence_weighted(low, high, env)
    lat = audio.gaussian_filter(lat, 2)
    return lat.contiguous(), {"seeds": seeds, "schedule": f"spline_loops(n_loops={n_loops}) x2 blended by onsets, gaussian sigma=2"}
