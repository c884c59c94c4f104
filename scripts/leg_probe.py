"""A few steps of one extra bench leg at the leg's OWN batch, launch by launch (no hipGraph), for the counter passes bench.py runs over it
(live_leg_traffic: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, separate child runs):
    python scripts/leg_probe.py diffusion [--batch 32] [--steps 3]     the GUIDED loop (speed "fast", image-MSE module) - one
                                                                       maua::ddim_step_kernel per step marks the steps
    python scripts/leg_probe.py upscale [--frames 4] [--steps 2]       render -> RealESRGAN x4 through the product path - one
                                                                       upwalk_fused_kernel per synthesis call marks the batches"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))

ap = argparse.ArgumentParser()
ap.add_argument("leg", choices=("diffusion", "upscale"))
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--frames", type=int, default=4)
a = ap.parse_args()
if a.leg == "diffusion":
    from maua_amd.diffusion import GuidedDiffusion, ImageTarget, MSEGuide, create_models
    model, diffusion, secondary = create_models("uncondImageNet256", "ddim100", allow_random_init=True, use_secondary=True,
                                                generator=torch.Generator().manual_seed(0))
    g = torch.Generator().manual_seed(3)
    prompts = [ImageTarget(torch.randn(3, 256, 256, generator=g).clamp(-1, 1) * 0.5 + s_) for s_ in (-0.4, 0.4)]
    gd = GuidedDiffusion([MSEGuide(1000.0)], timesteps=100, model=model, diffusion=diffusion, secondary_model=secondary)
    gd.use_graph = False
    x0, nz = (torch.randn(a.batch, 3, 256, 256, generator=g).cuda() for _ in range(2))
    n = diffusion.num_timesteps
    out = gd.run(x0, [prompts[j % 2] for j in range(a.batch)], n - 1, a.steps, noise=nz, per_sample=True)
    torch.cuda.synchronize()
    print("diffusion probe:", a.batch, a.steps, bool(torch.isfinite(out).all()))
else:
    import bench
    r = bench.extra_upscale(steps=a.steps, frames=a.frames, upscale_batch=4, traffic=False)
    print("upscale probe:", r.get("value"))
