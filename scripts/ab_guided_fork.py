"""GPU box: the guided loop (speed "fast", split-f32 secondary model) with the guidance branch beside / behind the UNet forward.
python scripts/ab_guided_fork.py [batch] [steps]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maua_amd.diffusion import GuidedDiffusion, ImageTarget, MSEGuide, create_models
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
S = int(sys.argv[2]) if len(sys.argv) > 2 else 30
model, diffusion, secondary = create_models("uncondImageNet256", f"ddim{S}", allow_random_init=True, use_secondary=True,
                                            generator=torch.Generator().manual_seed(0))
g = torch.Generator().manual_seed(1)
x0, nz = torch.randn(B, 3, 256, 256, generator=g).cuda(), torch.randn(B, 3, 256, 256, generator=g).cuda()
tg = [ImageTarget(torch.randn(3, 256, 256, generator=g) * 0.5)]
gd = GuidedDiffusion([MSEGuide(1000.0)], timesteps=S, model=model, diffusion=diffusion, secondary_model=secondary)
n = diffusion.num_timesteps
for fork in (1, 0, 1, 0):
    model.set_option("guided_fork", fork)
    gd.run(x0, tg, n - 1, n, noise=nz)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    gd.run(x0, tg, n - 1, n, noise=nz)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"batch {B}, guided_fork {fork}: {dt / S * 1e3:.2f} ms per step", flush=True)
