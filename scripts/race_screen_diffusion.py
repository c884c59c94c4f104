"""Race screen of the round-5 diffusion kernels: N bit-exact re-runs of (a) the full-size UNet's kept forward + input gradient, (b) the
guided loop (graph, forked guidance branch, split-f32 secondary model), (c) the attention / GroupNorm input-gradient operators on ragged
shapes.  python scripts/race_screen_diffusion.py [N]"""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
from maua_amd import _lib as L
from maua_amd.diffusion import GuidedDiffusion, ImageTarget, MSEGuide, create_models

N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
model, diffusion, secondary = create_models("uncondImageNet256", "ddim20", allow_random_init=True, use_secondary=True,
                                            generator=torch.Generator().manual_seed(0))
g = torch.Generator().manual_seed(1)
B = 4
x = torch.randn(B, 3, 256, 256, generator=g).cuda()
t = torch.full((B,), 500.0).cuda()
go = torch.randn(B, 6, 256, 256, generator=g).cuda()
model.forward_keep(x, t)
ref = model.vjp(go).clone()
bad_vjp = 0
for _ in range(N):
    model.forward_keep(x, t)
    bad_vjp += int(not torch.equal(model.vjp(go), ref))
print(f"UNet kept forward + input gradient, {N} re-runs at batch {B}: mismatching runs {bad_vjp}")

gd = GuidedDiffusion([MSEGuide(1000.0)], timesteps=20, model=model, diffusion=diffusion, secondary_model=secondary)
tgt = [ImageTarget(torch.randn(3, 256, 256, generator=g) * 0.5)]
nz = torch.randn(B, 3, 256, 256, generator=g)
img = torch.randn(B, 3, 256, 256, generator=g)
r0 = gd.forward(img, tgt, 0.3, t_end=0.8, noise=nz).clone()
bad_loop = 0
for _ in range(max(3, N // 5)):
    bad_loop += int(not torch.equal(gd.forward(img, tgt, 0.3, t_end=0.8, noise=nz), r0))
print(f"guided loop (graph {model.guided_graph_active()}), {max(3, N // 5)} re-runs of 10 steps: mismatching runs {bad_loop}")

lib, ctx = L.lib(), L.ctx()
bad_ops = 0
for T, heads, ch in ((200, 2, 64), (1024, 8, 64), (48, 1, 32)):
    qkv = torch.randn(2, T, 3 * heads * ch, generator=g).cuda().bfloat16()
    do = torch.randn(2, T, heads * ch, generator=g).cuda().bfloat16()
    a, b = torch.empty_like(qkv), torch.empty_like(qkv)
    L.check(lib.maua_attention_legacy_vjp(ctx, L.ptr(qkv), L.ptr(do), L.ptr(a), 2, T, heads, ch, L.BF16))
    for _ in range(N):
        L.check(lib.maua_attention_legacy_vjp(ctx, L.ptr(qkv), L.ptr(do), L.ptr(b), 2, T, heads, ch, L.BF16))
        bad_ops += int(not torch.equal(a, b))
print(f"attention input gradient, 3 shapes x {N} re-runs: mismatching runs {bad_ops}")
sys.exit(1 if bad_vjp + bad_loop + bad_ops else 0)
