"""Race screen of the image-prompt grad modules (round 6): N bit-exact re-runs of (a) VGGGrads / LPIPSGrads / ColorMatchGrads at 256 x 256
(forward, heads, hand-walked backward; the colour histogram's integer atomics), (b) flagged (grey / mirrored) cutouts and their adjoint,
(c) resample's adjoint, (d) the guided loop with the three modules as guides (graph, guidance branch beside the UNet forward).
python scripts/race_screen_grads.py [N]"""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from maua_amd import _lib as L
from maua_amd.diffusion import GuidedDiffusion, create_models
from maua_amd.grad import ColorMatchGrads, ContentPrompt, DangoCutouts, LPIPSGrads, StylePrompt, VGGGrads, _run_cutouts
from maua_amd.ops import resample_vjp

N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
g = torch.Generator().manual_seed(1)
B = 4
mods = {"VGGGrads": VGGGrads(scale=100.0, allow_random_init=True, generator=g), "ColorMatchGrads": ColorMatchGrads(scale=1e4),
        "LPIPSGrads": LPIPSGrads(scale=10.0, allow_random_init=True, generator=g)}
pr = [StylePrompt(img=torch.rand(1, 3, 256, 256, generator=g)).to("cuda"), ContentPrompt(img=torch.rand(1, 3, 256, 256, generator=g)).to("cuda")]
img = (torch.rand(B, 3, 256, 256, generator=g) * 2 - 1).cuda()
bad = 0
for name, m in mods.items():
    m.set_targets(pr)
    ref = m(img, None).clone()
    b = sum(int(not torch.equal(m(img, None), ref)) for _ in range(N))
    print(f"{name}, {N} re-runs at batch {B}: mismatching runs {b}")
    bad += b

dc = DangoCutouts(224, skip_augs=True)
torch.manual_seed(3)
rects = dc.rects(256, 256, 300)
ref = _run_cutouts(img, rects, 224, 0.5, 0.5).clone()
d = torch.randn_like(ref)
r = np.ascontiguousarray(np.asarray(rects, dtype=np.int32))
s3 = (C.c_float * 3)(1.0, 1.0, 1.0)
ga, gb = torch.empty_like(img), torch.empty_like(img)
L.check(L.lib().maua_cutouts_vjp(L.ctx(), L.ptr(d), B, 256, 256, r.ctypes.data_as(C.c_void_p), len(r), 224, C.c_float(0.5), s3, L.ptr(ga)))
b = 0
for _ in range(N):
    b += int(not torch.equal(_run_cutouts(img, rects, 224, 0.5, 0.5), ref))
    L.check(L.lib().maua_cutouts_vjp(L.ctx(), L.ptr(d), B, 256, 256, r.ctypes.data_as(C.c_void_p), len(r), 224, C.c_float(0.5), s3, L.ptr(gb)))
    b += int(not torch.equal(ga, gb))
print(f"flagged cutouts + adjoint (16 cutouts of 224^2), {N} re-runs: mismatching runs {b}")
bad += b

go = torch.randn(B, 3, 256, 256, generator=g).cuda()
ref = resample_vjp(go, (B, 3, 512, 384)).clone()
b = sum(int(not torch.equal(resample_vjp(go, (B, 3, 512, 384)), ref)) for _ in range(N))
print(f"resample adjoint 256 x 256 -> 512 x 384 ... (short side 256), {N} re-runs: mismatching runs {b}")
bad += b

model, diffusion, secondary = create_models("uncondImageNet256", "ddim20", allow_random_init=True, use_secondary=True,
                                            generator=torch.Generator().manual_seed(0))
gd = GuidedDiffusion(list(mods.values()), timesteps=20, model=model, diffusion=diffusion, secondary_model=secondary)
nz = torch.randn(B, 3, 256, 256, generator=g)
x0 = torch.randn(B, 3, 256, 256, generator=g)
r0 = gd.forward(x0, pr, 0.3, t_end=0.8, noise=nz).clone()
n_loop = max(3, N // 5)
b = sum(int(not torch.equal(gd.forward(x0, pr, 0.3, t_end=0.8, noise=nz), r0)) for _ in range(n_loop))
print(f"guided loop with three guides (graph {model.guided_graph_active()}), {n_loop} re-runs of 10 steps: mismatching runs {b}")
bad += b
sys.exit(1 if bad else 0)
