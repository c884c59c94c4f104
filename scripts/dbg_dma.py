import sys, torch
sys.path.insert(0, ".")
from math import sqrt
import maua_amd.ops as M
from maua_amd import _lib as L
B, ci, co, h, w = [int(x) for x in sys.argv[1:6]] if len(sys.argv) > 5 else (1, 64, 128, 8, 32)
g = torch.Generator().manual_seed(1)
x = torch.randn(B, ci, h, w, generator=g).bfloat16().cuda()
wt = torch.randn(co, ci, 3, 3, generator=g)
s = torch.ones(B, ci)
ctx = L.ctx(x.device)
ys = []
for v in (0, 1):
    L.check(L.lib().maua_ctx_set_option(ctx, b"dma_conv", v))
    ys.append(M.modulated_conv2d(x, wt, s, padding=1, demodulate=False).float().cpu())
L.check(L.lib().maua_ctx_set_option(ctx, b"dma_conv", 1))
d = (ys[0] - ys[1]).abs()
print("max err", float(d.max()), "ref max", float(ys[0].abs().max()))
print("err by row   ", [round(float(d[:, :, i].max()), 2) for i in range(h)])
print("err by col   ", [round(float(d[:, :, :, j].max()), 2) for j in range(w)])
print("err by ch/16 ", [round(float(d[:, c:c + 16].max()), 2) for c in range(0, co, 16)])
# which input channels contribute wrongly: one-hot channel probes
for c in range(0, ci, 8):
    xx = torch.zeros_like(x); xx[:, c:c + 8] = x[:, c:c + 8]
    yy = []
    for v in (0, 1):
        L.check(L.lib().maua_ctx_set_option(ctx, b"dma_conv", v))
        yy.append(M.modulated_conv2d(xx, wt, s, padding=1, demodulate=False).float().cpu())
    print(f"ci {c:3d}..{c+7}: err {float((yy[0]-yy[1]).abs().max()):.3f}")
L.check(L.lib().maua_ctx_set_option(ctx, b"dma_conv", 1))
