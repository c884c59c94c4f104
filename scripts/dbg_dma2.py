import sys, torch
sys.path.insert(0, ".")
import maua_amd.ops as M
from maua_amd import _lib as L
B, C, h, w = 1, 128, 8, 32
# x[c][y][x] encodes (c, y, x) exactly in bf16-friendly small ints: use value = c (channel id) + y*0 ... do separate probes
ctx = L.ctx(torch.device("cuda"))
L.check(L.lib().maua_ctx_set_option(ctx, b"dma_conv", 1))
s = torch.ones(B, C)
def run(x, wt):
    return M.modulated_conv2d(x.bfloat16().cuda(), wt, s, padding=1, demodulate=False).float().cpu()
for t in (4, 0, 8):
    wt = torch.zeros(C, C, 3, 3)
    for c in range(C):
        wt[c, c, t // 3, t % 3] = 1.0
    # probe 1: value = channel index
    xc = torch.arange(C).float().view(1, C, 1, 1).expand(B, C, h, w).contiguous()
    yc = run(xc, wt)
    # interior pixel (4, 16)
    print(f"tap {t}: out channel -> in channel seen at (4,16):", [int(v) for v in yc[0, :, 4, 16][:40]], "...", [int(v) for v in yc[0, 96:104, 4, 16]])
    # probe 2: value = x coordinate, 3: y coordinate
    xx = torch.arange(w).float().view(1, 1, 1, w).expand(B, C, h, w).contiguous()
    yx = run(xx, wt)
    print(f"   x-coordinate seen by out (c=0,y=4):", [int(v) for v in yx[0, 0, 4]])
    xy = torch.arange(h).float().view(1, 1, h, 1).expand(B, C, h, w).contiguous() + 1
    yy = run(xy, wt)
    print(f"   y-coordinate(+1) seen by out (c=0,x=16):", [int(v) for v in yy[0, 0, :, 16]], " (c=70):", [int(v) for v in yy[0, 70, :, 16]])
