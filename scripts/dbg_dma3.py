import sys, torch
sys.path.insert(0, ".")
import maua_amd.ops as M
from maua_amd import _lib as L
B, C, h, w = 1, 128, 8, 32
ctx = L.ctx(torch.device("cuda"))
L.check(L.lib().maua_ctx_set_option(ctx, b"dma_conv", 1))
s = torch.ones(B, C)
def run(x, wt):
    return M.modulated_conv2d(x.bfloat16().cuda(), wt, s, padding=1, demodulate=False).float().cpu()
xc = torch.arange(C).float().view(1, C, 1, 1).expand(B, C, h, w).contiguous()
for t in (4, 0):
    for k0 in (5, 37, 70, 101):
        wt = torch.zeros(C, C, 3, 3); wt[:, k0, t // 3, t % 3] = 1.0   # every out channel reads in-channel k0
        y = run(xc, wt)
        print(f"tap {t} k0 {k0}: out[:,4,16] unique values:", sorted(set(int(v) for v in y[0, :, 4, 16])))
    # row probe: W[c][k0=5] = c  (x = 1) -> out[c] = c
    wt = torch.zeros(C, C, 3, 3); wt[:, 5, t // 3, t % 3] = torch.arange(C).float()
    y = run(torch.ones(B, C, h, w), wt)
    print(f"tap {t} row probe k0=5:", [int(v) for v in y[0, :48, 4, 16]])
    wt = torch.zeros(C, C, 3, 3); wt[:, 37, t // 3, t % 3] = torch.arange(C).float()
    y = run(torch.ones(B, C, h, w), wt)
    print(f"tap {t} row probe k0=37:", [int(v) for v in y[0, :48, 4, 16]])
