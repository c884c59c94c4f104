import sys, torch
sys.path.insert(0, ".")
from maua_amd import _lib as L
from maua_amd.stylegan2 import SynthesisNetwork
B = 2
net = SynthesisNetwork(512, 1024, 3, dtype=torch.bfloat16, generator=torch.Generator().manual_seed(0))
ws = torch.randn(B, net.num_ws, 512, generator=torch.Generator().manual_seed(1)).cuda()
h = net._handle()
img = {}
for v in (1, 2):
    L.check(L.lib().maua_synth_set_option(h, b"upwalk", v))
    out = torch.empty((B, 3, 1024, 1024), device="cuda")
    net(ws, out=out)
    img[v] = out.clone()
d = (img[1] - img[2]).abs().amax(1)   # [B, H, W]
print("max", float(d.max()))
bad = (d > 1e-3).nonzero()
print("n bad", len(bad))
if len(bad):
    print("rows", torch.unique(bad[:, 1])[:40].tolist())
    print("cols", torch.unique(bad[:, 2])[:60].tolist())
    print(bad[:10].tolist())
