"""Race screen: N bit-exact re-runs of a full-size batch on the default kernel routing.  python scripts/race_screen.py [N]"""
import sys
import torch
sys.path.insert(0, ".")
from maua_amd.stylegan2 import SynthesisNetwork

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
net = SynthesisNetwork(512, 1024, 3, dtype=torch.bfloat16, generator=torch.Generator().manual_seed(0))
g = torch.Generator().manual_seed(1)
B = 8
ws = torch.randn(B, net.num_ws, 512, generator=g).cuda()
noise = [torch.randn(B, 1, s[3], s[3], generator=g).cuda() for s in net.layer_shapes()]
u0 = torch.empty((B, 1024, 1024, 3), dtype=torch.uint8, device="cuda")
net(ws, noise=noise, rgb8_out=u0)
bad = 0
for it in range(N):
    u = torch.empty_like(u0)
    net(ws, noise=noise, rgb8_out=u)
    bad += int(not torch.equal(u, u0))
print(f"race screen: {N} re-runs of a 1024^2 batch of {B}, mismatching runs: {bad}")
sys.exit(1 if bad else 0)
