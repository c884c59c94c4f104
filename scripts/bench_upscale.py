"""configs[4] leg: 1024^2 StyleGAN2 frame -> RealESRGAN x4 (23 RRDB blocks, random init) -> 4096^2 u8, per frame on one GPU."""
import sys, time, torch
sys.path.insert(0, ".")
from maua_amd.stylegan2 import SynthesisNetwork
from maua_amd.super import RRDBNet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
res = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
G = SynthesisNetwork(512, res, 3, dtype=torch.bfloat16, generator=torch.Generator().manual_seed(0))
S = RRDBNet(num_block=23, dtype=torch.bfloat16)
ws = torch.randn(B, G.num_ws, 512, generator=torch.Generator().manual_seed(1)).cuda()
img = torch.empty((B, 3, res, res), device="cuda")
u8 = torch.empty((B, 4 * res, 4 * res, 3), dtype=torch.uint8, device="cuda")
def step():
    G(ws, out=img)
    x = img.add(1).div(2).clamp_(0, 1)
    S(x, rgb8_out=u8)
step(); torch.cuda.synchronize()
t0 = time.perf_counter(); n = 3
for _ in range(n): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f"render + x4 upscale, B={B}, {res}^2 -> {4*res}^2: {dt*1e3:.1f} ms/step = {B/dt:.2f} frames/s")
