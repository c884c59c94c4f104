"""Probe: do two batches in flight on two HIP streams (two synthesis objects) raise the frame rate?"""
import sys, time, torch
sys.path.insert(0, ".")
from maua_amd.stylegan2 import SynthesisNetwork
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 2
nets = [SynthesisNetwork(512, 1024, 3, dtype=torch.bfloat16, generator=torch.Generator().manual_seed(0)).cuda() for _ in range(NS)]
g = torch.Generator().manual_seed(1)
ws = torch.randn(B, nets[0].num_ws, 512, generator=g).cuda()
u8 = [torch.empty((B, 1024, 1024, 3), dtype=torch.uint8, device="cuda") for _ in range(NS)]
streams = [torch.cuda.Stream() for _ in range(NS)]
def run(n_steps, two):
    torch.cuda.synchronize()
    t0 = time.time()
    for k in range(n_steps):
        i = k % NS if two else 0
        with torch.cuda.stream(streams[i]):
            nets[i](ws, rgb8_out=u8[i])
    torch.cuda.synchronize()
    return B * n_steps / (time.time() - t0)
for two in (False, True, False, True):
    run(4, two)
    print(f"{NS} streams B={B}" if two else f"one stream B={B}", f"{run(40, two):.1f} frames/s")
