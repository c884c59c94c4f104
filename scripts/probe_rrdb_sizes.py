"""GPU box: ms per megapixel of the x4plus network at different image sizes / batches - does a working set that fits the 256 MB
Infinity Cache run the HBM-bound dense blocks faster?  python scripts/probe_rrdb_sizes.py"""
import sys, time
sys.path.insert(0, ".")
import torch
from maua_amd.super import load_model

up = load_model("x4plus", dtype=torch.bfloat16, allow_random_init=True)
net = up.model
for B, H, W in [(4, 1024, 1024), (1, 1024, 1024), (1, 512, 1024), (1, 256, 1024), (1, 128, 1024), (1, 256, 256), (4, 256, 256), (16, 256, 256), (2, 128, 1024)]:
    x = torch.rand(B, 3, H, W, device="cuda")
    for _ in range(2):
        y = net(x, clamp=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        y = net(x, clamp=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    mp = B * H * W / 1e6
    dense_mb = B * H * W * 192 * 2 / 1e6
    print(f"B {B:2d} {H:4d}x{W:4d}: {dt * 1e3:8.2f} ms  {dt * 1e3 / mp:7.2f} ms/Mpx   dense buffer {dense_mb:7.1f} MB x 3", flush=True)
    del x, y
