"""GPU box: the secondary model's forward + VJP (what a "fast"-guided step adds) at batch 16, 256 x 256, per precision mode.
python scripts/time_secondary.py [mode ...]   modes: split exact bf16"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maua_amd.diffusion import SecondaryDiffusionImageNet2

B, S = 16, 256
x = torch.randn(B, 3, S, S, generator=torch.Generator().manual_seed(0)).cuda()
t = torch.full((B,), 0.5).cuda()
g = torch.randn(B, 3, S, S, generator=torch.Generator().manual_seed(1)).cuda()
for mode in (sys.argv[1:] or ["split", "exact", "bf16"]):
    net = SecondaryDiffusionImageNet2(dtype=torch.bfloat16 if mode == "bf16" else torch.float32, exact=mode != "split")
    net(x, t); net.vjp(g)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        net(x, t)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(10):
        net(x, t); net.vjp(g)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{mode:6s} forward {(t1 - t0) * 100:.2f} ms   forward + vjp {(t2 - t1) * 100:.2f} ms", flush=True)
    del net
