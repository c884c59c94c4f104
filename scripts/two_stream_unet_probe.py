"""GPU box probe: one UNet forward at batch 16 against two half-batch forwards on two streams (two network objects: each has its own
arena), to see whether the small-grid launches of one half hide under the other's large ones.  python scripts/two_stream_unet_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maua_amd.diffusion import create_models

gen = lambda: torch.Generator().manual_seed(0)
m1, _, _ = create_models("uncondImageNet256", timestep_respacing="ddim100", allow_random_init=True, generator=gen())
m2, _, _ = create_models("uncondImageNet256", timestep_respacing="ddim100", allow_random_init=True, generator=gen())
x = torch.randn(16, 3, 256, 256, generator=torch.Generator().manual_seed(1)).cuda()
t = torch.full((16,), 500.0).cuda()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
o1, o2 = torch.empty(8, 6, 256, 256, device="cuda"), torch.empty(8, 6, 256, 256, device="cuda")
xa, xb, ta, tb = x[:8].contiguous(), x[8:].contiguous(), t[:8].contiguous(), t[8:].contiguous()


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def two():
    with torch.cuda.stream(s1):
        m1(xa, ta, out=o1)
    with torch.cuda.stream(s2):
        m2(xb, tb, out=o2)


def seq():
    m1(xa, ta, out=o1)
    m2(xb, tb, out=o2)


print("one forward, batch 16: %.2f ms" % timed(lambda: m1(x, t)))
print("two forwards of batch 8, one stream: %.2f ms" % timed(seq))
print("two forwards of batch 8, two streams: %.2f ms" % timed(two))
