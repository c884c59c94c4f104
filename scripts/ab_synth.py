"""Same-process A/B of synthesis-network options (GPU box): interleaved rounds, median / min per arm, per-launch
profile of each arm.   python scripts/ab_synth.py <option> [B] [rounds]     e.g.  dma_conv 32 5"""
import ctypes as C
import statistics
import sys
import time

import torch

sys.path.insert(0, ".")
from maua_amd import _lib as L
from maua_amd.stylegan2 import SynthesisNetwork

opt = sys.argv[1] if len(sys.argv) > 1 else "dma_conv"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 5
V = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [0, 1]  # the option values to compare
res = 1024
net = SynthesisNetwork(512, res, 3, dtype=torch.bfloat16, generator=torch.Generator().manual_seed(0))
ws = torch.randn(B, net.num_ws, 512, generator=torch.Generator().manual_seed(1)).cuda()
u8 = {v: torch.empty((B, res, res, 3), dtype=torch.uint8, device="cuda") for v in V}
img = torch.empty((B, 3, res, res), device="cuda")
h = net._handle()
lib = L.lib()


def run(v, n, out=None):
    L.check(lib.maua_synth_set_option(h, opt.encode(), v))
    for _ in range(n):
        net(ws, rgb8_out=u8[v], out=out)


for v in V:
    run(v, 2)
torch.cuda.synchronize()
# agreement of the two arms
run(V[0], 1, img); a = img.clone(); run(V[1], 1, img); b = img.clone()
torch.cuda.synchronize()
d = (a - b).abs()
rng = float(a.max() - a.min())
mse = float(((a - b) ** 2).mean())
print(f"arms differ: max {float(d.max()):.3e} of range {rng:.3f}; PSNR(a,b) = {10 * torch.log10(torch.tensor(rng * rng / max(mse, 1e-30))).item():.1f} dB; "
      f"u8 frames differ in {float((u8[V[0]] != u8[V[1]]).float().mean()) * 100:.3f} % of bytes")
times = {v: [] for v in V}
for r in range(rounds):
    for v in V:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(v, 10)
        torch.cuda.synchronize()
        times[v].append((time.perf_counter() - t0) / 10 * 1e3)
for v in V:
    t = times[v]
    print(f"{opt}={v}: median {statistics.median(t):.3f} ms/forward  min {min(t):.3f}  -> {B / statistics.median(t) * 1e3:.0f} frames/s")
names = ["styles"]
for v in V:
    L.check(lib.maua_synth_set_option(h, opt.encode(), v))
    L.check(lib.maua_synth_set_option(h, b"profile", 1))
    for _ in range(5):
        net(ws, rgb8_out=u8[v])
    n = C.c_int()
    L.check(lib.maua_synth_get_profile(h, None, 0, C.byref(n)))
    ms = (C.c_float * n.value)()
    L.check(lib.maua_synth_get_profile(h, ms, n.value, C.byref(n)))
    L.check(lib.maua_synth_set_option(h, b"profile", 0))
    per = n.value // 5
    avg = [sum(ms[f * per + j] for f in range(5)) / 5 for j in range(per)]
    print(f"{opt}={v}: per-launch ms ({per} slots): " + " ".join(f"{x:.3f}" for x in avg) + f"  | sum {sum(avg):.3f}")
