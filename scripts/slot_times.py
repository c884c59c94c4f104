"""GPU box: per-launch times of the 1024^2 forward (profile slots) for one library build; run once per MAUA_HIP_LIB.
python scripts/slot_times.py [B] [reps] [opt=val ...]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, ".")
from maua_amd import _lib as L
from maua_amd.stylegan2 import SynthesisNetwork

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
net = SynthesisNetwork(512, 1024, 3, dtype=torch.bfloat16, generator=torch.Generator().manual_seed(0))
ws = torch.randn(B, net.num_ws, 512, generator=torch.Generator().manual_seed(1)).cuda()
u8 = torch.empty((B, 1024, 1024, 3), dtype=torch.uint8, device="cuda")
h = net._handle()
lib = L.lib()
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    L.check(lib.maua_synth_set_option(h, k.encode(), int(v)))
for _ in range(3):
    net(ws, rgb8_out=u8)
L.check(lib.maua_synth_set_option(h, b"profile", 1))
for _ in range(reps):
    net(ws, rgb8_out=u8)
n = C.c_int()
L.check(lib.maua_synth_get_profile(h, None, 0, C.byref(n)))
ms = (C.c_float * n.value)()
L.check(lib.maua_synth_get_profile(h, ms, n.value, C.byref(n)))
per = n.value // reps
avg = [sum(ms[f * per + j] for f in range(reps)) / reps for j in range(per)]
names = {9: "up16", 12: "up32", 13: "fir32", 14: "c1_64", 16: "up64", 17: "fir64", 18: "c1_128", 20: "up128", 21: "fir128", 22: "c1_256", 24: "up256", 25: "fir256", 26: "c1_512", 28: "walk"}
tag = os.path.basename(os.environ.get("MAUA_HIP_LIB", "tree"))
print(f"{tag:28s} " + " ".join(f"{names[j]}={avg[j]:.3f}" for j in sorted(names)) + f" | sum {sum(avg):.3f}")
if os.environ.get("MAUA_SLOTS_ALL"):   # every profile slot of the forward, in launch order
    print("  all slots (ms): " + " ".join(f"{j}:{avg[j]:.3f}" for j in range(per)))
