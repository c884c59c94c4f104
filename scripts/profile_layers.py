"""Per-launch HIP-event timings of one synthesis forward (GPU box).  python scripts/profile_layers.py [B] [res] [dtype]"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, ".")
from maua_amd import _lib as L
from maua_amd.stylegan2 import SynthesisNetwork

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
res = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dt = torch.float32 if (len(sys.argv) > 3 and sys.argv[3] == "f32") else torch.bfloat16
t0 = time.time()
net = SynthesisNetwork(512, res, 3, dtype=dt, generator=torch.Generator().manual_seed(0))
ws = torch.randn(B, net.num_ws, 512, generator=torch.Generator().manual_seed(1)).cuda()
out = torch.empty((B, 3, res, res), device="cuda")
u8 = torch.empty((B, res, res, 3), dtype=torch.uint8, device="cuda")
net(ws, out=out, rgb8_out=u8)
torch.cuda.synchronize()
print(f"setup+first forward {time.time()-t0:.1f}s")
h = net._handle()
lib = L.lib()
import os
for opt in ("use_hires", "fuse_torgb", "tconv_up", "lowres", "dma_conv", "tconv_dma"):
    if os.environ.get("MAUA_" + opt.upper()) is not None:
        L.check(lib.maua_synth_set_option(h, opt.encode(), int(os.environ["MAUA_" + opt.upper()])))
for it in range(3):
    net(ws, out=out, rgb8_out=u8)
torch.cuda.synchronize()
t0 = time.time()
for it in range(5):
    net(ws, out=out, rgb8_out=u8)
torch.cuda.synchronize()
dtm = (time.time() - t0) / 5
print(f"B={B} res={res} {dt}: {dtm*1e3:.2f} ms/forward  -> {B/dtm:.1f} frames/s (un-profiled)")
L.check(lib.maua_synth_set_option(h, b"profile", 1))
net(ws, out=out, rgb8_out=u8)
n = C.c_int()
ms = (C.c_float * 64)()
L.check(lib.maua_synth_get_profile(h, ms, 64, C.byref(n)))
tc = int(os.environ.get("MAUA_TCONV_UP", "1"))  # mirrors synth.hip: 0 off, 1 = 32..512, v > 1 = every input <= v
tc_lo, tc_hi = (32, 512) if tc == 1 else (1, tc) if tc > 1 else (1, 0)
names = ["styles"]
shapes = net.layer_shapes()
li = 0
for i, r in enumerate(net.block_resolutions):
    for k in range(1 if i == 0 else 2):
        pfx, ci, co, rr, up = shapes[li]
        hires_up = tc == 1 and dt == torch.bfloat16 and (ci, co) == (64, 32) and os.environ.get('MAUA_USE_HIRES', '1') != '0'
        if up == 2 and tc_lo <= rr // up <= tc_hi and not hires_up:
            names.append("  (tconv part of next row)")
        names.append(shapes[li]); li += 1
    names.append(("torgb", shapes[li - 1][2], r))
names.append("pack_rgb8")
tot = 0
es = 2 if dt == torch.bfloat16 else 4
for i in range(n.value):
    nm = names[i]
    t = ms[i]
    tot += t
    if isinstance(nm, tuple) and nm[0] != "torgb":
        pfx, ci, co, r, up = nm
        hin = r // up
        gmac = B * hin * hin * 9 * ci * co / 1e9
        gmac_exec = gmac * (4 if up == 2 else 1)
        byts = B * (hin * hin * ci + r * r * co) * es
        print(f"{pfx:14s} {ci:4d}->{co:4d} {r:5d} up{up}: {t:8.3f} ms  alg {2*gmac/t:8.1f} TFLOP/s  exec {2*gmac_exec/t:8.1f} TFLOP/s  {byts/t/1e6:8.1f} GB/s")
    elif isinstance(nm, tuple):
        _, c, r = nm
        byts = B * (r * r * c * es + r * r * 3 * 4 + (r // 2) ** 2 * 3 * 4)
        print(f"{'torgb':14s} {c:4d}->   3 {r:5d}    : {t:8.3f} ms  {byts/t/1e6:8.1f} GB/s")
    else:
        print(f"{nm:14s}: {t:8.3f} ms")
print(f"sum {tot:.3f} ms -> {B/tot*1e3:.1f} fps (profiled)")
