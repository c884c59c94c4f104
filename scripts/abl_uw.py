"""timing of one synthesis forward per ablation arm of the fused walk (MAUA_UW_ABL is read once per process)"""
import os, sys, time, torch
sys.path.insert(0, ".")
from maua_amd import _lib as L
from maua_amd.stylegan2 import SynthesisNetwork
B = 32
net = SynthesisNetwork(512, 1024, 3, dtype=torch.bfloat16, generator=torch.Generator().manual_seed(0))
ws = torch.randn(B, net.num_ws, 512, generator=torch.Generator().manual_seed(1)).cuda()
u8 = torch.empty((B, 1024, 1024, 3), dtype=torch.uint8, device="cuda")
h = net._handle()
L.check(L.lib().maua_synth_set_option(h, b"profile", 1))
for _ in range(8):
    net(ws, rgb8_out=u8)
torch.cuda.synchronize()
import ctypes as C
n = C.c_int(0)
L.check(L.lib().maua_synth_get_profile(h, None, 0, C.byref(n)))
ms = (C.c_float * n.value)()
L.check(L.lib().maua_synth_get_profile(h, ms, n.value, C.byref(n)))
per = n.value // 8
print("abl", os.environ.get("MAUA_UW_ABL", "0"), "walk slot ms:", " ".join(f"{ms[(8 - 1) * per + i]:.3f}" for i in range(per - 6, per)))
