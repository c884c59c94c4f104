"""Per-launch time of the 512^2 conv1 with modconv_cwalk.hip under MAUA_CW_SKIP=<mask> (ablation of its phases; results are
wrong with any bit set):  for s in 0 1 2 4 8 6 14 15; do MAUA_CW_SKIP=$s python scripts/cwalk_ablation.py; done"""
import sys, torch, os
sys.path.insert(0, ".")
from maua_amd import _lib as L
from maua_amd.stylegan2 import SynthesisNetwork
B = 32
net = SynthesisNetwork(512, 1024, 3, dtype=torch.bfloat16, generator=torch.Generator().manual_seed(0))
ws = torch.randn(B, net.num_ws, 512, generator=torch.Generator().manual_seed(1)).cuda()
u8 = torch.empty((B, 1024, 1024, 3), dtype=torch.uint8, device="cuda")
h = net._handle(); lib = L.lib()
L.check(lib.maua_synth_set_option(h, b"cwalk", 1))
for _ in range(3): net(ws, rgb8_out=u8)
L.check(lib.maua_synth_set_option(h, b"profile", 1))
for _ in range(5): net(ws, rgb8_out=u8)
import ctypes as C
n = C.c_int(0)
L.check(lib.maua_synth_get_profile(h, None, 0, C.byref(n)))
ms = (C.c_float * n.value)()
L.check(lib.maua_synth_get_profile(h, ms, n.value, C.byref(n)))
per = n.value // 5
avg = [sum(ms[f * per + j] for f in range(5)) / 5 for j in range(per)]
print("MAUA_CW_SKIP", os.environ.get("MAUA_CW_SKIP"), "slot26 ms", round(avg[26], 3))
