"""GPU box: maua_linear_nt (csrc/gemm.hip) on the guided-diffusion UNet's 1x1 / attention shapes at batch 16: ms and TFLOP/s."""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maua_amd import _lib as L

shapes = [("qkv 32^2", 16384, 1536, 512), ("proj 32^2", 16384, 512, 512), ("qkv 16^2", 4096, 3072, 1024), ("proj 16^2", 4096, 1024, 1024),
          ("qkv 8^2", 1024, 3072, 1024), ("skip 256^2 512->256", 1048576, 256, 512), ("skip 128^2 512->256", 262144, 256, 512),
          ("skip 64^2 768->512", 65536, 512, 768), ("skip 32^2 1024->512", 16384, 512, 1024), ("skip 16^2 2048->1024", 4096, 1024, 2048),
          ("skip 8^2 2048->1024", 1024, 1024, 2048)]
lib, ctx = L.lib(), L.ctx()
for name, M, N, K in shapes:
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = torch.randn(N, K, device="cuda").bfloat16()
    b = torch.randn(N, device="cuda")
    c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    f = lambda: L.check(lib.maua_linear_nt(ctx, L.ptr(a), L.ptr(w), L.ptr(b), None, L.ptr(c), C.c_long(M), N, K, L.BF16))
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        f()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    gb = (M * K + N * K + M * N) * 2 / 1e9
    print(f"{name:24s} M {M:8d} N {N:5d} K {K:5d}  {ms:7.3f} ms  {2.0 * M * N * K / ms / 1e9:7.1f} TFLOP/s  {gb / ms * 1e3:7.1f} GB/s", flush=True)
