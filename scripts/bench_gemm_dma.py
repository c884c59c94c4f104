"""The CLIP tower's GEMM shapes (M = 1024 images x 197 tokens) through maua_linear_nt on gemm_dma.hip (ctx option "linear_dma"):
ms and TFLOP/s per shape, checked against torch on a slice.  `python scripts/bench_gemm_dma.py [--reps 5]`"""
import argparse
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from maua_amd import _lib as L  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--m", type=int, default=1024 * 197)
ap.add_argument("--dma", type=int, default=1)
a = ap.parse_args()
lib, ctx = L.lib(), L.ctx()
L.check(lib.maua_ctx_set_option(ctx, b"linear_dma", a.dma))
g = torch.Generator().manual_seed(0)
tot_f, tot_t = 0.0, 0.0
for (N, K, name) in ((768, 768, "conv1 / out_proj"), (2304, 768, "in_proj"), (3072, 768, "c_fc"), (768, 3072, "c_proj"), (768, 2304, "in_proj^T")):
    M = a.m
    x = (torch.rand(M, K, generator=g) * 2 - 1).to(torch.bfloat16).cuda()
    w = ((torch.rand(N, K, generator=g) * 2 - 1) * K ** -0.5).to(torch.bfloat16).cuda()
    b = torch.randn(N, generator=g).cuda()
    y = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    def run():
        L.check(lib.maua_linear_nt(ctx, L.ptr(x), L.ptr(w), L.ptr(b), None, L.ptr(y), C.c_long(M), N, K, L.BF16))
    run()
    torch.cuda.synchronize()
    ref = x[:512].float() @ w.float().t() + b
    err = float((y[:512].float() - ref).abs().max()) / float(ref.abs().max())
    ref2 = x[-300:].float() @ w.float().t() + b
    err2 = float((y[-300:].float() - ref2).abs().max()) / float(ref2.abs().max())
    t0 = time.perf_counter()
    for _ in range(a.reps):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.reps
    fl = 2.0 * M * N * K
    tot_f += fl; tot_t += dt
    print(f"{name:18s} M={M} N={N} K={K}: {dt * 1e3:7.3f} ms  {fl / dt / 1e12:7.1f} TFLOP/s  rel-err {err:.1e} {err2:.1e}")
print(f"all five: {tot_f / tot_t / 1e12:.1f} TFLOP/s")
