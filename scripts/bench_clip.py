"""Text-prompt guidance (CLIPGrads, maua/grad.py:96-165) on one GPU: time of one grad-module call at configs[3]'s shape - `batch`
256^2 image estimates, `cutn` cutouts x `batches` cutout batches through a random-init ViT-B/16 image tower forward AND backward -
and the algorithmic FLOP rate.  `python scripts/bench_clip.py [--batch 32] [--batches 8] [--reps 3]`; under rocprofv3 for the
per-kernel split (scripts/prof_clip.sh)."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def vit_gflop(res=224, patch=16, width=768, layers=12, heads=12, out=512, backward=True):
    """Algorithmic GFLOP per image: patch embedding + per layer the four Linear layers and the two attention products (+ the head);
    the input gradient repeats every Linear against the transposed weight and costs four attention products (dV, dP, dQ, dK)."""
    G = res // patch
    T, kp = G * G + 1, 3 * patch * patch
    lin = T * (width * 3 * width + width * width + 2 * width * 4 * width)
    att = 2 * T * T * width
    fwd = G * G * kp * width + layers * (lin + att) + width * out
    bwd = G * G * kp * width + layers * (lin + 2 * att) + width * out if backward else 0
    return 2 * (fwd + bwd) / 1e9


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--cutn", type=int, default=32)
    ap.add_argument("--batches", type=int, default=8)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--dma", type=int, default=1)
    a = ap.parse_args()
    import ctypes as C
    from maua_amd import _lib as L
    from maua_amd.clip import load
    from maua_amd.grad import CLIPGrads, EmbeddingPrompt
    model, _ = load("ViT-B/16", allow_random_init=True, generator=torch.Generator().manual_seed(0))
    L.check(L.lib().maua_ctx_set_option(L.ctx(), b"gemm_dma", a.dma))
    gm = CLIPGrads(scale=1000.0, clip_models=[model], cutout_kwargs=dict(cutn=a.cutn), cutout_batches=a.batches)
    g = torch.Generator().manual_seed(1)
    gm.set_targets([EmbeddingPrompt(torch.randn(512, generator=g)), EmbeddingPrompt(torch.randn(512, generator=g), 0.5)])
    img = (torch.rand(a.batch, 3, a.size, a.size, generator=g) * 2 - 1).cuda()
    t = torch.full((a.batch,), 500.0)
    torch.manual_seed(2)
    out = gm(img, t)
    torch.cuda.synchronize()
    best = None
    for _ in range(a.reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = gm(img, t)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    n_img = a.batch * a.cutn * a.batches
    gf = vit_gflop() * n_img
    print({"batch": a.batch, "cutn": a.cutn, "cutout_batches": a.batches, "images_per_call": n_img, "seconds_per_call": best,
           "ms_per_cutout_batch": best / a.batches * 1e3, "gflop_per_image_fwd_bwd": vit_gflop(), "tflops": gf / best / 1e3,
           "frac_of_bf16_peak": gf / best / 1e3 / 2500.0, "finite": bool(torch.isfinite(out).all()), "gemm_dma": a.dma,
           "hbm_gb_allocated": torch.cuda.mem_get_info()[1] / 1e9 - torch.cuda.mem_get_info()[0] / 1e9})


if __name__ == "__main__":
    main()
