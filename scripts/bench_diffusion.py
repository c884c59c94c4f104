"""BASELINE configs[3]: guided-diffusion 256x256 UNet (random init, the architecture of guided.py:171-190), 100-step DDIM.
Prints one JSON line: seconds per 100-step sample batch, samples/s, UNet forward ms (eager and inside the hipGraph loop),
algorithmic TFLOP/s.  python scripts/bench_diffusion.py [--batch 4] [--steps 100] [--size 256] [--reps 3]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def secondary_gflop(H, W, vjp=True):
    """algorithmic GFLOP per sample of SecondaryDiffusionImageNet2.forward (guided.py:77-134: 24 3x3 convolutions, two per level
    of a 6-level U) and, with ``vjp``, of the input-gradient pass the "fast" conditioning needs (the same convolutions transposed)."""
    c = [64, 128, 128, 256, 256, 512]
    ci = [19, c[0], c[0], c[1], c[1], c[2], c[2], c[3], c[3], c[4], c[4], c[5], c[5], c[5], 2 * c[4], c[4], 2 * c[3], c[3], 2 * c[2],
          c[2], 2 * c[1], c[1], 2 * c[0], c[0]]
    co = [c[0], c[0], c[1], c[1], c[2], c[2], c[3], c[3], c[4], c[4], c[5], c[5], c[5], c[4], c[4], c[3], c[3], c[2], c[2], c[1], c[1],
          c[0], c[0], 3]
    lev = [0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 5, 5, 4, 4, 3, 3, 2, 2, 1, 1, 0, 0]
    total = sum(2.0 * 9 * a * b * (H >> l) * (W >> l) for a, b, l in zip(ci, co, lev))
    return total * (2.0 if vjp else 1.0) / 1e9


def unet_gflop(net, H, W):
    """algorithmic GFLOP of one forward per sample: 3x3 and 1x1 convolutions + attention products (2 * MACs)."""
    s = net._structure
    hc = net.num_head_channels
    total = 0.0

    def walk(layers, h, w):
        nonlocal total
        for l in layers:
            if l[0] == "conv":
                total += 2 * 9 * l[1] * l[2] * h * w
            elif l[0] == "res":
                _, ci, co = l
                total += 2 * 9 * ci * co * h * w + 2 * 9 * co * co * h * w + (2 * ci * co * h * w if ci != co else 0)
            else:
                c = l[1]
                total += 2 * c * 3 * c * h * w + 2 * c * c * h * w + 4 * (h * w) ** 2 * c
        return h, w
    # resolutions: a level's down block (its convolutions run behind the average pool) / up block (behind the nearest x2)
    h, w = H, W
    nrb, L = net.num_res_blocks, len(net.channel_mult)
    walk(s["input"][0], h, w)
    i = 1
    for level in range(L):
        for _ in range(nrb):
            walk(s["input"][i], h, w)
            i += 1
        if level != L - 1:
            h, w = h // 2, w // 2
            walk(s["input"][i], h, w)
            i += 1
    walk(s["middle"], h, w)
    i = 0
    for level in range(L - 1, -1, -1):
        for k in range(nrb + 1):
            layers = s["output"][i]
            i += 1
            if level and k == nrb:
                walk(layers[:-1], h, w)
                h, w = h * 2, w * 2
                walk(layers[-1:], h, w)
            else:
                walk(layers, h, w)
    total += 2 * 9 * s["final_ch"] * net.out_channels * H * W
    return total / 1e9


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--no-graph", action="store_true")
    a = ap.parse_args()
    torch.set_num_threads(min(torch.get_num_threads(), 16))
    from maua_amd.diffusion import create_models
    t0 = time.perf_counter()
    model, diffusion, _ = create_models("uncondImageNet256", f"ddim{a.steps}", allow_random_init=True,
                                        generator=torch.Generator().manual_seed(0))
    model._handle()
    torch.cuda.synchronize()
    t_init = time.perf_counter() - t0
    B, S = a.batch, a.size
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, 3, S, S, generator=g).cuda()
    t = torch.full((B,), 500.0).cuda()
    out = torch.empty((B, 6, S, S), device="cuda")
    for _ in range(2):
        model(x, t, out=out)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(5):
        model(x, t, out=out)
    torch.cuda.synchronize()
    fwd_ms = (time.perf_counter() - t1) / 5 * 1e3
    xs = x.clone()
    diffusion.ddim_sample_loop(model, xs, use_graph=not a.no_graph)   # capture + first replay
    torch.cuda.synchronize()
    times = []
    for _ in range(a.reps):
        xs.copy_(x)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        _, pred = diffusion.ddim_sample_loop(model, xs, use_graph=not a.no_graph)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t2)
    sec = min(times)
    gf = unet_gflop(model, S, S)
    res = {"workload": f"configs[3]: guided-diffusion UNet (256 ch, attention at 32/16/8, learn_sigma) {S}x{S}, {a.steps}-step DDIM, "
                       f"random init, bf16, batch {B}",
           "batch": B, "steps": a.steps, "seconds_per_batch": sec, "samples_per_s": B / sec,
           "step_ms_in_loop": sec / a.steps * 1e3, "forward_ms_eager": fwd_ms, "graph": model.graph_active(),
           "gflop_per_forward_per_sample": gf, "tflops": gf * B * a.steps / sec / 1e3, "frac_mfma_bf16": gf * B * a.steps / sec / 1e3 / 2500.0,
           "init_s": t_init, "finite": bool(torch.isfinite(pred).all())}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
