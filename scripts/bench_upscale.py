"""configs[4] leg: 1024^2 StyleGAN2 frame -> RealESRGAN x4 (23 RRDB blocks, random init) -> 4096^2 u8, per frame, frames sharded
by contiguous range over the ranks (one process per GPU: python -m torch.distributed.run --nproc-per-node N scripts/bench_upscale.py).
Prints one JSON line on rank 0: whole-job frames/s, ms per frame, algorithmic TFLOP/s of the up-scaler."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, ".")
from maua_amd.stylegan2 import SynthesisNetwork  # noqa: E402
from maua_amd.super import RRDBNet  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
res = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
torch.cuda.set_device(local)
dist = None
if world > 1:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
G = SynthesisNetwork(512, res, 3, dtype=torch.bfloat16, generator=torch.Generator().manual_seed(0))
S = RRDBNet(num_block=23, dtype=torch.bfloat16)
ws = torch.randn(B, G.num_ws, 512, generator=torch.Generator().manual_seed(1 + rank)).cuda()
img = torch.empty((B, 3, res, res), device="cuda")
u8 = torch.empty((B, 4 * res, 4 * res, 3), dtype=torch.uint8, device="cuda")


def step():
    G(ws, out=img)
    x = img.add(1).div(2).clamp_(0, 1)
    S(x, rgb8_out=u8)


def fence():
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()


step()
fence()
t0 = time.perf_counter()
for _ in range(steps):
    step()
fence()
dt = torch.tensor([(time.perf_counter() - t0) / steps], device="cuda")
if dist is not None:
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
dt = float(dt)
# RRDBNet(64, 23 blocks, grow 32) MACs per input pixel: dense blocks + conv_first/body + the two up convs (x4, x16 pixels) + hr / last
f, g = 64, 32
rdb = 9 * sum((f + k * g) * (f if k == 4 else g) for k in range(5))
macs_px = 23 * 3 * rdb + 9 * (3 * f + f * f) + 9 * f * f * 4 + 9 * f * f * 16 + 9 * f * f * 16 + 9 * f * 3 * 16
tflop = 2 * macs_px * res * res * B / 1e12
if rank == 0:
    print(json.dumps({"metric": "frames/sec (whole job), 1024^2 StyleGAN2 render -> RealESRGAN x4 -> 4096^2 u8", "value": world * B / dt,
                      "unit": "frames/s", "n_gpus": world, "ms_per_frame_per_gpu": dt * 1e3 / B, "dtype": "bf16", "data": "synthetic",
                      "upscaler_tflop_per_frame": tflop / B, "upscaler_tflops_per_gpu": tflop / dt,
                      "config": {"workload": "configs[4]", "render_res": res, "rrdb_blocks": 23, "frames_per_step_per_gpu": B}}))
    print(f"render + x4 upscale, B={B}, {res}^2 -> {4*res}^2: {dt*1e3:.1f} ms/step = {world * B/dt:.2f} frames/s")
if dist is not None:
    dist.destroy_process_group()
