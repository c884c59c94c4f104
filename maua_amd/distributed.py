"""Frame-parallel multi-GPU plumbing: one process per GPU, ``torch.distributed`` (backend "nccl" == RCCL over xGMI on
ROCm; "gloo" on CPU for the tests).  Frames are independent after the clip-global pre-pass, so the render is sharded
by contiguous frame range with NO data-path collective; the only exchange is one gather of the packed u8 frames to
rank 0 at the end (precedent for the process layout: reference maua/super/image/bulk.py:31-109)."""
import os

import torch
import torch.distributed as dist

from .pipeline import frame_range


def world_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def maybe_init_process_group(backend=None):
    """Initialise from the torchrun environment (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*) if present."""
    if "RANK" not in os.environ or int(os.environ.get("WORLD_SIZE", "1")) <= 1 or dist.is_initialized():
        return world_info()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        dist.init_process_group(backend, device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend)
    return world_info()


def gather_frames(local, n_frames, rank=None, world=None, dst=0):
    """Gather per-rank frame shards [n_r, ...] (contiguous ranges from ``frame_range``) into [n_frames, ...] on
    ``dst``; other ranks get None.  Shards may differ by one frame: they are padded to the largest shard so that a
    single gather call moves them, then trimmed."""
    if rank is None or world is None:
        rank, world = world_info()
    if world == 1:
        return local
    sizes = [frame_range(n_frames, r, world) for r in range(world)]
    maxn = max(hi - lo for lo, hi in sizes)
    pad = local
    if local.shape[0] < maxn:
        pad = torch.zeros((maxn, *local.shape[1:]), dtype=local.dtype, device=local.device)
        pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad.contiguous(), bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([bufs[r][: hi - lo] for r, (lo, hi) in enumerate(sizes)])
