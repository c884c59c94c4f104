"""world_size-2 CPU (gloo) test of the frame-sharding + gather path that bench.py / sample.py use on N GPUs."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, T, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from maua_amd.distributed import gather_frames, maybe_init_process_group, world_info
    from maua_amd.pipeline import frame_range
    assert maybe_init_process_group("gloo") == (rank, world)
    assert world_info() == (rank, world)
    lo, hi = frame_range(T, rank, world)
    # each "frame" f is a tiny u8 image whose bytes encode f: the gathered clip must be in frame order
    local = torch.stack([torch.full((2, 3, 3), (f * 7) % 251, dtype=torch.uint8) for f in range(lo, hi)]) \
        if hi > lo else torch.zeros((0, 2, 3, 3), dtype=torch.uint8)
    out = gather_frames(local, T, rank, world)
    if rank == 0:
        want = torch.stack([torch.full((2, 3, 3), (f * 7) % 251, dtype=torch.uint8) for f in range(T)])
        q.put(bool(torch.equal(out, want)))
    else:
        assert out is None
    # max-over-ranks timing reduction used by bench.py
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.item() == world
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("T", [9, 16])
def test_shard_and_gather_world2(T):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, T, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True
