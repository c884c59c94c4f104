"""CPU (gloo) tests of the frame-sharding + gather path that bench.py / sample.py use on N GPUs: world 2 and world 8,
uneven and empty shards, the one-shot exact-size gather and the streamed (chunk-by-chunk) gather protocol."""
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


def _frame(f):
    return torch.full((2, 3, 3), (f * 7) % 251, dtype=torch.uint8)


def _worker_stream(rank, world, port, T, chunk, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from maua_amd.distributed import StreamingGather, gather_frames, maybe_init_process_group
    from maua_amd.pipeline import frame_range
    assert maybe_init_process_group("gloo") == (rank, world)
    lo, hi = frame_range(T, rank, world)
    want = torch.stack([_frame(f) for f in range(T)])
    # one-shot exact-size gather (empty shards included)
    local = want[lo:hi].clone()
    out = gather_frames(local, T, rank, world)
    ok = out is None if rank else bool(torch.equal(out, want))
    # streamed: chunks announced as they are "rendered"; the root renders into the clip buffer itself
    g = StreamingGather(T, (2, 3, 3), chunk, dtype=torch.uint8, device="cpu")
    assert (g.lo, g.hi) == (lo, hi) and sum(n for _, n in g.chunks()) == hi - lo
    for off, n in g.chunks():
        g.local[off:off + n] = want[lo + off:lo + off + n]
        g.chunk_done()
    clip = g.finish()
    ok = ok and (clip is None if rank else bool(torch.equal(clip, want)))
    if rank == 0:
        assert clip.data_ptr() == g.local.data_ptr() - lo * 18   # the root's shard is a view of the clip (no copy)
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,T,chunk", [(8, 29, 2), (8, 5, 4), (2, 13, 3), (3, 10, 100)])
def test_streamed_and_one_shot_gather(world, T, chunk):
    """uneven shards (29 frames on 8 ranks: 4 4 4 4 4 3 3 3), empty shards (5 frames on 8 ranks), a chunk larger than a
    shard, several rounds per rank."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_stream, args=(r, world, port, T, chunk, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(res[r] for r in range(world)), res
