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


class _FakeLib:
    """Stand-in for the C-ABI library's communicator entry points: fails where told to (no GPU / RCCL needed)."""

    def __init__(self, fail_id, fail_init):
        self.fail_id, self.fail_init, self.destroyed = fail_id, fail_init, 0

    def maua_comm_unique_id(self, buf):
        if self.fail_id:
            return 1
        buf.raw = bytes(range(128))
        return 0

    def maua_comm_init(self, ctx, idbuf, rank, world, out):
        assert bytes(idbuf.raw) == bytes(range(128))   # every rank received rank 0's id
        return 1 if self.fail_init else 0

    def maua_comm_destroy(self, comm):
        self.destroyed += 1
        return 0


def _worker_agree(rank, world, port, case, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import maua_amd._lib as L
    import maua_amd.distributed as D
    assert D.maybe_init_process_group("gloo") == (rank, world)
    fake = _FakeLib(fail_id=(case == "id" and rank == 0), fail_init=(case == "init" and rank == 1))
    L.lib = lambda: fake
    L.ctx = lambda device=None: None

    def check(rc):
        if rc:
            raise RuntimeError("injected failure")
    L.check = check
    comm, err = D._cabi_comm(rank, world, "cpu")
    # the collectives that follow must line up on every rank (a rank stuck in _cabi_comm would hang this all_reduce)
    t = torch.tensor([1 if comm is None else 0])
    dist.all_reduce(t)
    q.put((rank, comm is None, err is not None, int(t.item()), fake.destroyed))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case", ["ok", "id", "init"])
def test_rccl_communicator_agreement_never_splits_the_ranks(case):
    """rank 0 failing before the id broadcast (RCCL not loadable) or one rank failing in maua_comm_init must end
    with EVERY rank holding no communicator (and falling back together), never with ranks waiting in different collectives."""
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_agree, args=(r, world, port, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    none = [r[1] for r in res]
    assert none == [case != "ok"] * world, res
    assert all(r[3] == (world if case != "ok" else 0) for r in res)
    if case != "ok":
        assert all(r[2] for r in res)                     # every rank can say why
    if case == "init":
        assert [r[4] for r in res] == [1, 0, 1], res       # the ranks that had built theirs released them
