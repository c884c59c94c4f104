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
        self.fail_id, self.fail_init, self.destroyed, self.inits = fail_id, fail_init, 0, 0

    def maua_comm_unique_id(self, buf):
        if self.fail_id:
            return 1
        buf.raw = bytes(range(128))
        return 0

    def maua_comm_init(self, ctx, idbuf, rank, world, out):
        assert bytes(idbuf.raw) == bytes(range(128))   # every rank received rank 0's id
        self.inits += 1
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

    def ctx(device=None):
        if case == "ctx" and rank == 2:
            raise RuntimeError("injected: no context on this rank")
    L.ctx = ctx

    def check(rc):
        if rc:
            raise RuntimeError("injected failure")
    L.check = check
    comm, err = D._cabi_comm(rank, world, "cpu")
    # the collectives that follow must line up on every rank (a rank stuck in _cabi_comm would hang this all_reduce)
    t = torch.tensor([1 if comm is None else 0])
    dist.all_reduce(t)
    # (a second call answers from the cache - no collective, no second attempt)
    again = D._cabi_comm(rank, world, "cpu")
    assert (again[0] is None) == (comm is None)
    q.put((rank, comm is None, err is not None, int(t.item()), fake.destroyed, fake.inits))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case", ["ok", "id", "init", "ctx"])
def test_rccl_communicator_agreement_never_splits_the_ranks(case):
    """rank 0 failing before the id broadcast (RCCL not loadable) or one rank failing in maua_comm_init must end
    with EVERY rank holding no communicator (and falling back together), never with ranks waiting in different collectives.
    ADVICE r4: ncclCommInitRank is itself a collective - a rank that cannot even create its context ("ctx") must keep its peers
    OUT of maua_comm_init (pre-flight agreement), and a failed build is cached, not retried at every gather."""
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
    assert [r[5] for r in res] == ([1] * world if case in ("ok", "init") else [0] * world), res   # init entered by all or by none


def _worker_parts(rank, world, port, T, tmp, fail_rank, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import maua_amd.distributed as D
    from maua_amd.pipeline import frame_range
    assert D.maybe_init_process_group("gloo") == (rank, world)
    stem = os.path.join(tmp, "clip_x4plus_4096x4096")
    calls = []

    def write_part(path, lo, hi):   # the stub renderer: one line per frame, tagged with the rank that wrote it
        if rank == fail_rank:
            raise RuntimeError("injected: the renderer died on this rank")
        with open(path, "w") as f:
            for i in range(lo, hi):
                f.write(f"{i} {rank}\n")
        calls.append((path, lo, hi))
        return hi - lo
    ran = []

    def run(cmd, check):   # stands in for subprocess.run(ffmpeg ...): concatenates what the list names, in its order
        ran.append(cmd)
        lst = cmd[cmd.index("-i") + 1]
        names = [ln.split("'")[1] for ln in open(lst).read().splitlines()]
        with open(cmd[-1], "w") as out:
            for n in names:
                out.write(open(os.path.join(os.path.dirname(lst), n)).read())
    saved = []
    try:
        out = D.write_parts_and_join(stem, T, rank, world, write_part, audio_file="clip.mp3", audio_offset=1.5, audio_duration=4,
                                     on_rank0=lambda: saved.append(rank), run=run, which=lambda name: "/usr/bin/" + name)
        err = None
    except RuntimeError as e:
        out, err = None, str(e)
    # the collectives that follow line up on every rank, failure or not
    t = torch.tensor([1])
    dist.all_reduce(t)
    lo, hi = frame_range(T, rank, world)
    q.put((rank, out, err, calls, saved, ran, int(t.item()), (lo, hi)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,T,fail_rank", [(2, 7, -1), (3, 2, -1), (3, 10, 1)], ids=["two-ranks", "more-ranks-than-frames", "a-rank-fails"])
def test_upscaled_render_part_files_are_joined_in_rank_order(tmp_path, world, T, fail_rank):
    """configs[4]'s multi-rank half (the reference's only multi-process writer pattern is
    super/image/bulk.py:31-109): every rank writes ``<stem>_partRRR.mp4`` for its contiguous frame range through a stub renderer,
    rank 0 alone writes the ordered concat list, the side file and the join command (stream copy, any ffmpeg-readable audio,
    -shortest).  Empty shards write no file and are not listed; a rank whose renderer dies makes EVERY rank raise, none hangs."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_parts, args=(r, world, port, T, str(tmp_path), fail_rank, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {r[0]: r for r in (q.get(timeout=120) for _ in range(world))}
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(res[r][6] == world for r in range(world))
    stem = str(tmp_path / "clip_x4plus_4096x4096")
    if fail_rank >= 0:
        assert all(res[r][1] is None and res[r][2] for r in range(world)), res          # everybody raised
        assert "this one" in res[fail_rank][2] and not res[0][4] and not res[0][5]        # rank 0 joined nothing
        return
    nonempty = [r for r in range(world) if res[r][7][1] > res[r][7][0]]
    for r in range(world):
        lo, hi = res[r][7]
        assert res[r][3] == ([(f"{stem}_part{r:03d}.mp4", lo, hi)] if hi > lo else []), res[r]
        assert os.path.exists(f"{stem}_part{r:03d}.mp4") == (hi > lo)
        assert res[r][1] == (stem + ".mp4" if r == 0 else None) and res[r][4] == ([0] if r == 0 else [])
    listed = [ln.split("'")[1] for ln in open(stem + "_parts.txt").read().splitlines()]
    assert listed == [f"clip_x4plus_4096x4096_part{r:03d}.mp4" for r in nonempty]
    joined = [ln.split() for ln in open(stem + ".mp4").read().splitlines()]
    assert [int(a) for a, _ in joined] == list(range(T))                                  # every frame once, in order
    from maua_amd.pipeline import frame_range
    assert all(frame_range(T, int(rk), world)[0] <= int(i) < frame_range(T, int(rk), world)[1] for i, rk in joined)
    (cmd,) = res[0][5]
    assert cmd[:3] == ["ffmpeg", "-y", "-loglevel"] and cmd[cmd.index("-f") + 1] == "concat" and cmd[-3:-1] == ["-c:v", "copy"]
    assert cmd[cmd.index("-ss") + 1] == "1.5" and cmd[cmd.index("-t") + 1] == "4" and "clip.mp3" in cmd and "-shortest" in cmd


def test_clip_leg_batch_divides_the_shard():
    """bench.py's clip leg: frames per synthesis call = distributed.clip_batch(longest shard, the timed steps'
    batch) - a divisor of the shard near the preferred batch, so that no rank ends its range with a ragged call (450 frames at 8
    GPUs: 3 x 150 instead of 128 + 128 + 128 + 66); shards without such a divisor keep the preferred batch."""
    from maua_amd.distributed import clip_batch
    from maua_amd.pipeline import frame_range
    for world in (1, 2, 4, 8):
        lo, hi = frame_range(3600, 0, world)
        cb = clip_batch(hi - lo, 128)
        assert cb == 150 and (hi - lo) % cb == 0, (world, cb)
    assert clip_batch(451, 128) == 128 and clip_batch(449, 128) == 128      # (primes / near-primes: no divisor in [64, 160])
    assert clip_batch(100, 128) == 100 and clip_batch(160, 128) == 160 and clip_batch(161, 128) == 128
    assert clip_batch(3600, 32) == 40 and clip_batch(0, 128) == 128 and clip_batch(7, 128) == 7
    # every rank of a world takes the SAME batch (the streamed gather's rounds are equal chunks): it comes from rank 0's range, the longest
    for world in (3, 7):
        longest = max(frame_range(3600, r, world)[1] - frame_range(3600, r, world)[0] for r in range(world))
        assert longest == frame_range(3600, 0, world)[1] - frame_range(3600, 0, world)[0]
