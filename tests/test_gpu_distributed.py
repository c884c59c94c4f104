"""world_size-2 render over RCCL ("nccl" backend on ROCm): each rank renders its contiguous frame range of a small clip
on its own GPU, one gather to rank 0, compared with the single-GPU render.  Needs two visible GPUs; the 1-GPU test box
skips it (the sharding / gather logic itself is covered on CPU by tests/test_distributed_gloo.py)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _render(lo, hi, T, B=4):
    from maua_amd.noise import Loop, loop_batch
    from maua_amd.stylegan2 import SynthesisNetwork
    net = SynthesisNetwork(64, 64, 3, channel_base=2048, channel_max=128, dtype=torch.bfloat16,
                           generator=torch.Generator().manual_seed(0))
    g = torch.Generator().manual_seed(1)
    ws = torch.randn(T, net.num_ws, 64, generator=g).cuda()
    rng = torch.Generator().manual_seed(42)
    mods = [Loop(rng, T, (s[3], s[3]), n_loops=2, sigma=5) for s in net.layer_shapes()]
    out = torch.empty((hi - lo, 64, 64, 3), dtype=torch.uint8, device="cuda")
    for i in range(lo, hi, B):
        b = min(B, hi - i)
        net(ws[i:i + b], noise=loop_batch(mods, i, b), rgb8_out=out[i - lo:i - lo + b])
    return out


def _worker(rank, world, port, T, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0")
    from maua_amd.distributed import gather_frames, maybe_init_process_group
    from maua_amd.pipeline import frame_range
    assert maybe_init_process_group("nccl") == (rank, world)
    lo, hi = frame_range(T, rank, world)
    shard = _render(lo, hi, T)
    full = gather_frames(shard, T, rank, world)           # default on the device: the library's maua_gather_frames
    from maua_amd.distributed import StreamingGather, _gather_p2p
    full2 = _gather_p2p(shard, T, rank, world, 0)         # the same exchange through torch.distributed's RCCL send / recv
    # streamed: chunks of 3 frames travel on a side stream while the next ones render
    g = StreamingGather(T, (64, 64, 3), 3)
    for off, n in g.chunks():
        g.local[off:off + n] = shard[off:off + n]
        g.chunk_done()
    full3 = g.finish()
    torch.cuda.synchronize()
    if rank == 0:
        assert torch.equal(full, full2) and torch.equal(full, full3)
        q.put(full.cpu())
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL gather)")
def test_two_rank_render_equals_single_gpu():
    T = 13  # uneven shards: 7 + 6 frames
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, T, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    torch.cuda.set_device(0)
    want = _render(0, T, T).cpu()
    assert torch.equal(got, want)



def test_cabi_gather_single_rank():
    """maua_comm_* / maua_gather_frames on one GPU (world 1: RCCL is bound and a communicator created, the root's own shard
    is copied into place); the two-rank exchange is the test above wherever two GPUs exist."""
    from maua_amd.distributed import StreamingGather, gather_frames_cabi
    x = torch.randint(0, 255, (5, 8, 8, 3), dtype=torch.uint8, device="cuda")
    out = gather_frames_cabi(x, 5, rank=0, world=1)
    torch.cuda.synchronize()
    assert out.data_ptr() != x.data_ptr() and torch.equal(out, x)
    # the streamed form at world 1: the clip buffer IS the shard, chunks are only counted
    g = StreamingGather(5, (8, 8, 3), 2, rank=0, world=1)
    for off, n in g.chunks():
        g.local[off:off + n] = x[off:off + n]
        g.chunk_done()
    clip = g.finish()
    assert torch.equal(clip, x) and clip.data_ptr() == g.local.data_ptr()
    with pytest.raises(RuntimeError):
        StreamingGather(5, (8, 8, 3), 2, rank=0, world=1).finish()   # chunks not announced


def test_communicator_runs_on_its_own_stream():
    """maua_comm_set_stream: the communicator's transfers are enqueued on the stream it was given (the streamed gather's side
    stream), not on whatever stream the shared context was last bound to - checked at world 1, where the gather is the root's
    own copy: a copy held back by an event on the side stream must not complete before the event is released."""
    import ctypes as C
    from maua_amd import _lib as L
    from maua_amd.distributed import _cabi_comm
    x = torch.randint(0, 255, (4, 16, 16, 3), dtype=torch.uint8, device="cuda")
    out = torch.zeros_like(x)
    comm, _ = _cabi_comm(0, 1, x.device)
    side = torch.cuda.Stream()
    L.check(L.lib().maua_comm_set_stream(comm, C.c_void_p(side.cuda_stream), 0))
    nbytes = (C.c_long * 1)(x.numel())
    # work queued on the side stream ahead of the gather: the gather must come after it in stream order
    big = torch.empty((1 << 28,), dtype=torch.uint8, device="cuda")
    with torch.cuda.stream(side):
        for _ in range(8):
            big.fill_(1)
        marker = torch.cuda.Event()
        marker.record(side)
    L.ctx(x.device)                                   # the shared context is (re-)bound to the CURRENT stream, not the side stream
    L.check(L.lib().maua_gather_frames(comm, L.ptr(x.reshape(-1)), nbytes, L.ptr(out.reshape(-1)), 0))
    done = torch.cuda.Event()
    done.record(side)
    done.synchronize()
    assert marker.query() and torch.equal(out, x)
    L.check(L.lib().maua_comm_set_stream(comm, None, 1))
    out.zero_()
    L.check(L.lib().maua_gather_frames(comm, L.ptr(x.reshape(-1)), nbytes, L.ptr(out.reshape(-1)), 0))
    torch.cuda.current_stream().synchronize()
    assert torch.equal(out, x)
