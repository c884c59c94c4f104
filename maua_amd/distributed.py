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


_comm = {}


def _want_cabi(transport=None):
    """Which transport carries device frames between ranks: "cabi" (default: the library's own RCCL communicator,
    ``maua_gather_frames*``) or "torch" (torch.distributed point-to-point calls on torch's RCCL group) - argument, else the
    MAUA_GATHER environment variable (``bench.py --gather torch`` sets it)."""
    t = (transport or os.environ.get("MAUA_GATHER", "cabi")).lower()
    if t not in ("cabi", "torch"):
        raise ValueError(f"gather transport must be 'cabi' or 'torch', not {t!r}")
    return t == "cabi"


def _cabi_comm(rank, world, device):
    """RCCL communicator of the C-ABI (maua_comm_*), one per process - COLLECTIVE, and it returns the same answer on every
    rank: the communicator, or ``(None, reason)`` everywhere if ANY rank could not build its part (then the callers take
    torch.distributed's point-to-point path together).  ``ncclCommInitRank`` inside ``maua_comm_init`` is itself a collective,
    so nobody enters it before everybody can: every rank executes, in this order and whatever fails where,
      (1) a MIN all-reduce of a PRE-FLIGHT flag - library loaded, context created, RCCL resolved (``maua_comm_unique_id`` on a
          throw-away buffer: it is what dlopens RCCL) - a rank that cannot get that far takes everyone to the fallback here;
      (2) a broadcast of [status byte | 128-byte id] from rank 0 (status 0 if drawing the real id failed);
      (3) ``maua_comm_init``, entered by all or by none;
      (4) a MIN all-reduce of "my maua_comm_init succeeded".
    The outcome - also a negative one - is cached per (rank, world, device): a failed build is not retried (and not warned
    about) at every later gather; ``reset_cabi_comm()`` forgets it.  What this cannot cover: a rank that dies INSIDE
    ncclCommInitRank leaves its peers to RCCL's own time-out."""
    import ctypes as C
    from . import _lib as L
    key = (rank, world, torch.device(device).index)
    if key in _comm:
        return _comm[key]
    cdev = device if (world > 1 and dist.get_backend() == "nccl") else "cpu"
    err, lib, ctx = None, None, None
    try:   # (1) pre-flight: everything maua_comm_init needs that can fail on ONE rank only
        lib = L.lib()
        ctx = L.ctx(device)
        L.check(lib.maua_comm_unique_id((C.c_char * 128)()))
    except Exception as e:   # noqa: BLE001 - reported after the ranks have agreed
        err = e
    if world > 1:
        flag = torch.tensor([0 if err is not None else 1], device=cdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            _comm[key] = (None, err or RuntimeError("a peer rank could not load the library / RCCL or create its context"))
            return _comm[key]
    elif err is not None:
        _comm[key] = (None, err)
        return _comm[key]
    idbuf = (C.c_char * 128)()
    status = 1
    if rank == 0:
        try:
            L.check(lib.maua_comm_unique_id(idbuf))
        except Exception as e:   # noqa: BLE001
            status, err = 0, e
    t = torch.tensor([status] + list(bytes(idbuf.raw)), dtype=torch.uint8)
    if world > 1:   # (2)
        t = t.to(cdev)
        dist.broadcast(t, src=0)
        t = t.cpu()
    status = int(t[0])
    comm = None
    if status:   # (3) every rank passed the pre-flight and holds the id: all enter
        idbuf.raw = bytes(t[1:].tolist())
        c = C.c_void_p()
        try:
            L.check(lib.maua_comm_init(ctx, idbuf, rank, world, C.byref(c)))
            comm = c
        except Exception as e:   # noqa: BLE001
            err = e
    elif err is None:
        err = RuntimeError("rank 0 could not create the RCCL unique id")
    if world > 1:   # (4)
        flag = torch.tensor([0 if comm is None else 1], device=cdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            if comm is not None:
                lib.maua_comm_destroy(comm)
                comm = None
            if err is None:
                err = RuntimeError("a peer rank could not build its RCCL communicator")
    _comm[key] = (comm, None) if comm is not None else (None, err)
    return _comm[key]


def reset_cabi_comm():
    """Forget cached communicator outcomes (tests; after the environment that made a build fail has changed)."""
    _comm.clear()


def gather_frames_cabi(local, n_frames, rank=None, world=None, dst=0):
    """The gather through the library's own entry point (``maua_gather_frames``: grouped RCCL send / recv on the render
    stream, exact shard sizes, no padding); ``local`` is this rank's packed u8 shard on the device.  Collective; if any
    rank has no communicator every rank falls back to ``_gather_p2p`` (same result, torch.distributed transport)."""
    import ctypes as C
    import warnings
    from . import _lib as L
    if rank is None or world is None:
        rank, world = world_info()
    sizes = [frame_range(n_frames, r, world) for r in range(world)]
    # bytes per frame from the trailing dimensions: known on every rank, also on one whose shard is empty (n_frames < world) -
    # no rank-dependent collective here
    per = local.element_size()
    for d in local.shape[1:]:
        per *= int(d)
    local = local.contiguous()
    assert local.shape[0] == sizes[rank][1] - sizes[rank][0], "shard does not match frame_range"
    comm, err = _cabi_comm(rank, world, local.device)
    if comm is None:
        warnings.warn(f"gather_frames: the library's RCCL communicator is unavailable ({err}); using torch.distributed send / recv")
        return _gather_p2p(local, n_frames, rank, world, dst)
    nbytes = (C.c_long * world)(*[(hi - lo) * per for lo, hi in sizes])
    out = torch.empty((n_frames, *local.shape[1:]), dtype=local.dtype, device=local.device) if rank == dst else None
    L.check(L.lib().maua_gather_frames(comm, L.ptr(local.view(torch.uint8).reshape(-1)) if local.numel() else None, nbytes,
                                       L.ptr(out.view(torch.uint8).reshape(-1)) if out is not None else None, dst))
    return out


def _gather_p2p(local, n_frames, rank, world, dst):
    """Exact-size point-to-point gather through torch.distributed (any backend; the CPU / gloo form of
    maua_gather_frames): every rank sends its shard once, the root receives each straight into its slice of the clip."""
    sizes = [frame_range(n_frames, r, world) for r in range(world)]
    local = local.contiguous()
    if rank != dst:
        if local.shape[0]:
            dist.send(local, dst=dst)
        return None
    shape = list(local.shape[1:])
    if not local.shape[0]:   # (the root always owns frames under frame_range; kept for dst != 0)
        raise ValueError("the gather root must own at least one frame")
    out = torch.empty((n_frames, *shape), dtype=local.dtype, device=local.device)
    reqs = []
    for r, (lo, hi) in enumerate(sizes):
        if r == rank:
            out[lo:hi] = local
        elif hi > lo:
            reqs.append(dist.irecv(out[lo:hi], src=r))
    for q in reqs:
        q.wait()
    return out


def gather_frames(local, n_frames, rank=None, world=None, dst=0, transport=None):
    """Gather per-rank frame shards [n_r, ...] (contiguous ranges from ``frame_range``) into [n_frames, ...] on ``dst``;
    other ranks get None.  Exact shard sizes, one transfer per rank, received in place (no padding, no concatenation):
    on the device through the library's own ``maua_gather_frames`` (grouped RCCL send / recv over xGMI; ``transport="torch"``
    or MAUA_GATHER=torch: torch.distributed's point-to-point calls instead, also the automatic fallback when the library's
    communicator cannot be built on some rank), on the CPU (gloo: the tests) through torch.distributed point-to-point calls."""
    if rank is None or world is None:
        rank, world = world_info()
    if world == 1:
        return local
    if local.is_cuda and _want_cabi(transport):
        return gather_frames_cabi(local, n_frames, rank, world, dst)
    return _gather_p2p(local, n_frames, rank, world, dst)


def clip_batch(n_local, preferred):
    """Frames per synthesis call for a rank that renders ``n_local`` frames of a clip: a batch that DIVIDES the shard (no ragged
    last call - at 8 GPUs a 450-frame shard in batches of 128 ends in a 66-frame call that costs most of a full one), as close to
    ``preferred`` as the divisors allow: the largest divisor in [preferred / 2, 1.25 * preferred], else ``preferred`` itself (the
    tail then stays).  3600 -> 150 (24 calls), 1800 -> 150, 900 -> 150, 450 -> 150 (3 calls) for preferred = 128."""
    n_local, preferred = int(n_local), int(preferred)
    if n_local <= 0 or preferred <= 0:
        return max(1, preferred)
    if n_local <= preferred * 5 // 4:
        return n_local
    cands = [d for d in range(max(1, preferred // 2), preferred * 5 // 4 + 1) if n_local % d == 0]
    return max(cands) if cands else preferred


class StreamingGather:
    """The gather of a frame-sharded render, STREAMED: every rank renders its contiguous frame range in chunks; as soon as
    a chunk is finished it travels to the root (side stream, RCCL point-to-point over xGMI) while the next chunk renders,
    so the root's ingress (8 x 1.4 GB at 8 GPUs) hides behind the render instead of following it.

        g = StreamingGather(n_frames, frame_shape, chunk, dtype=torch.uint8)     # collective: all ranks
        for off, n in g.chunks():                # this rank's chunks: (offset inside the shard, frames)
            render into g.local[off:off + n]
            g.chunk_done()
        clip = g.finish()                        # root: [n_frames, *frame_shape]; other ranks: None

    The root renders straight into its slice of the clip buffer (``g.local`` is a view of it).  Round k = the k-th chunk
    of every rank; the root posts the round's receives (one group) when its own k-th chunk is done, the other ranks
    send theirs - rank 0 owns the longest range (``frame_range``), so it has a chunk in every round.  CPU tensors
    (gloo) follow the same protocol with torch.distributed isend / irecv (the tests)."""

    def __init__(self, n_frames, frame_shape, chunk, dtype=torch.uint8, device=None, rank=None, world=None, dst=0,
                 transport=None):
        if rank is None or world is None:
            rank, world = world_info()
        if dst != 0:
            raise NotImplementedError("the streamed gather collects on rank 0 (the rank with the longest frame range)")
        self.rank, self.world, self.dst, self.n_frames, self.chunk = rank, world, dst, n_frames, int(chunk)
        self.ranges = [frame_range(n_frames, r, world) for r in range(world)]
        lo, hi = self.ranges[rank]
        self.lo, self.hi = lo, hi
        if device is None:
            device = "cuda" if torch.cuda.is_available() else "cpu"
        self.device = torch.device(device)
        if rank == dst:
            self.clip = torch.empty((n_frames, *frame_shape), dtype=dtype, device=self.device)
            self.local = self.clip[lo:hi]
        else:
            self.clip = None
            self.local = torch.empty((hi - lo, *frame_shape), dtype=dtype, device=self.device)
        self.frame_bytes = int(torch.empty((1, *frame_shape), dtype=dtype).numel()) * torch.empty((), dtype=dtype).element_size()
        self.round = 0
        self._reqs = []
        self._side = None
        if self.device.type == "cuda" and world > 1:
            import ctypes as C
            from . import _lib as L
            self._side = torch.cuda.Stream(device=self.device)
            # every rank takes the same transport: the library's RCCL rounds, or - if any rank has no communicator (or the
            # caller asked for it) - torch.distributed's own point-to-point calls on the same side stream.  _cabi_comm is
            # collective and answers identically everywhere, so no rank can be left alone in a collective.
            self._comm, err = (None, None)
            if _want_cabi(transport):
                self._comm, err = _cabi_comm(rank, world, self.device)
                if self._comm is None:
                    if dist.get_backend() != "nccl":
                        raise err
                    import warnings
                    warnings.warn(f"StreamingGather: the library's RCCL communicator is unavailable ({err}); using torch.distributed isend / irecv")
            if self._comm is not None:
                # this communicator's transfers run on the side stream from here on (finish() hands it back)
                L.check(L.lib().maua_comm_set_stream(self._comm, C.c_void_p(self._side.cuda_stream), 0))
        # the exchange's rank count as the TRANSPORT reports it (read back, not echoed): RCCL's ncclCommCount for the library's
        # communicator, torch.distributed's group size otherwise
        self.nranks = 1
        if world > 1:
            self.nranks = dist.get_world_size()
            if getattr(self, "_comm", None) is not None:
                import ctypes as C
                from . import _lib as L
                nr = C.c_int(0)
                L.check(L.lib().maua_comm_count(self._comm, C.byref(nr), None))
                self.nranks = int(nr.value)
        self.transport = ("none (one rank)" if world == 1 else "torch.distributed isend / irecv" if self._side is None or
                          getattr(self, "_comm", None) is None else "maua_gather_frames_at (RCCL point-to-point, C ABI)")

    def n_rounds(self, r=None):
        lo, hi = self.ranges[self.rank if r is None else r]
        return -(-(hi - lo) // self.chunk)

    def chunks(self):
        for off in range(0, self.hi - self.lo, self.chunk):
            yield off, min(self.chunk, self.hi - self.lo - off)

    def _round_pieces(self, k):
        """(frames, clip frame offset) of every rank's k-th chunk (0 frames: the rank has no such chunk)."""
        out = []
        for lo, hi in self.ranges:
            off = k * self.chunk
            n = max(0, min(self.chunk, hi - lo - off))
            out.append((n, lo + off))
        return out

    def chunk_done(self):
        """This rank's next chunk (in ``chunks()`` order) is rendered (its kernels are queued on the current stream)."""
        k = self.round
        self.round += 1
        if self.world == 1:
            return
        pieces = self._round_pieces(k)
        if self._side is not None and self._comm is not None:
            import ctypes as C
            from . import _lib as L
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            nbytes = (C.c_long * self.world)(*[n * self.frame_bytes for n, _ in pieces])
            offs = (C.c_long * self.world)(*[o * self.frame_bytes for _, o in pieces])
            n_mine, _ = pieces[self.rank]
            off = k * self.chunk
            with torch.cuda.stream(self._side):
                self._side.wait_event(ev)
                send = self.local[off:off + n_mine].view(torch.uint8).reshape(-1) if self.rank != self.dst and n_mine else None
                recv = self.clip.view(torch.uint8).reshape(-1) if self.rank == self.dst else None
                L.check(L.lib().maua_gather_frames_at(self._comm, L.ptr(send), nbytes, L.ptr(recv), offs, self.dst))
            return
        # torch.distributed point-to-point: CPU tensors over gloo (the tests), or device tensors over torch's own RCCL group
        # when the library's communicator is unavailable (then on the side stream, behind the render's event)
        def post():
            if self.rank == self.dst:
                for r, (n, o) in enumerate(pieces):
                    if r != self.rank and n:
                        self._reqs.append(dist.irecv(self.clip[o:o + n], src=r))
            else:
                n, _ = pieces[self.rank]
                if n:
                    off = k * self.chunk
                    self._reqs.append(dist.isend(self.local[off:off + n].contiguous(), dst=self.dst))
        if self._side is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self._side):
                self._side.wait_event(ev)
                post()
        else:
            post()

    def finish(self):
        """Every chunk announced: wait for the transfers; the root returns the clip."""
        if self.round != self.n_rounds():
            raise RuntimeError(f"{self.round} of {self.n_rounds()} chunks were announced")
        if self.world > 1 and self.rank == self.dst:
            # rounds this rank has no chunk in cannot exist (rank 0 owns the longest range)
            assert all(self.n_rounds(r) <= self.n_rounds() for r in range(self.world))
        if self._side is not None and self._comm is None:
            with torch.cuda.stream(self._side):
                for q in self._reqs:
                    q.wait()
        else:
            for q in self._reqs:
                q.wait()
        self._reqs = []
        if self._side is not None:
            from . import _lib as L
            torch.cuda.current_stream(self.device).wait_stream(self._side)
            if self._comm is not None:
                L.check(L.lib().maua_comm_set_stream(self._comm, None, 1))   # the communicator is shared with gather_frames_cabi
        return self.clip if self.rank == self.dst else None


# ---------------------------------------------------------------------------------------------- per-rank part files
def part_path(stem, rank):
    return f"{stem}_part{rank:03d}.mp4"


def write_parts_and_join(stem, n_frames, rank, world, write_part, audio_file=None, audio_offset=0, audio_duration=None,
                         on_rank0=None, run=None, which=None):
    """configs[4]'s multi-rank writer (a 4096^2 frame is 48 MiB: no gather): every rank with a non-empty ``frame_range`` writes
    ``<stem>_partRRR.mp4`` through ``write_part(path, lo, hi) -> frames written``; rank 0 then joins the parts in rank order by
    stream copy (ffmpeg concat demuxer) and muxes the clip's audio in.  The reference's only multi-process writer has this
    shape (super/image/bulk.py:31-109: one output per worker, ordered by index).
    * an empty shard (more ranks than frames) writes nothing and is left out of the list;
    * a rank whose ``write_part`` raises still reaches the rendezvous: a MIN all-reduce of "my part is complete" replaces the
      barrier, and EVERY rank raises when any part is missing - nobody hangs, rank 0 never joins a clip with a hole;
    * audio: any file the ``ffmpeg`` executable can read (it decodes mp3 / flac / ... itself), cut with -ss / -t and ``-shortest``
      so that it cannot outlast the frames.
    -> on rank 0: the joined file (or the concat list when no ``ffmpeg`` executable is on PATH); elsewhere None."""
    import shutil
    import subprocess
    from pathlib import Path
    lo, hi = frame_range(n_frames, rank, world)
    err = None
    try:
        if hi > lo:
            n = write_part(part_path(stem, rank), lo, hi)
            if n != hi - lo:
                raise RuntimeError(f"rank {rank} wrote {n} of {hi - lo} frames")
    except Exception as e:   # noqa: BLE001 - re-raised below, after the ranks have met
        err = e
    if world > 1:
        ok = torch.tensor([0 if err is not None else 1])
        if dist.get_backend() == "nccl":
            ok = ok.cuda()
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            raise RuntimeError(f"a rank failed to write its part of {stem}" + (f" (this one: {err})" if err is not None else "")) from err
    elif err is not None:
        raise err
    if rank != 0:
        return None
    parts = [part_path(stem, r) for r in range(world) if frame_range(n_frames, r, world)[1] > frame_range(n_frames, r, world)[0]]
    lst = stem + "_parts.txt"
    Path(lst).write_text("".join(f"file '{Path(p).name}'\n" for p in parts))
    if on_rank0 is not None:
        on_rank0()
    joined = stem + ".mp4"
    which = which or shutil.which
    if which("ffmpeg") and all(Path(p).exists() for p in parts):
        cmd = ["ffmpeg", "-y", "-loglevel", "error", "-f", "concat", "-safe", "0", "-i", lst]
        if audio_file:
            cmd += ["-ss", str(audio_offset)] + (["-t", str(audio_duration)] if audio_duration else []) + ["-i", str(audio_file), "-c:a", "aac",
                                                                                                           "-shortest"]
        (run or subprocess.run)(cmd + ["-c:v", "copy", joined], check=True)
        return joined
    return lst
