// The one exchange step of the frame-sharded render (BASELINE north_star: "a single RCCL gather over
// xGMI at the end"): every rank's packed u8 shard -> root, ordered by rank, as grouped point-to-point transfers (a gather
// is exactly one send per rank on xGMI's point-to-point links - no ring, no reduction).  Replaces the reference's
// single-process frame list (maua/audiovisual/generate.py:57-98 renders every frame on one device; the process layout
// follows maua/super/image/bulk.py:31-109).
//
// RCCL is bound at first use (dlopen): libmaua_hip.so has no link-time dependency on it, single-GPU hosts never load
// it, and inside a PyTorch process the already-loaded RCCL is the one that is found.
#include <dlfcn.h>

#include <vector>

#include "common.h"

namespace {

typedef int (*fn_get_unique_id)(void*);
typedef int (*fn_comm_init_rank)(void**, int, maua_comm_id, int);
typedef int (*fn_comm_destroy)(void*);
typedef int (*fn_send)(const void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_recv)(void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_group)(void);
typedef const char* (*fn_err)(int);
typedef int (*fn_count)(void*, int*);

struct Rccl {
  void* lib = nullptr;
  fn_get_unique_id get_unique_id = nullptr;
  fn_comm_init_rank comm_init_rank = nullptr;
  fn_comm_destroy comm_destroy = nullptr;
  fn_send send = nullptr;
  fn_recv recv = nullptr;
  fn_group group_start = nullptr, group_end = nullptr;
  fn_err error_string = nullptr;
  fn_count comm_count = nullptr, comm_user_rank = nullptr;   // optional (ncclCommCount / ncclCommUserRank)
};

Rccl g_rccl;

int bind_rccl() {
  if (g_rccl.lib) return MAUA_OK;
  // a process that already carries RCCL (PyTorch) resolves to that copy; otherwise the system's
  void* lib = dlopen(nullptr, RTLD_NOW);
  if (!lib || !dlsym(lib, "ncclCommInitRank")) {
    lib = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
      if ((lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
  }
  if (!lib) return maua::fail("maua_comm: RCCL (librccl.so) not found");
  Rccl r;
  r.lib = lib;
  r.get_unique_id = (fn_get_unique_id)dlsym(lib, "ncclGetUniqueId");
  r.comm_init_rank = (fn_comm_init_rank)dlsym(lib, "ncclCommInitRank");
  r.comm_destroy = (fn_comm_destroy)dlsym(lib, "ncclCommDestroy");
  r.send = (fn_send)dlsym(lib, "ncclSend");
  r.recv = (fn_recv)dlsym(lib, "ncclRecv");
  r.group_start = (fn_group)dlsym(lib, "ncclGroupStart");
  r.group_end = (fn_group)dlsym(lib, "ncclGroupEnd");
  r.error_string = (fn_err)dlsym(lib, "ncclGetErrorString");
  r.comm_count = (fn_count)dlsym(lib, "ncclCommCount");
  r.comm_user_rank = (fn_count)dlsym(lib, "ncclCommUserRank");
  if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.send || !r.recv || !r.group_start || !r.group_end)
    return maua::fail("maua_comm: librccl lacks the point-to-point API");
  g_rccl = r;
  return MAUA_OK;
}

int rccl_fail(const char* what, int rc) {
  return maua::fail(std::string(what) + ": " + (g_rccl.error_string ? g_rccl.error_string(rc) : "RCCL error") + " (" +
                    std::to_string(rc) + ")");
}

}  // namespace

struct maua_comm {
  maua_ctx* ctx;
  void* comm;
  int rank, world;
  bool own_stream = false;   // maua_comm_set_stream: the transfers run on `stream` instead of the context's
  hipStream_t stream = nullptr;
};

static hipStream_t comm_stream(const maua_comm* c) { return c->own_stream ? c->stream : c->ctx->stream; }

extern "C" {

int maua_comm_unique_id(maua_comm_id* id) {
  MAUA_REQUIRE(id, "maua_comm_unique_id: NULL argument");
  if (int rc = bind_rccl()) return rc;
  if (int rc = g_rccl.get_unique_id(id)) return rccl_fail("ncclGetUniqueId", rc);
  return MAUA_OK;
}

int maua_comm_init(maua_ctx* ctx, const maua_comm_id* id, int rank, int world, maua_comm** out) {
  MAUA_REQUIRE(ctx && id && out, "maua_comm_init: NULL argument");
  MAUA_REQUIRE(world >= 1 && rank >= 0 && rank < world, "maua_comm_init: rank outside [0, world)");
  if (int rc = bind_rccl()) return rc;
  MAUA_HIP_CHECK(hipSetDevice(ctx->device));
  void* c = nullptr;
  if (int rc = g_rccl.comm_init_rank(&c, world, *id, rank)) return rccl_fail("ncclCommInitRank", rc);
  *out = new maua_comm{ctx, c, rank, world};
  return MAUA_OK;
}

// the stream the communicator's transfers are enqueued on from now on (the streamed gather's side stream); use_ctx_stream != 0
// returns to the context's stream
int maua_comm_set_stream(maua_comm* comm, void* stream, int use_ctx_stream) {
  MAUA_REQUIRE(comm, "maua_comm_set_stream: comm is NULL");
  comm->own_stream = !use_ctx_stream;
  comm->stream = (hipStream_t)stream;
  return MAUA_OK;
}

// what the COMMUNICATOR says about itself (ncclCommCount / ncclCommUserRank), not what the caller passed to maua_comm_init: a
// benchmark line that names its rank count reads it back from RCCL
int maua_comm_count(maua_comm* comm, int* nranks, int* rank) {
  MAUA_REQUIRE(comm && comm->comm && nranks, "maua_comm_count: NULL argument");
  MAUA_REQUIRE(g_rccl.comm_count, "maua_comm_count: this librccl has no ncclCommCount");
  if (int rc = g_rccl.comm_count(comm->comm, nranks)) return rccl_fail("ncclCommCount", rc);
  if (rank) {
    *rank = -1;
    if (g_rccl.comm_user_rank)
      if (int rc = g_rccl.comm_user_rank(comm->comm, rank)) return rccl_fail("ncclCommUserRank", rc);
  }
  return MAUA_OK;
}

int maua_comm_destroy(maua_comm* comm) {
  if (!comm) return MAUA_OK;
  int rc = comm->comm ? g_rccl.comm_destroy(comm->comm) : 0;
  delete comm;
  return rc ? rccl_fail("ncclCommDestroy", rc) : MAUA_OK;
}

int maua_gather_frames(maua_comm* comm, const uint8_t* send, const long* bytes_per_rank, uint8_t* recv, int root) {
  MAUA_REQUIRE(comm && bytes_per_rank, "maua_gather_frames: NULL argument");
  MAUA_REQUIRE(root >= 0 && root < comm->world, "maua_gather_frames: root outside [0, world)");
  const long mine = bytes_per_rank[comm->rank];
  MAUA_REQUIRE(mine >= 0 && (mine == 0 || send), "maua_gather_frames: this rank's shard is missing");
  hipStream_t st = comm_stream(comm);
  if (comm->rank != root) {
    if (mine == 0) return MAUA_OK;
    if (int rc = g_rccl.send(send, (size_t)mine, /*ncclUint8*/ 1, root, comm->comm, st)) return rccl_fail("ncclSend", rc);
    return MAUA_OK;
  }
  MAUA_REQUIRE(recv, "maua_gather_frames: the root needs the receive buffer");
  long off = 0;
  if (int rc = g_rccl.group_start()) return rccl_fail("ncclGroupStart", rc);
  for (int r = 0; r < comm->world; r++) {
    const long n = bytes_per_rank[r];
    if (n < 0) { g_rccl.group_end(); return maua::fail("maua_gather_frames: negative shard size"); }
    if (n > 0 && r != root)
      if (int rc = g_rccl.recv(recv + off, (size_t)n, 1, r, comm->comm, st)) { g_rccl.group_end(); return rccl_fail("ncclRecv", rc); }
    off += n;
  }
  if (int rc = g_rccl.group_end()) return rccl_fail("ncclGroupEnd", rc);
  off = 0;
  for (int r = 0; r < root; r++) off += bytes_per_rank[r];
  if (mine > 0 && recv + off != send)
    MAUA_HIP_CHECK(hipMemcpyAsync(recv + off, send, (size_t)mine, hipMemcpyDeviceToDevice, st));
  return MAUA_OK;
}

// One ROUND of the streamed gather: like maua_gather_frames, but every rank's piece lands at its own byte offset of the
// root's clip buffer (offsets_per_rank) - the pieces of a round are the k-th finished chunk of every rank, which are not
// adjacent in the clip.  Ranks with bytes_per_rank[r] == 0 take no part in the round.  The root's own piece is not moved
// (the root renders straight into the clip buffer).  Called on a side stream while the next chunk renders.
int maua_gather_frames_at(maua_comm* comm, const uint8_t* send, const long* bytes_per_rank, uint8_t* recv_base,
                          const long* offsets_per_rank, int root) {
  MAUA_REQUIRE(comm && bytes_per_rank && offsets_per_rank, "maua_gather_frames_at: NULL argument");
  MAUA_REQUIRE(root >= 0 && root < comm->world, "maua_gather_frames_at: root outside [0, world)");
  const long mine = bytes_per_rank[comm->rank];
  hipStream_t st = comm_stream(comm);
  if (comm->rank != root) {
    if (mine <= 0) return MAUA_OK;
    MAUA_REQUIRE(send, "maua_gather_frames_at: this rank's piece is missing");
    if (int rc = g_rccl.send(send, (size_t)mine, /*ncclUint8*/ 1, root, comm->comm, st)) return rccl_fail("ncclSend", rc);
    return MAUA_OK;
  }
  MAUA_REQUIRE(recv_base, "maua_gather_frames_at: the root needs the clip buffer");
  bool any = false;
  for (int r = 0; r < comm->world; r++) any = any || (r != root && bytes_per_rank[r] > 0);
  if (!any) return MAUA_OK;
  if (int rc = g_rccl.group_start()) return rccl_fail("ncclGroupStart", rc);
  for (int r = 0; r < comm->world; r++) {
    const long n = bytes_per_rank[r];
    if (n <= 0 || r == root) continue;
    if (offsets_per_rank[r] < 0) { g_rccl.group_end(); return maua::fail("maua_gather_frames_at: negative offset"); }
    if (int rc = g_rccl.recv(recv_base + offsets_per_rank[r], (size_t)n, 1, r, comm->comm, st)) {
      g_rccl.group_end();
      return rccl_fail("ncclRecv", rc);
    }
  }
  if (int rc = g_rccl.group_end()) return rccl_fail("ncclGroupEnd", rc);
  return MAUA_OK;
}

}  // extern "C"
