// Data-parallel communication of the UC2 hot path on RCCL over xGMI (replaces the Horovod calls of reference
// utils/distributed.py:15-42 all_reduce_and_rescale_tensors and :99-147 broadcast_tensors, and the hvd.init /
// rank / size bookkeeping of pretrain.py:384-388).
//
// One communicator per process (one process per GPU).  The library owns a side HIP stream: a bucket's all-reduce is
// ordered after the compute stream by an event (the gradients of that bucket are final), runs on the side stream
// while the compute stream goes on with the rest of backward, and uc2_comm_wait orders the compute stream behind
// everything issued so far.  RCCL chooses the algorithm (8 GPUs fully connected, 7 xGMI links each): nothing forces
// a ring.  librccl is opened lazily (dlopen) by uc2_comm_init, so the library loads and every compute entry point
// works on a machine without it.
//
// Return codes: 0 ok, < 0 argument / state error, > 0 hipError_t, >= 10000 ncclResult_t + 10000.
#include "common.h"
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <string.h>

namespace {
struct Rccl {
  void* so = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
} g_rccl;

struct Comm {
  ncclComm_t comm = nullptr;
  hipStream_t side = nullptr;
  hipEvent_t ready = nullptr, ready2 = nullptr, done = nullptr;
  int rank = 0, world = 1;
  bool pending = false;
} g_comm;

int rccl_open() {
  if (g_rccl.so) return 0;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* so = nullptr;
  for (const char* n : names) { so = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (so) break; }
  if (!so) { uc2_set_error(__FILE__, __LINE__, "librccl.so not found (dlopen)"); return -2; }
  // resolved into a local table and published only when every REQUIRED symbol exists: a failed open must not leave a handle
  // that makes the next rccl_open() return 0 with null function pointers behind it
  Rccl r;
#define UC2_SYM(F, N) do { *(void**)(&r.F) = dlsym(so, N); if (!r.F) { uc2_set_error(__FILE__, __LINE__, "librccl: missing symbol " N); dlclose(so); return -2; } } while (0)
#define UC2_SYM_OPT(F, N) do { *(void**)(&r.F) = dlsym(so, N); } while (0)
  UC2_SYM(GetUniqueId, "ncclGetUniqueId"); UC2_SYM(CommInitRank, "ncclCommInitRank"); UC2_SYM(CommDestroy, "ncclCommDestroy");
  UC2_SYM(AllReduce, "ncclAllReduce"); UC2_SYM(Broadcast, "ncclBroadcast"); UC2_SYM(GetErrorString, "ncclGetErrorString");
  UC2_SYM_OPT(CommCount, "ncclCommCount"); UC2_SYM_OPT(GetVersion, "ncclGetVersion"); UC2_SYM_OPT(CommAbort, "ncclCommAbort");   // diagnostics / teardown: optional
#undef UC2_SYM
#undef UC2_SYM_OPT
  r.so = so;
  g_rccl = r;
  return 0;
}
}  // namespace

#define UC2_NCCL(call) do { ncclResult_t r__ = (call); if (r__ != ncclSuccess) { uc2_set_error(__FILE__, __LINE__, g_rccl.GetErrorString(r__)); return 10000 + (int)r__; } } while (0)
#define UC2_HIP(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { uc2_set_error(__FILE__, __LINE__, hipGetErrorString(e__)); return (int)e__; } } while (0)

extern "C" int uc2_comm_unique_id_bytes(void) { return NCCL_UNIQUE_ID_BYTES; }

// rank 0 creates the id; the host side carries it to the other ranks (any out-of-band channel: torch.distributed store,
// MPI, a file) and every rank passes the same bytes to uc2_comm_init
extern "C" int uc2_comm_unique_id(void* out, int bytes) {
  UC2_CHECK_ARG(out && bytes >= NCCL_UNIQUE_ID_BYTES);
  if (int rc = rccl_open()) return rc;
  ncclUniqueId id;
  UC2_NCCL(g_rccl.GetUniqueId(&id));
  memcpy(out, id.internal, NCCL_UNIQUE_ID_BYTES);
  return 0;
}

// collective: every rank of the job calls it with the same world and id, after hipSetDevice to its own GPU
extern "C" int uc2_comm_init(int rank, int world, const void* unique_id, int bytes) {
  UC2_CHECK_ARG(world >= 1 && rank >= 0 && rank < world && unique_id && bytes >= NCCL_UNIQUE_ID_BYTES);
  UC2_CHECK_ARG(g_comm.comm == nullptr);
  if (int rc = rccl_open()) return rc;
  ncclUniqueId id;
  memcpy(id.internal, unique_id, NCCL_UNIQUE_ID_BYTES);
  // built in a local object and published only when every piece exists: a failed stream / event creation must not
  // leave a communicator that looks initialised (comm != nullptr) with null handles and refuses a second init
  Comm c;
  UC2_NCCL(g_rccl.CommInitRank(&c.comm, world, id, rank));
  hipError_t e = hipStreamCreateWithFlags(&c.side, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c.ready, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c.ready2, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c.done, hipEventDisableTiming);
  int count = world;                                   // (a librccl without ncclCommCount: trust the argument)
  ncclResult_t r = (e == hipSuccess && g_rccl.CommCount) ? g_rccl.CommCount(c.comm, &count) : ncclSuccess;
  if (e != hipSuccess || r != ncclSuccess || count != world) {
    if (c.done) (void)hipEventDestroy(c.done);
    if (c.ready2) (void)hipEventDestroy(c.ready2);
    if (c.ready) (void)hipEventDestroy(c.ready);
    if (c.side) (void)hipStreamDestroy(c.side);
    g_rccl.CommDestroy(c.comm);
    if (e != hipSuccess) { uc2_set_error(__FILE__, __LINE__, hipGetErrorString(e)); return (int)e; }
    if (r != ncclSuccess) { uc2_set_error(__FILE__, __LINE__, g_rccl.GetErrorString(r)); return 10000 + (int)r; }
    uc2_set_error(__FILE__, __LINE__, "ncclCommCount disagrees with the world size passed to uc2_comm_init");
    return -3;
  }
  c.rank = rank; c.world = count; c.pending = false;
  g_comm = c;
  return 0;
}
extern "C" int uc2_comm_rank(void) { return g_comm.comm ? g_comm.rank : -1; }
// ranks in the communicator as RCCL itself counted them at init (ncclCommCount), 0 when there is none
extern "C" int uc2_comm_world(void) { return g_comm.comm ? g_comm.world : 0; }
// "major.minor.patch" of the loaded librccl (ncclGetVersion); works before uc2_comm_init
extern "C" int uc2_comm_version(char* out, int bytes) {
  UC2_CHECK_ARG(out && bytes >= 16);
  if (int rc = rccl_open()) return rc;
  int v = 0;
  if (!g_rccl.GetVersion) { uc2_set_error(__FILE__, __LINE__, "librccl: no ncclGetVersion"); return -2; }
  UC2_NCCL(g_rccl.GetVersion(&v));
  const int major = v >= 10000 ? v / 10000 : v / 1000, minor = v >= 10000 ? (v % 10000) / 100 : (v % 1000) / 100, patch = v % 100;
  snprintf(out, bytes, "%d.%d.%d", major, minor, patch);
  return 0;
}

static int comm_dtype(int dtype, ncclDataType_t* t) {
  if (dtype == 0) { *t = ncclFloat32; return 0; }
  if (dtype == 1) { *t = ncclBfloat16; return 0; }
  return -1;
}

// in-place all-reduce of one gradient bucket on the side stream, ordered after everything enqueued so far on
// `compute_stream` AND (if not NULL) on `other_stream`; average != 0 -> mean over ranks (Horovod's default,
// utils/distributed.py:34), else sum.  Returns immediately; the result may be read on a stream only after uc2_comm_wait on
// that stream.  `other_stream` is the caller's weight-gradient side stream: a layer's bucket is final when both the main
// stream (bias / LayerNorm gradients, dX chain) and that stream (dW GEMMs) have reached this point -- the collective waits
// for the two events itself, so the main stream is never serialised behind the weight-gradient GEMMs (round 3 joined them).
extern "C" int uc2_comm_allreduce_bucket_after(void* buf, size_t count, int dtype, int average, void* compute_stream,
                                               void* other_stream) {
  UC2_CHECK_ARG(g_comm.comm != nullptr);
  ncclDataType_t t;
  UC2_CHECK_ARG(comm_dtype(dtype, &t) == 0);
  if (count == 0) return 0;
  UC2_CHECK_ARG(buf != nullptr);
  UC2_HIP(hipEventRecord(g_comm.ready, (hipStream_t)compute_stream));
  UC2_HIP(hipStreamWaitEvent(g_comm.side, g_comm.ready, 0));
  if (other_stream && other_stream != compute_stream) {
    UC2_HIP(hipEventRecord(g_comm.ready2, (hipStream_t)other_stream));
    UC2_HIP(hipStreamWaitEvent(g_comm.side, g_comm.ready2, 0));
  }
  UC2_NCCL(g_rccl.AllReduce(buf, buf, count, t, average ? ncclAvg : ncclSum, g_comm.comm, g_comm.side));
  g_comm.pending = true;
  return 0;
}
extern "C" int uc2_comm_allreduce_bucket(void* buf, size_t count, int dtype, int average, void* compute_stream) {
  return uc2_comm_allreduce_bucket_after(buf, count, dtype, average, compute_stream, nullptr);
}

// rank `root`'s bytes everywhere (initial parameter broadcast, pretrain.py:457), in place, same ordering rules
extern "C" int uc2_comm_broadcast(void* buf, size_t count, int dtype, int root, void* compute_stream) {
  UC2_CHECK_ARG(g_comm.comm != nullptr && root >= 0 && root < g_comm.world);
  ncclDataType_t t;
  UC2_CHECK_ARG(comm_dtype(dtype, &t) == 0);
  if (count == 0) return 0;
  UC2_CHECK_ARG(buf != nullptr);
  UC2_HIP(hipEventRecord(g_comm.ready, (hipStream_t)compute_stream));
  UC2_HIP(hipStreamWaitEvent(g_comm.side, g_comm.ready, 0));
  UC2_NCCL(g_rccl.Broadcast(buf, buf, count, t, root, g_comm.comm, g_comm.side));
  g_comm.pending = true;
  return 0;
}

// `stream` waits (on the device, not the host) for every collective issued so far
extern "C" int uc2_comm_wait(void* stream) {
  UC2_CHECK_ARG(g_comm.comm != nullptr);
  if (!g_comm.pending) return 0;
  UC2_HIP(hipEventRecord(g_comm.done, g_comm.side));
  UC2_HIP(hipStreamWaitEvent((hipStream_t)stream, g_comm.done, 0));
  return 0;
}

static void comm_release_handles() {
  (void)hipEventDestroy(g_comm.ready); (void)hipEventDestroy(g_comm.ready2); (void)hipEventDestroy(g_comm.done);
  (void)hipStreamDestroy(g_comm.side);
  g_comm = Comm();
}
extern "C" int uc2_comm_destroy(void) {
  if (!g_comm.comm) return 0;
  (void)hipStreamSynchronize(g_comm.side);
  g_rccl.CommDestroy(g_comm.comm);
  comm_release_handles();
  return 0;
}
// teardown of a rank that is leaving on an error while its peers may be inside a collective: ncclCommAbort does not wait for
// outstanding operations (ncclCommDestroy can block for ever there); falls back to nothing but releasing the handles when the
// loaded librccl has no abort
extern "C" int uc2_comm_abort(void) {
  if (!g_comm.comm) return 0;
  if (g_rccl.CommAbort) g_rccl.CommAbort(g_comm.comm);
  comm_release_handles();
  return 0;
}
