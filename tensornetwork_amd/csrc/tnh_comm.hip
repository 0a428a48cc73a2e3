// K8: collectives over xGMI (RCCL) on the library's own stream.
//
// One process per GPU, one communicator per process.  The reference has no
// distributed path; these entry points exist for the two
// partitions of SURVEY.md 8e: bond-sliced contractor paths (ONE all-reduce of
// the small result) and an M-sharded pairwise contraction (ONE all-gather of
// the row blocks).  Everything is enqueued on tnh::stream(), i.e. in order
// with the kernels that produced the buffers -- no device-wide synchronise on
// either side of a collective.
//
// librccl is opened lazily (dlopen) by tnh_comm_unique_id / tnh_comm_init, so
// single-GPU processes never load it.  Bootstrap: rank 0 calls
// tnh_comm_unique_id and hands the 128 bytes to the other ranks over any host
// channel (tensornetwork_amd/comm.py uses a TCP rendezvous on
// MASTER_ADDR:MASTER_PORT+k); every rank then calls tnh_comm_init.
#include "tnh_internal.h"
#include <dlfcn.h>
#include <stdlib.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
// A ROCm install without the RCCL development headers still builds the single-GPU library: the handful of types
// the dlopen'ed entry points need are declared here (ABI of rccl.h 2.x), and load_rccl() reports UNSUPPORTED at run
// time when librccl itself is missing.
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclInt32 = 2, ncclInt64 = 4, ncclFloat16 = 6, ncclFloat32 = 7, ncclFloat64 = 8,
               ncclBfloat16 = 9 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3 } ncclRedOp_t;
#endif

namespace tnh {
namespace {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;   // optional
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi g_api;
ncclComm_t g_comm = nullptr;
std::atomic<bool> g_init_abandoned{false};   // a bring-up timed out: its helper thread may still be inside RCCL
int g_rank = 0, g_world = 1;

template <typename F>
bool load_sym(F& slot, const char* name) {
  slot = reinterpret_cast<F>(dlsym(g_api.handle, name));
  return slot != nullptr;
}

int load_rccl() {
  if (g_api.handle) return TNH_OK;
  const char* override_path = getenv("TNH_RCCL_LIBRARY");
  const char* candidates[] = {override_path, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* c : candidates) {
    if (!c || !*c) continue;
    g_api.handle = dlopen(c, RTLD_NOW | RTLD_LOCAL);
    if (g_api.handle) break;
  }
  if (!g_api.handle) {
    set_error("cannot load librccl (%s)", dlerror());
    return TNH_ERR_UNSUPPORTED;
  }
  const bool ok = load_sym(g_api.GetUniqueId, "ncclGetUniqueId") && load_sym(g_api.CommInitRank, "ncclCommInitRank") &&
                  load_sym(g_api.CommDestroy, "ncclCommDestroy") && load_sym(g_api.AllReduce, "ncclAllReduce") &&
                  load_sym(g_api.AllGather, "ncclAllGather") && load_sym(g_api.Broadcast, "ncclBroadcast") &&
                  load_sym(g_api.GetErrorString, "ncclGetErrorString");
  load_sym(g_api.CommAbort, "ncclCommAbort");
  if (!ok) {
    set_error("librccl is missing a required symbol (%s)", dlerror());
    dlclose(g_api.handle);
    g_api = RcclApi();
    return TNH_ERR_UNSUPPORTED;
  }
  return TNH_OK;
}

#define TNH_NCCL(call)                                                                        \
  do {                                                                                        \
    ncclResult_t _r = (call);                                                                 \
    if (_r != ncclSuccess) {                                                                  \
      set_error("%s failed: %s (%s:%d)", #call, g_api.GetErrorString(_r), __FILE__, __LINE__); \
      return TNH_ERR_HIP;                                                                     \
    }                                                                                         \
  } while (0)

// element type RCCL reduces in, and how many of them one element of `dt` is
bool reduce_type(int dt, ncclDataType_t* t, int* mult) {
  *mult = 1;
  switch (dt) {
    case TNH_F32: *t = ncclFloat32; return true;
    case TNH_F64: *t = ncclFloat64; return true;
    case TNH_BF16: *t = ncclBfloat16; return true;
    case TNH_F16: *t = ncclFloat16; return true;
    case TNH_I32: *t = ncclInt32; return true;
    case TNH_I64: *t = ncclInt64; return true;
    case TNH_C64: *t = ncclFloat32; *mult = 2; return true;   // sum acts on (re, im) separately
    case TNH_C128: *t = ncclFloat64; *mult = 2; return true;
    default: return false;
  }
}

}  // namespace
}  // namespace tnh

using namespace tnh;

extern "C" {

int tnh_comm_available(void) {
  // Pre-flight for the lock-step bootstrap (comm.py): everything ncclCommInitRank needs locally, checked
  // WITHOUT entering a collective -- a rank that would fail before ncclCommInitRank must say so while the
  // others can still listen, otherwise they block inside it for good.
  TNH_NEED_INIT();
  int rc = load_rccl();
  if (rc != TNH_OK) return rc;
  int dev = -1;
  TNH_HIP(hipGetDevice(&dev));
  TNH_REQUIRE(g_comm == nullptr, "tnh_comm_available: a communicator already exists (tnh_comm_destroy first)");
  return TNH_OK;
}

int tnh_comm_unique_id(void* host_id) {
  TNH_REQUIRE(host_id != nullptr, "tnh_comm_unique_id: null buffer");
  int rc = load_rccl();
  if (rc != TNH_OK) return rc;
  ncclUniqueId id;
  TNH_NCCL(g_api.GetUniqueId(&id));
  static_assert(sizeof(id) == TNH_COMM_ID_BYTES, "ncclUniqueId size");
  memcpy(host_id, &id, sizeof(id));
  return TNH_OK;
}

int tnh_comm_init(const void* host_id, int rank, int world) {
  TNH_NEED_INIT();
  TNH_REQUIRE(host_id != nullptr && world >= 1 && rank >= 0 && rank < world, "tnh_comm_init: bad rank %d / world %d",
              rank, world);
  TNH_REQUIRE(g_comm == nullptr, "tnh_comm_init: a communicator already exists (tnh_comm_destroy first)");
  // After a bring-up timeout a helper thread may still sit inside ncclCommInitRank (see below): a second bring-up in
  // this process would race with it, and a communicator it completes later is never tracked.  The library stays
  // poisoned for communicators: report and exit is the only way on.
  TNH_REQUIRE(!g_init_abandoned.load(), "tnh_comm_init: an earlier bring-up timed out in this process and its helper "
              "thread may still be inside RCCL; no further communicator can be created here (exit the process)");
  int rc = load_rccl();
  if (rc != TNH_OK) return rc;
  ncclUniqueId id;
  memcpy(&id, host_id, sizeof(id));
  // RCCL 2.2x printf()s a version banner ("RCCL version : ...", 5 lines) to stdout from its first
  // communicator, whatever NCCL_DEBUG says.  A host program whose stdout is a protocol (bench.py: ONE JSON
  // line) must not get library chatter there: stdout is pointed at stderr for the duration of the call.
  fflush(stdout);
  const int saved_stdout = dup(STDOUT_FILENO);
  if (saved_stdout >= 0) dup2(STDERR_FILENO, STDOUT_FILENO);
  // Bounded bring-up (round 4): ncclCommInitRank is collective and has no timeout of its own -- a peer that never
  // arrives, or a bootstrap interface that does not route (measured on the GPU box: no return within 150 s), blocks
  // it for good.  It runs on a helper thread; this thread waits TNH_COMM_INIT_TIMEOUT_S seconds (default 600 -- on a
  // fresh box the first touch of librccl.so alone was measured at minutes, page by page from the image; 0 =
  // wait for ever, the old behaviour).  On a timeout the call returns TNH_ERR_TIMEOUT and the helper is left behind,
  // still inside RCCL (there is no communicator handle to abort yet): the process is expected to report and exit.
  struct InitState {
    std::mutex mu;
    std::condition_variable cv;
    bool done = false;
    bool abandoned = false;
    ncclResult_t rc = ncclSuccess;
    ncclComm_t comm = nullptr;
  };
  auto st = std::make_shared<InitState>();
  int dev = 0;
  (void)hipGetDevice(&dev);
  double limit_s = 600.0;
  if (const char* e = getenv("TNH_COMM_INIT_TIMEOUT_S")) limit_s = atof(e);
  auto fn = g_api.CommInitRank;
  auto abort_fn = g_api.CommAbort;
  std::thread helper([st, fn, abort_fn, id, world, rank, dev]() {
    (void)hipSetDevice(dev);                       // the current device is per thread
    ncclComm_t c = nullptr;
    const ncclResult_t r = fn(&c, world, id, rank);
    bool abandoned;
    {
      std::lock_guard<std::mutex> lk(st->mu);
      st->rc = r;
      st->comm = c;
      st->done = true;
      abandoned = st->abandoned;
      st->cv.notify_all();
    }
    // nobody is waiting any more (the caller returned TNH_ERR_TIMEOUT): the communicator is ours to release
    if (abandoned && r == ncclSuccess && c != nullptr && abort_fn) (void)abort_fn(c);
  });
  bool finished;
  {
    std::unique_lock<std::mutex> lk(st->mu);
    if (limit_s > 0.0)
      finished = st->cv.wait_for(lk, std::chrono::duration<double>(limit_s), [&] { return st->done; });
    else {
      st->cv.wait(lk, [&] { return st->done; });
      finished = true;
    }
  }
  if (!finished) {
    std::lock_guard<std::mutex> lk(st->mu);
    if (st->done) finished = true;            // it returned between the wait and here
    else st->abandoned = true;
  }
  // (after a timeout the abandoned helper may still print RCCL's banner to the restored stdout; the caller gets
  //  TNH_ERR_TIMEOUT and is expected to report on stderr and exit)
  fflush(stdout);
  if (saved_stdout >= 0) {
    dup2(saved_stdout, STDOUT_FILENO);
    close(saved_stdout);
  }
  if (!finished) {
    g_init_abandoned.store(true);
    helper.detach();
    set_error("ncclCommInitRank did not return within %.0f s on rank %d of %d (TNH_COMM_INIT_TIMEOUT_S): a peer is "
              "missing or RCCL's bootstrap interface does not route (NCCL_SOCKET_IFNAME)", limit_s, rank, world);
    return TNH_ERR_TIMEOUT;
  }
  helper.join();
  const ncclResult_t init_rc = st->rc;
  g_comm = st->comm;
  if (init_rc != ncclSuccess) {
    g_comm = nullptr;
    set_error("ncclCommInitRank failed: %s", g_api.GetErrorString(init_rc));
    return TNH_ERR_HIP;
  }
  g_rank = rank;
  g_world = world;
  return TNH_OK;
}

int tnh_comm_info(int* rank, int* world) {
  if (rank) *rank = g_comm ? g_rank : 0;
  if (world) *world = g_comm ? g_world : 0;   // 0 = no communicator
  return TNH_OK;
}

int tnh_comm_abort(void) {
  // error path: peers may never have joined, so nothing here may wait for them
  if (g_comm) {
    ncclResult_t r = g_api.CommAbort ? g_api.CommAbort(g_comm) : ncclSuccess;
    g_comm = nullptr;
    g_rank = 0;
    g_world = 1;
    if (r != ncclSuccess) {
      set_error("ncclCommAbort failed: %s", g_api.GetErrorString(r));
      return TNH_ERR_HIP;
    }
  }
  return TNH_OK;
}

int tnh_comm_destroy(void) {
  if (g_comm) {
    if (stream()) (void)hipStreamSynchronize(stream());
    ncclResult_t r = g_api.CommDestroy(g_comm);
    g_comm = nullptr;
    g_rank = 0;
    g_world = 1;
    if (r != ncclSuccess) {
      set_error("ncclCommDestroy failed: %s", g_api.GetErrorString(r));
      return TNH_ERR_HIP;
    }
  }
  return TNH_OK;
}

int tnh_allreduce(void* buf, int64_t count, int dtype, int op) {
  TNH_NEED_INIT();
  TNH_REQUIRE(g_comm != nullptr, "tnh_allreduce: no communicator (tnh_comm_init)");
  TNH_REQUIRE(buf != nullptr || count == 0, "tnh_allreduce: null buffer");
  TNH_REQUIRE(count >= 0, "tnh_allreduce: negative count");
  ncclDataType_t t;
  int mult;
  TNH_REQUIRE(reduce_type(dtype, &t, &mult), "tnh_allreduce: unsupported dtype %d", dtype);
  TNH_REQUIRE(op >= 0 && op <= 2, "tnh_allreduce: op must be 0 (sum), 1 (max) or 2 (min)");
  TNH_REQUIRE(mult == 1 || op == 0, "tnh_allreduce: complex tensors support sum only");
  if (count == 0) return TNH_OK;
  const ncclRedOp_t ops[3] = {ncclSum, ncclMax, ncclMin};
  TNH_NCCL(g_api.AllReduce(buf, buf, (size_t)count * mult, t, ops[op], g_comm, stream()));
  return TNH_OK;
}

int tnh_allreduce_sum(void* buf, int64_t count, int dtype) { return tnh_allreduce(buf, count, dtype, 0); }

int tnh_allgather(void* dst, const void* src, int64_t nbytes) {
  TNH_NEED_INIT();
  TNH_REQUIRE(g_comm != nullptr, "tnh_allgather: no communicator (tnh_comm_init)");
  TNH_REQUIRE(nbytes >= 0 && (nbytes == 0 || (dst && src)), "tnh_allgather: bad arguments");
  if (nbytes == 0) return TNH_OK;
  // 8-byte words where the block allows it: fewer, wider elements for RCCL's copy kernels
  if (nbytes % 8 == 0)
    TNH_NCCL(g_api.AllGather(src, dst, (size_t)(nbytes / 8), ncclInt64, g_comm, stream()));
  else
    TNH_NCCL(g_api.AllGather(src, dst, (size_t)nbytes, ncclInt8, g_comm, stream()));
  return TNH_OK;
}

int tnh_broadcast(void* buf, int64_t nbytes, int root) {
  TNH_NEED_INIT();
  TNH_REQUIRE(g_comm != nullptr, "tnh_broadcast: no communicator (tnh_comm_init)");
  TNH_REQUIRE(nbytes >= 0 && (nbytes == 0 || buf) && root >= 0 && root < g_world, "tnh_broadcast: bad arguments");
  if (nbytes == 0) return TNH_OK;
  TNH_NCCL(g_api.Broadcast(buf, buf, (size_t)nbytes, ncclInt8, root, g_comm, stream()));
  return TNH_OK;
}

}  // extern "C"
