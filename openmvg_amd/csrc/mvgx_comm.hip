// libmvgx_hip.so — RCCL communicator for the one exchange step of the BA path: one process per GPU,
// points (and all their observations) partitioned across ranks, camera blocks replicated; per LM iteration the partial
// reduced camera system / camera column norms / scalars are summed with ncclAllReduce over xGMI on the solver's stream.
// RCCL is bound at run time (dlopen) so that single-GPU users need no librccl.
#include <dlfcn.h>

#include <atomic>

#include <cstring>

#include "mvgx_comm.h"

namespace mvgx {
namespace {

typedef struct { char internal[128]; } UniqueId;  // ncclUniqueId (rccl.h:43, NCCL_UNIQUE_ID_BYTES = 128)
typedef void* Comm;
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(Comm*, int, UniqueId, int);
typedef int (*AllReduceFn)(const void*, void*, size_t, int /*dtype*/, int /*op*/, Comm, hipStream_t);
typedef int (*CommDestroyFn)(Comm);
typedef const char* (*GetErrorStringFn)(int);

constexpr int kNcclFloat64 = 8;  // rccl.h:467
constexpr int kNcclSum = 0, kNcclMax = 2;  // rccl.h:448-450

struct Api {
  void* handle = nullptr;
  GetUniqueIdFn get_unique_id = nullptr;
  CommInitRankFn comm_init_rank = nullptr;
  AllReduceFn all_reduce = nullptr;
  CommDestroyFn comm_destroy = nullptr;
  CommDestroyFn comm_abort = nullptr;   // ncclCommAbort(comm): same signature as ncclCommDestroy
  GetErrorStringFn error_string = nullptr;
};

Api* api() {
  static Api a;
  static bool tried = false;
  if (!tried) {
    tried = true;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
      a.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (a.handle) break;
    }
    if (a.handle) {
      a.get_unique_id = reinterpret_cast<GetUniqueIdFn>(dlsym(a.handle, "ncclGetUniqueId"));
      a.comm_init_rank = reinterpret_cast<CommInitRankFn>(dlsym(a.handle, "ncclCommInitRank"));
      a.all_reduce = reinterpret_cast<AllReduceFn>(dlsym(a.handle, "ncclAllReduce"));
      a.comm_destroy = reinterpret_cast<CommDestroyFn>(dlsym(a.handle, "ncclCommDestroy"));
      a.comm_abort = reinterpret_cast<CommDestroyFn>(dlsym(a.handle, "ncclCommAbort"));
      a.error_string = reinterpret_cast<GetErrorStringFn>(dlsym(a.handle, "ncclGetErrorString"));
    }
  }
  if (!a.handle || !a.get_unique_id || !a.comm_init_rank || !a.all_reduce || !a.comm_destroy) {
    set_error("RCCL not available: %s", a.handle ? "missing symbols in librccl" : dlerror());
    return nullptr;
  }
  return &a;
}

const char* nccl_err(Api* a, int rc) { return a->error_string ? a->error_string(rc) : "nccl error"; }

}  // namespace

struct RcclComm {
  Comm comm = nullptr;               // set once by rccl_init, released by rccl_destroy only (after the shard threads have joined)
  std::atomic<bool> aborted{false};  // rccl_abort ran: all-reduce refuses; the communicator object itself stays until destroy
  int world = 1, rank = 0;
};

int rccl_unique_id(void* out128) {
  Api* a = api();
  if (!a) return MVGX_ERR_HIP;
  UniqueId id;
  const int rc = a->get_unique_id(&id);
  MVGX_REQUIRE(rc == 0, MVGX_ERR_HIP, "ncclGetUniqueId: %s", nccl_err(a, rc));
  memcpy(out128, id.internal, sizeof(id.internal));
  return MVGX_OK;
}

int rccl_init(RcclComm** out, int world, int rank, const void* unique_id) {
  Api* a = api();
  if (!a) return MVGX_ERR_HIP;
  MVGX_REQUIRE(out && unique_id && world >= 1 && rank >= 0 && rank < world, MVGX_ERR_ARG, "rccl_init: bad argument");
  UniqueId id;
  memcpy(id.internal, unique_id, sizeof(id.internal));
  auto* c = new RcclComm();
  c->world = world; c->rank = rank;
  const int rc = a->comm_init_rank(&c->comm, world, id, rank);
  if (rc != 0) {
    set_error("ncclCommInitRank(world %d, rank %d): %s", world, rank, nccl_err(a, rc));
    delete c;
    return MVGX_ERR_HIP;
  }
  *out = c;
  return MVGX_OK;
}

void rccl_destroy(RcclComm* c) {
  if (!c) return;
  Api* a = api();
  // (an aborted communicator was released by ncclCommAbort: ncclCommDestroy on it would be a second release)
  if (a && c->comm && !c->aborted.load()) a->comm_destroy(c->comm);
  delete c;
}

// A rank of the same process failed outside a collective: the others may be blocked inside ncclAllReduce (or in the stream
// synchronisation behind it) waiting for a contribution that will never come. ncclCommAbort may be called from another thread
// while the communicator is in use: it fails the operations in flight. The handle is NOT cleared here (a shard thread
// could load it just before the abort and call ncclAllReduce on freed state, and the plain pointer was a data race): `aborted` is
// an atomic flag, set first and exactly once; all-reduce checks it before every call, and the RcclComm object lives until
// rccl_destroy, which the owner calls after the shard threads have joined.
void rccl_abort(RcclComm* c) {
  if (!c || !c->comm) return;
  if (c->aborted.exchange(true)) return;
  Api* a = api();
  if (a && a->comm_abort) a->comm_abort(c->comm);
}

int rccl_allreduce_f64(RcclComm* c, double* device_buffer, uint64_t count, int op, hipStream_t stream) {
  Api* a = api();
  if (!a) return MVGX_ERR_HIP;
  MVGX_REQUIRE(c && c->comm && !c->aborted.load(), MVGX_ERR_STATE, "all-reduce on an aborted RCCL communicator (another device shard failed): destroy the context");
  const int rc = a->all_reduce(device_buffer, device_buffer, count, kNcclFloat64, op == MVGX_REDUCE_MAX ? kNcclMax : kNcclSum,
                               c->comm, stream);
  MVGX_REQUIRE(rc == 0, MVGX_ERR_HIP, "ncclAllReduce(%llu doubles): %s", (unsigned long long)count, nccl_err(a, rc));
  return MVGX_OK;
}

// The data-type / operator codes above are restated from rccl.h (the library is bound by dlopen, its header is not
// included): one tiny all-reduce of known values per operator right after the communicator exists turns a mismatch with
// the installed librccl into an error instead of silently wrong sums.
int rccl_self_check(RcclComm* c, hipStream_t stream) {
  double* d = nullptr;
  MVGX_HIP(hipMalloc(reinterpret_cast<void**>(&d), 4 * sizeof(double)));
  const double r1 = (double)(c->rank + 1);
  const double h[4] = {1.0, r1, r1, -r1};
  double got[4] = {0, 0, 0, 0};
  int rc = MVGX_OK;
  if (hipMemcpyAsync(d, h, sizeof(h), hipMemcpyHostToDevice, stream) != hipSuccess) rc = MVGX_ERR_HIP;
  if (!rc) rc = rccl_allreduce_f64(c, d, 2, MVGX_REDUCE_SUM, stream);
  if (!rc) rc = rccl_allreduce_f64(c, d + 2, 2, MVGX_REDUCE_MAX, stream);
  if (!rc && (hipMemcpyAsync(got, d, sizeof(got), hipMemcpyDeviceToHost, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess)) rc = MVGX_ERR_HIP;
  (void)hipFree(d);
  if (rc) return rc;
  const double w = (double)c->world;
  MVGX_REQUIRE(got[0] == w && got[1] == w * (w + 1) / 2 && got[2] == w && got[3] == -1.0, MVGX_ERR_HIP,
               "RCCL self-check failed (sum: %g %g, max: %g %g for world %d): data-type / operator codes of this librccl differ from "
               "the ones mvgx_comm.hip assumes (rccl.h ncclFloat64 = 8, ncclSum = 0, ncclMax = 2)", got[0], got[1], got[2], got[3], c->world);
  return MVGX_OK;
}

}  // namespace mvgx

extern "C" int mvgx_comm_unique_id(void* out128) {
  MVGX_REQUIRE(out128 != nullptr, MVGX_ERR_ARG, "mvgx_comm_unique_id: NULL buffer");
  return mvgx::rccl_unique_id(out128);
}
