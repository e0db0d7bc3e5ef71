// Native RCCL communicator: the process-group layer under DistributedDataParallel / the hvd engine
// when this package drives RCCL itself instead of going through torch.distributed
// (reference: `dist.init_process_group(backend='nccl')` + `dist.all_reduce` / `dist.barrier`,
// /root/reference/2.distributed.py:98,73,219; ProcessGroupNCCL there).
//
// One communicator = one RCCL comm + one dedicated high-priority HIP stream.  A collective is
// enqueued as  record(event on the caller's compute stream) -> comm stream waits for it -> RCCL call
// on the comm stream -> record(done event); `wait` makes a compute stream wait for the done event
// on the device, without blocking the host, so gradient buckets overlap with the rest of backward.
// The 128-byte unique id travels through the rendezvous store (python side).
//
// RCCL is resolved with dlopen at first use, after PyTorch has loaded its own copy, so the process
// ends up with a single RCCL instance whichever library provides it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "common.h"

namespace {

struct Api {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string error;
};

Api& api() {
  static Api a;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      a.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (a.lib) break;
    }
    if (!a.lib) {
      a.error = std::string("cannot load RCCL: ") + dlerror();
      return;
    }
#define PDA_SYM(field, name)                                        \
  a.field = reinterpret_cast<decltype(a.field)>(dlsym(a.lib, name)); \
  if (!a.field) a.error = std::string("RCCL symbol missing: ") + name;
    PDA_SYM(GetUniqueId, "ncclGetUniqueId")
    PDA_SYM(CommInitRank, "ncclCommInitRank")
    PDA_SYM(CommDestroy, "ncclCommDestroy")
    PDA_SYM(CommAbort, "ncclCommAbort")
    PDA_SYM(CommGetAsyncError, "ncclCommGetAsyncError")
    PDA_SYM(AllReduce, "ncclAllReduce")
    PDA_SYM(Broadcast, "ncclBroadcast")
    PDA_SYM(AllGather, "ncclAllGather")
    PDA_SYM(ReduceScatter, "ncclReduceScatter")
    PDA_SYM(GetErrorString, "ncclGetErrorString")
#undef PDA_SYM
  });
  return a;
}

// dtype codes of this package (ops/_lib.py) + integer types used by the metric / buffer collectives
bool to_nccl_dtype(int dt, ncclDataType_t* out) {
  switch (dt) {
    case 0: *out = ncclFloat32; return true;
    case 1: *out = ncclFloat16; return true;
    case 2: *out = ncclBfloat16; return true;
    case 3: *out = ncclUint8; return true;
    case 4: *out = ncclFloat64; return true;
    case 5: *out = ncclInt32; return true;
    case 6: *out = ncclInt64; return true;
    default: return false;
  }
}
bool to_nccl_op(int op, ncclRedOp_t* out) {  // SUM, PROD, MIN, MAX, AVG: the codes of the shm backend
  switch (op) {
    case 0: *out = ncclSum; return true;
    case 1: *out = ncclProd; return true;
    case 2: *out = ncclMin; return true;
    case 3: *out = ncclMax; return true;
    case 4: *out = ncclAvg; return true;
    default: return false;
  }
}

struct Comm {
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;
  int device = 0, rank = 0, world = 1;
  std::mutex mu;
  std::vector<hipEvent_t> free_events;
  std::vector<hipEvent_t> works;  // work id -> done event (null once released)
  std::string last_error;

  hipEvent_t get_event() {
    if (!free_events.empty()) {
      hipEvent_t e = free_events.back();
      free_events.pop_back();
      return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
    return e;
  }
};

int fail(Comm* c, const std::string& what) {
  if (c) c->last_error = what;
  return -1;
}

// Order the comm stream after `compute`, run fn on it, hand back a work id.
template <typename F>
long long enqueue(Comm* c, hipStream_t compute, F&& fn) {
  std::lock_guard<std::mutex> g(c->mu);
  int prev = 0;
  if (hipGetDevice(&prev) != hipSuccess) return fail(c, "hipGetDevice failed");
  if (prev != c->device && hipSetDevice(c->device) != hipSuccess) return fail(c, "hipSetDevice failed");
  long long id = -1;
  hipEvent_t ready = c->get_event(), done = c->get_event();
  do {
    if (!ready || !done) { fail(c, "hipEventCreate failed"); break; }
    if (hipEventRecord(ready, compute) != hipSuccess) { fail(c, "hipEventRecord failed"); break; }
    if (hipStreamWaitEvent(c->stream, ready, 0) != hipSuccess) { fail(c, "hipStreamWaitEvent failed"); break; }
    const ncclResult_t r = fn(c->stream);
    if (r != ncclSuccess) {
      fail(c, std::string("RCCL: ") + api().GetErrorString(r));
      break;
    }
    if (hipEventRecord(done, c->stream) != hipSuccess) { fail(c, "hipEventRecord failed"); break; }
    c->works.push_back(done);
    id = (long long)c->works.size() - 1;
    done = nullptr;
  } while (false);
  if (ready) c->free_events.push_back(ready);  // safe to reuse: stream-ordered record/wait pairs
  if (done) c->free_events.push_back(done);
  if (prev != c->device) (void)hipSetDevice(prev);
  return id;
}

}  // namespace

PDA_EXPORT int pda_rccl_available() { return api().error.empty() && api().lib ? 1 : 0; }

PDA_EXPORT int pda_rccl_unique_id(char* out128) {
  Api& a = api();
  if (!a.error.empty()) return -1;
  ncclUniqueId id;
  if (a.GetUniqueId(&id) != ncclSuccess) return -2;
  memcpy(out128, id.internal, NCCL_UNIQUE_ID_BYTES);
  return 0;
}

// Blocking: every rank of the communicator must call it (ncclCommInitRank is collective).
PDA_EXPORT void* pda_rccl_create(const char* id128, int rank, int world, int device) {
  Api& a = api();
  if (!a.error.empty() || rank < 0 || rank >= world) return nullptr;
  int prev = 0;
  if (hipGetDevice(&prev) != hipSuccess || hipSetDevice(device) != hipSuccess) return nullptr;
  Comm* c = new Comm();
  c->device = device; c->rank = rank; c->world = world;
  ncclUniqueId id;
  memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
  int lo = 0, hi = 0;
  bool ok = hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess &&
            hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, hi) == hipSuccess &&
            a.CommInitRank(&c->comm, world, id, rank) == ncclSuccess;
  (void)hipSetDevice(prev);
  if (!ok) {
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return nullptr;
  }
  return c;
}

PDA_EXPORT void pda_rccl_destroy(void* h) {
  Comm* c = static_cast<Comm*>(h);
  if (!c) return;
  int prev = 0;
  (void)hipGetDevice(&prev);
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  if (c->comm) api().CommDestroy(c->comm);
  for (hipEvent_t e : c->works) if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : c->free_events) (void)hipEventDestroy(e);
  (void)hipStreamDestroy(c->stream);
  (void)hipSetDevice(prev);
  delete c;
}

PDA_EXPORT const char* pda_rccl_last_error(void* h) { return static_cast<Comm*>(h)->last_error.c_str(); }
PDA_EXPORT void* pda_rccl_stream(void* h) { return static_cast<Comm*>(h)->stream; }

PDA_EXPORT long long pda_rccl_all_reduce(void* h, const void* send, void* recv, long long count, int dtype, int op,
                                         hipStream_t compute) {
  Comm* c = static_cast<Comm*>(h);
  ncclDataType_t dt; ncclRedOp_t ro;
  if (!to_nccl_dtype(dtype, &dt) || !to_nccl_op(op, &ro) || count < 0) return fail(c, "bad dtype/op/count");
  return enqueue(c, compute, [&](hipStream_t s) { return api().AllReduce(send, recv, (size_t)count, dt, ro, c->comm, s); });
}

PDA_EXPORT long long pda_rccl_broadcast(void* h, void* buf, long long count, int dtype, int root, hipStream_t compute) {
  Comm* c = static_cast<Comm*>(h);
  ncclDataType_t dt;
  if (!to_nccl_dtype(dtype, &dt) || root < 0 || root >= c->world || count < 0) return fail(c, "bad dtype/root/count");
  return enqueue(c, compute, [&](hipStream_t s) { return api().Broadcast(buf, buf, (size_t)count, dt, root, c->comm, s); });
}

PDA_EXPORT long long pda_rccl_all_gather(void* h, const void* send, void* recv, long long count_per_rank, int dtype,
                                         hipStream_t compute) {
  Comm* c = static_cast<Comm*>(h);
  ncclDataType_t dt;
  if (!to_nccl_dtype(dtype, &dt) || count_per_rank < 0) return fail(c, "bad dtype/count");
  return enqueue(c, compute,
                 [&](hipStream_t s) { return api().AllGather(send, recv, (size_t)count_per_rank, dt, c->comm, s); });
}

PDA_EXPORT long long pda_rccl_reduce_scatter(void* h, const void* send, void* recv, long long count_per_rank, int dtype,
                                             int op, hipStream_t compute) {
  Comm* c = static_cast<Comm*>(h);
  ncclDataType_t dt; ncclRedOp_t ro;
  if (!to_nccl_dtype(dtype, &dt) || !to_nccl_op(op, &ro) || count_per_rank < 0) return fail(c, "bad dtype/op/count");
  return enqueue(c, compute, [&](hipStream_t s) {
    return api().ReduceScatter(send, recv, (size_t)count_per_rank, dt, ro, c->comm, s);
  });
}

// Make `compute` wait (on the device) for work `id`; release its event.
PDA_EXPORT int pda_rccl_wait(void* h, long long id, hipStream_t compute) {
  Comm* c = static_cast<Comm*>(h);
  std::lock_guard<std::mutex> g(c->mu);
  if (id < 0 || id >= (long long)c->works.size() || !c->works[(size_t)id]) return fail(c, "unknown work id");
  hipEvent_t e = c->works[(size_t)id];
  if (hipStreamWaitEvent(compute, e, 0) != hipSuccess) return fail(c, "hipStreamWaitEvent failed");
  c->works[(size_t)id] = nullptr;
  c->free_events.push_back(e);
  // compact the table once everything issued so far has been waited for
  bool all_free = true;
  for (hipEvent_t w : c->works) all_free = all_free && !w;
  if (all_free) c->works.clear();
  return 0;
}

// 1 = finished, 0 = still running, <0 = error.  Does not release the work.
PDA_EXPORT int pda_rccl_query(void* h, long long id) {
  Comm* c = static_cast<Comm*>(h);
  std::lock_guard<std::mutex> g(c->mu);
  if (id < 0 || id >= (long long)c->works.size() || !c->works[(size_t)id]) return 1;
  const hipError_t r = hipEventQuery(c->works[(size_t)id]);
  if (r == hipSuccess) return 1;
  if (r == hipErrorNotReady) return 0;
  return fail(c, "hipEventQuery failed");
}

// Host-blocking completion of everything enqueued so far (barrier semantics need it).
PDA_EXPORT int pda_rccl_synchronize(void* h) {
  Comm* c = static_cast<Comm*>(h);
  return hipStreamSynchronize(c->stream) == hipSuccess ? 0 : fail(c, "hipStreamSynchronize failed");
}

// Watchdog support: 0 = healthy, >0 = RCCL async error code.
PDA_EXPORT int pda_rccl_async_error(void* h) {
  Comm* c = static_cast<Comm*>(h);
  ncclResult_t st = ncclSuccess;
  if (api().CommGetAsyncError(c->comm, &st) != ncclSuccess) return -1;
  return (int)st;
}

PDA_EXPORT int pda_rccl_abort(void* h) {
  Comm* c = static_cast<Comm*>(h);
  if (!c->comm) return 0;
  const ncclResult_t r = api().CommAbort(c->comm);
  c->comm = nullptr;
  return r == ncclSuccess ? 0 : -1;
}
