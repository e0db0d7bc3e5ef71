// Native RCCL communicator: the process-group layer under DistributedDataParallel / the hvd engine
// when this package drives RCCL itself instead of going through torch.distributed
// (reference: `dist.init_process_group(backend='nccl')` + `dist.all_reduce` / `dist.barrier`,
// /root/reference/2.distributed.py:98,73,219; ProcessGroupNCCL there).
//
// This file owns the RCCL communicator (unique id, ncclCommInitRank, the collective calls, async
// error query / abort); comm/rccl.py owns the stream discipline around it: a dedicated high-priority
// comm stream that waits for the producing compute stream, runs the collective and records an event
// the consumer stream later waits on - on the device, without blocking the host, so gradient
// buckets overlap with the rest of backward.  The 128-byte unique id travels through the rendezvous
// store (python side).
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
  int device = 0, rank = 0, world = 1;
  std::string last_error;
};

int fail(Comm* c, const std::string& what) {
  if (c) c->last_error = what;
  return -1;
}

// Run one RCCL call on `stream` with the communicator's device current.
template <typename F>
int run_on(Comm* c, F&& fn) {
  int prev = 0;
  if (hipGetDevice(&prev) != hipSuccess) return fail(c, "hipGetDevice failed");
  if (prev != c->device && hipSetDevice(c->device) != hipSuccess) return fail(c, "hipSetDevice failed");
  const ncclResult_t r = fn();
  if (prev != c->device) (void)hipSetDevice(prev);
  if (r != ncclSuccess) return fail(c, std::string("RCCL: ") + api().GetErrorString(r));
  return 0;
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
  const bool ok = a.CommInitRank(&c->comm, world, id, rank) == ncclSuccess;
  (void)hipSetDevice(prev);
  if (!ok) {
    delete c;
    return nullptr;
  }
  return c;
}

// The caller has drained the streams it used with this communicator.
PDA_EXPORT void pda_rccl_destroy(void* h) {
  Comm* c = static_cast<Comm*>(h);
  if (!c) return;
  int prev = 0;
  (void)hipGetDevice(&prev);
  (void)hipSetDevice(c->device);
  if (c->comm) api().CommDestroy(c->comm);
  (void)hipSetDevice(prev);
  delete c;
}

PDA_EXPORT const char* pda_rccl_last_error(void* h) { return static_cast<Comm*>(h)->last_error.c_str(); }

PDA_EXPORT int pda_rccl_all_reduce(void* h, const void* send, void* recv, long long count, int dtype, int op,
                                   hipStream_t stream) {
  Comm* c = static_cast<Comm*>(h);
  ncclDataType_t dt; ncclRedOp_t ro;
  if (!to_nccl_dtype(dtype, &dt) || !to_nccl_op(op, &ro) || count < 0) return fail(c, "bad dtype/op/count");
  return run_on(c, [&] { return api().AllReduce(send, recv, (size_t)count, dt, ro, c->comm, stream); });
}

PDA_EXPORT int pda_rccl_broadcast(void* h, void* buf, long long count, int dtype, int root, hipStream_t stream) {
  Comm* c = static_cast<Comm*>(h);
  ncclDataType_t dt;
  if (!to_nccl_dtype(dtype, &dt) || root < 0 || root >= c->world || count < 0) return fail(c, "bad dtype/root/count");
  return run_on(c, [&] { return api().Broadcast(buf, buf, (size_t)count, dt, root, c->comm, stream); });
}

PDA_EXPORT int pda_rccl_all_gather(void* h, const void* send, void* recv, long long count_per_rank, int dtype,
                                   hipStream_t stream) {
  Comm* c = static_cast<Comm*>(h);
  ncclDataType_t dt;
  if (!to_nccl_dtype(dtype, &dt) || count_per_rank < 0) return fail(c, "bad dtype/count");
  return run_on(c, [&] { return api().AllGather(send, recv, (size_t)count_per_rank, dt, c->comm, stream); });
}

PDA_EXPORT int pda_rccl_reduce_scatter(void* h, const void* send, void* recv, long long count_per_rank, int dtype,
                                       int op, hipStream_t stream) {
  Comm* c = static_cast<Comm*>(h);
  ncclDataType_t dt; ncclRedOp_t ro;
  if (!to_nccl_dtype(dtype, &dt) || !to_nccl_op(op, &ro) || count_per_rank < 0) return fail(c, "bad dtype/op/count");
  return run_on(c, [&] { return api().ReduceScatter(send, recv, (size_t)count_per_rank, dt, ro, c->comm, stream); });
}

// Watchdog support: 0 = healthy, >0 = RCCL async error code.
PDA_EXPORT int pda_rccl_async_error(void* h) {
  Comm* c = static_cast<Comm*>(h);
  if (!c->comm) return -1;
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
