// Native RCCL communicator: the process-group layer under DistributedDataParallel / the hvd engine
// when this package drives RCCL itself instead of going through torch.distributed
// (reference: `dist.init_process_group(backend='nccl')` + `dist.all_reduce` / `dist.barrier`,
// /root/reference/2.distributed.py:98,73,219; ProcessGroupNCCL there).
//
// This file owns the RCCL communicator (unique id, ncclCommInitRank, the collective calls, async
// error query / abort) AND the stream discipline around it: a dedicated comm stream per communicator
// that waits (hipStreamWaitEvent) for the producing compute stream, runs the collective and records
// a completion event from a ring the consumer stream later waits on - on the device, without
// blocking the host, so gradient buckets overlap with the rest of backward.  A collective returns a
// ticket; pda_rccl_wait(ticket, stream) orders `stream` after it, pda_rccl_query polls it.  The
// 128-byte unique id travels through the rendezvous store (python side).
//
// RCCL is resolved with dlopen at first use, after PyTorch has loaded its own copy, so the process
// ends up with a single RCCL instance whichever library provides it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <time.h>

#include <cstdlib>
#include <cstring>
#include <map>
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
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*ReduceScatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
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
    PDA_SYM(CommCount, "ncclCommCount")
    PDA_SYM(CommUserRank, "ncclCommUserRank")
    PDA_SYM(GetVersion, "ncclGetVersion")
    PDA_SYM(GroupStart, "ncclGroupStart")
    PDA_SYM(GroupEnd, "ncclGroupEnd")
    PDA_SYM(AllReduce, "ncclAllReduce")
    PDA_SYM(Broadcast, "ncclBroadcast")
    PDA_SYM(AllGather, "ncclAllGather")
    PDA_SYM(ReduceScatter, "ncclReduceScatter")
    PDA_SYM(Send, "ncclSend")
    PDA_SYM(Recv, "ncclRecv")
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

constexpr int kEventRing = 512;

struct Comm {
  ncclComm_t comm = nullptr;
  int device = 0, rank = 0, world = 1;
  hipStream_t stream = nullptr;        // dedicated comm stream (overlap mode)
  // "inputs produced" markers and completion events, slot = ticket % kEventRing: one event per
  // collective in flight on both sides, nothing is re-recorded while a wait on it may still be queued.
  hipEvent_t ready[kEventRing] = {};
  hipEvent_t done[kEventRing] = {};
  int handoff_single = 0;              // diagnostics only: re-record ready[0] for every collective
  // Value hand-off (PDA_RCCL_HANDOFF=value, where the device supports stream memory operations): the
  // producer stream writes a sequence number into signal memory (hipStreamWriteValue64), the comm
  // stream waits for "counter >= sequence" (hipStreamWaitValue64), and the same in the other
  // direction with done_ctr.  The runtime runs both as one-thread kernels (the wait spins on a wave),
  // so the event ring, whose waits are barrier packets handled by the command processor, is the default.
  int handoff_value = 0;
  std::map<hipStream_t, std::pair<unsigned long long*, unsigned long long>> ready_ctr;  // per producer stream
  unsigned long long* done_ctr = nullptr;
  long long next_ticket = 0;
  int probe_memset = 0;                // diagnostics: put a real device operation on the comm stream per collective
  void* probe_word = nullptr;
  std::mutex mu;                       // collectives may be issued from autograd worker threads
  std::string last_error;
};

int fail(Comm* c, const std::string& what) {
  if (c) c->last_error = what;
  return -1;
}

struct DeviceScope {                   // make the communicator's device current for this thread
  int prev = -1;
  bool ok = true;
  explicit DeviceScope(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) { ok = false; prev = -1; return; }
    if (prev != dev && hipSetDevice(dev) != hipSuccess) ok = false;
    if (prev == dev) prev = -1;
  }
  ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// One collective: `fn(stream)` issues the RCCL call.  overlap = 0: it is simply the next operation
// of `compute` (ticket -> already ordered).  overlap = 1: comm stream waits for everything enqueued
// on `compute` so far, runs the call, records the ticket's completion event.
template <typename F>
long long issue(Comm* c, hipStream_t compute, int overlap, F&& fn) {
  std::lock_guard<std::mutex> lk(c->mu);
  if (!c->comm) return fail(c, "communicator was aborted or destroyed");
  DeviceScope ds(c->device);
  if (!ds.ok) return fail(c, "hipSetDevice failed");
  hipStream_t s = overlap ? c->stream : compute;
  if (overlap && c->handoff_value) {
    auto it = c->ready_ctr.find(compute);
    if (it == c->ready_ctr.end()) {
      unsigned long long* p = nullptr;
      if (hipExtMallocWithFlags((void**)&p, 8, hipMallocSignalMemory) != hipSuccess) return fail(c, "signal memory");
      *p = 0;
      it = c->ready_ctr.emplace(compute, std::make_pair(p, 0ull)).first;
    }
    const unsigned long long seq = ++it->second.second;
    if (hipStreamWriteValue64(compute, it->second.first, seq, 0) != hipSuccess)
      return fail(c, "hipStreamWriteValue64(ready) failed");
    if (hipStreamWaitValue64(c->stream, it->second.first, seq, hipStreamWaitValueGte, ~0ull) != hipSuccess)
      return fail(c, "hipStreamWaitValue64(comm) failed");
  } else if (overlap) {
    hipEvent_t rdy = c->ready[c->handoff_single ? 0 : c->next_ticket % kEventRing];
    if (hipEventRecord(rdy, compute) != hipSuccess) return fail(c, "hipEventRecord(ready) failed");
    if (hipStreamWaitEvent(c->stream, rdy, 0) != hipSuccess) return fail(c, "hipStreamWaitEvent(comm) failed");
    if (c->probe_memset) (void)hipMemsetAsync(c->probe_word, 0, 4, c->stream);
  }
  const ncclResult_t r = fn(s);
  if (r != ncclSuccess) return fail(c, std::string("RCCL: ") + api().GetErrorString(r));
  const long long t = c->next_ticket++;
  // host-side completion query / bounded host wait always go through an event of the issuing stream
  if (hipEventRecord(c->done[t % kEventRing], s) != hipSuccess) return fail(c, "hipEventRecord(done) failed");
  if (overlap && c->handoff_value &&
      hipStreamWriteValue64(c->stream, c->done_ctr, (unsigned long long)t + 1, 0) != hipSuccess)
    return fail(c, "hipStreamWriteValue64(done) failed");
  return t;
}

}  // namespace

PDA_EXPORT int pda_rccl_available() { return api().error.empty() && api().lib ? 1 : 0; }

PDA_EXPORT int pda_rccl_version() {
  int v = 0;
  if (!api().error.empty() || api().GetVersion(&v) != ncclSuccess) return -1;
  return v;
}

PDA_EXPORT int pda_rccl_unique_id(char* out128) {
  Api& a = api();
  if (!a.error.empty()) return -1;
  ncclUniqueId id;
  if (a.GetUniqueId(&id) != ncclSuccess) return -2;
  memcpy(out128, id.internal, NCCL_UNIQUE_ID_BYTES);
  return 0;
}

// Blocking: every rank of the communicator must call it (ncclCommInitRank is collective).
// priority: 0 = the comm stream has the same (default) priority as PyTorch's compute streams - the
// only safe choice.  +1 / -1 (greatest / least priority of the device) exist for diagnostics: the
// command processor serves hardware queues in strict priority order and keeps serving a queue whose
// head packet is a wait (barrier packet or stream-wait kernel), so a wait at the head of the
// higher-priority queue for work that sits in the lower-priority queue never ends.  The comm stream
// waits for the compute stream (inputs ready) AND the compute stream waits for the comm stream
// (results ready), so either priority order deadlocks as soon as the host has queued the wait before
// the device reached the work it waits for.
PDA_EXPORT void* pda_rccl_create(const char* id128, int rank, int world, int device, int priority) {
  Api& a = api();
  if (!a.error.empty() || rank < 0 || rank >= world) return nullptr;
  DeviceScope ds(device);
  if (!ds.ok) return nullptr;
  Comm* c = new Comm();
  c->device = device; c->rank = rank; c->world = world;
  int least = 0, greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&least, &greatest);   // greatest = numerically smallest value
  bool ok = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking,
                                        priority > 0 ? greatest : (priority < 0 ? least : 0)) == hipSuccess;
  for (int i = 0; ok && i < kEventRing; ++i)
    ok = hipEventCreateWithFlags(&c->ready[i], hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&c->done[i], hipEventDisableTiming) == hipSuccess;
  const char* ho = getenv("PDA_RCCL_HANDOFF");
  c->handoff_single = ho && !strcmp(ho, "single");
  int can_value = 0;
  (void)hipDeviceGetAttribute(&can_value, hipDeviceAttributeCanUseStreamWaitValue, device);
  c->handoff_value = can_value && ho && !strcmp(ho, "value");
  if (ok && c->handoff_value) {
    ok = hipExtMallocWithFlags((void**)&c->done_ctr, 8, hipMallocSignalMemory) == hipSuccess;
    if (ok) *c->done_ctr = 0;
  }
  if (ok && getenv("PDA_RCCL_PROBE_MEMSET") && getenv("PDA_RCCL_PROBE_MEMSET")[0] == '1') {
    c->probe_memset = hipMalloc(&c->probe_word, 256) == hipSuccess;
  }
  if (ok) {
    ncclUniqueId id;
    memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
    ok = a.CommInitRank(&c->comm, world, id, rank) == ncclSuccess;
  }
  if (!ok) {
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return nullptr;
  }
  return c;
}

// The caller has drained the streams it used with this communicator.
PDA_EXPORT void pda_rccl_destroy(void* h) {
  Comm* c = static_cast<Comm*>(h);
  if (!c) return;
  DeviceScope ds(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm) api().CommDestroy(c->comm);
  for (int i = 0; i < kEventRing; ++i) {
    if (c->done[i]) (void)hipEventDestroy(c->done[i]);
    if (c->ready[i]) (void)hipEventDestroy(c->ready[i]);
  }
  if (c->stream) (void)hipStreamDestroy(c->stream);
  for (auto& kv : c->ready_ctr) (void)hipFree(kv.second.first);
  if (c->done_ctr) (void)hipFree(c->done_ctr);
  delete c;
}

// 2 = value hand-off (stream memory operations), 1 = event ring, 0 = single re-recorded event (diagnostics)
PDA_EXPORT int pda_rccl_handoff_mode(void* h) {
  Comm* c = static_cast<Comm*>(h);
  return c->handoff_value ? 2 : (c->handoff_single ? 0 : 1);
}

PDA_EXPORT const char* pda_rccl_last_error(void* h) { return static_cast<Comm*>(h)->last_error.c_str(); }

PDA_EXPORT void* pda_rccl_stream(void* h) { return static_cast<Comm*>(h)->stream; }

// Size / rank as RCCL itself reports them (what a benchmark prints to prove N ranks took part).
PDA_EXPORT int pda_rccl_nranks(void* h) {
  Comm* c = static_cast<Comm*>(h);
  int n = -1;
  if (!c->comm || api().CommCount(c->comm, &n) != ncclSuccess) return -1;
  return n;
}
PDA_EXPORT int pda_rccl_user_rank(void* h) {
  Comm* c = static_cast<Comm*>(h);
  int n = -1;
  if (!c->comm || api().CommUserRank(c->comm, &n) != ncclSuccess) return -1;
  return n;
}

// Order `stream` after the collective `ticket` (device-side; the host does not block).  A ticket
// older than the event ring has been overwritten by a later collective of the same in-order comm
// stream, so waiting for the slot's current record is still sufficient.
PDA_EXPORT int pda_rccl_wait(void* h, long long ticket, hipStream_t stream) {
  Comm* c = static_cast<Comm*>(h);
  if (ticket < 0) return 0;
  std::lock_guard<std::mutex> lk(c->mu);
  DeviceScope ds(c->device);
  if (c->handoff_value) {
    if (hipStreamWaitValue64(stream, c->done_ctr, (unsigned long long)ticket + 1, hipStreamWaitValueGte, ~0ull) !=
        hipSuccess)
      return (int)fail(c, "hipStreamWaitValue64(consumer) failed");
    return 0;
  }
  if (hipStreamWaitEvent(stream, c->done[ticket % kEventRing], 0) != hipSuccess)
    return (int)fail(c, "hipStreamWaitEvent(consumer) failed");
  return 0;
}

// 1 = complete, 0 = still running, <0 = error.
PDA_EXPORT int pda_rccl_query(void* h, long long ticket) {
  Comm* c = static_cast<Comm*>(h);
  if (ticket < 0) return 1;
  std::lock_guard<std::mutex> lk(c->mu);
  DeviceScope ds(c->device);
  const hipError_t e = hipEventQuery(c->done[ticket % kEventRing]);
  if (e == hipSuccess) return 1;
  if (e == hipErrorNotReady) { (void)hipGetLastError(); return 0; }
  return (int)fail(c, "hipEventQuery failed");
}

// Host-blocking wait for one ticket, bounded: 0 = done, 1 = timed out, <0 = error.
PDA_EXPORT int pda_rccl_sync(void* h, long long ticket, int timeout_ms) {
  Comm* c = static_cast<Comm*>(h);
  if (ticket < 0) return 0;
  hipEvent_t ev;
  { std::lock_guard<std::mutex> lk(c->mu); ev = c->done[ticket % kEventRing]; }
  DeviceScope ds(c->device);
  for (int waited = 0; timeout_ms <= 0 || waited < timeout_ms; ++waited) {
    const hipError_t e = hipEventQuery(ev);
    if (e == hipSuccess) return 0;
    if (e != hipErrorNotReady) return (int)fail(c, "hipEventQuery failed");
    (void)hipGetLastError();
    struct timespec ts = {0, 1000000};
    nanosleep(&ts, nullptr);
  }
  return 1;
}

PDA_EXPORT long long pda_rccl_all_reduce(void* h, const void* send, void* recv, long long count, int dtype, int op,
                                         hipStream_t compute, int overlap) {
  Comm* c = static_cast<Comm*>(h);
  ncclDataType_t dt; ncclRedOp_t ro;
  if (!to_nccl_dtype(dtype, &dt) || !to_nccl_op(op, &ro) || count < 0) return fail(c, "bad dtype/op/count");
  return issue(c, compute, overlap, [&](hipStream_t s) {
    return api().AllReduce(send, recv, (size_t)count, dt, ro, c->comm, s); });
}

PDA_EXPORT long long pda_rccl_broadcast(void* h, void* buf, long long count, int dtype, int root, hipStream_t compute,
                                        int overlap) {
  Comm* c = static_cast<Comm*>(h);
  ncclDataType_t dt;
  if (!to_nccl_dtype(dtype, &dt) || root < 0 || root >= c->world || count < 0) return fail(c, "bad dtype/root/count");
  return issue(c, compute, overlap, [&](hipStream_t s) {
    return api().Broadcast(buf, buf, (size_t)count, dt, root, c->comm, s); });
}

PDA_EXPORT long long pda_rccl_all_gather(void* h, const void* send, void* recv, long long count_per_rank, int dtype,
                                         hipStream_t compute, int overlap) {
  Comm* c = static_cast<Comm*>(h);
  ncclDataType_t dt;
  if (!to_nccl_dtype(dtype, &dt) || count_per_rank < 0) return fail(c, "bad dtype/count");
  return issue(c, compute, overlap, [&](hipStream_t s) {
    return api().AllGather(send, recv, (size_t)count_per_rank, dt, c->comm, s); });
}

PDA_EXPORT long long pda_rccl_reduce_scatter(void* h, const void* send, void* recv, long long count_per_rank,
                                             int dtype, int op, hipStream_t compute, int overlap) {
  Comm* c = static_cast<Comm*>(h);
  ncclDataType_t dt; ncclRedOp_t ro;
  if (!to_nccl_dtype(dtype, &dt) || !to_nccl_op(op, &ro) || count_per_rank < 0) return fail(c, "bad dtype/op/count");
  return issue(c, compute, overlap, [&](hipStream_t s) {
    return api().ReduceScatter(send, recv, (size_t)count_per_rank, dt, ro, c->comm, s); });
}

// Exchange with one peer: send `send_count` elements and receive `recv_count` in one RCCL group (what the
// halving / doubling steps of Adasum need).
PDA_EXPORT long long pda_rccl_sendrecv(void* h, const void* send, long long send_count, void* recv, long long recv_count,
                                       int dtype, int peer, hipStream_t compute, int overlap) {
  Comm* c = static_cast<Comm*>(h);
  ncclDataType_t dt;
  if (!to_nccl_dtype(dtype, &dt) || peer < 0 || peer >= c->world || send_count < 0 || recv_count < 0)
    return fail(c, "bad dtype/peer/count");
  return issue(c, compute, overlap, [&](hipStream_t s) {
    ncclResult_t r = api().GroupStart();
    if (r == ncclSuccess && send_count > 0) r = api().Send(send, (size_t)send_count, dt, peer, c->comm, s);
    if (r == ncclSuccess && recv_count > 0) r = api().Recv(recv, (size_t)recv_count, dt, peer, c->comm, s);
    const ncclResult_t e = api().GroupEnd();
    return r != ncclSuccess ? r : e;
  });
}

// Several all-reduces as ONE RCCL group launch (the buckets of a delayed / flushed backward).
PDA_EXPORT long long pda_rccl_all_reduce_group(void* h, int n, void* const* bufs, const long long* counts, int dtype,
                                               int op, hipStream_t compute, int overlap) {
  Comm* c = static_cast<Comm*>(h);
  ncclDataType_t dt; ncclRedOp_t ro;
  if (!to_nccl_dtype(dtype, &dt) || !to_nccl_op(op, &ro) || n < 0) return fail(c, "bad dtype/op/count");
  return issue(c, compute, overlap, [&](hipStream_t s) {
    ncclResult_t r = api().GroupStart();
    for (int i = 0; i < n && r == ncclSuccess; ++i)
      r = api().AllReduce(bufs[i], bufs[i], (size_t)counts[i], dt, ro, c->comm, s);
    const ncclResult_t e = api().GroupEnd();
    return r != ncclSuccess ? r : e;
  });
}

// Diagnostics for a stalled job: which of the last tickets have completed, whether the hand-off
// marker and the comm stream are idle ("t<ticket>=<1|0> ... ready=<1|0> stream=<1|0>").
PDA_EXPORT int pda_rccl_debug_state(void* h, char* out, int cap) {
  Comm* c = static_cast<Comm*>(h);
  DeviceScope ds(c->device);
  std::string s = "next_ticket=" + std::to_string(c->next_ticket);
  const long long lo = c->next_ticket > 16 ? c->next_ticket - 16 : 0;
  for (long long t = lo; t < c->next_ticket; ++t) {
    const hipError_t e = hipEventQuery(c->done[t % kEventRing]);
    s += " t" + std::to_string(t) + "=" + (e == hipSuccess ? "1" : e == hipErrorNotReady ? "0" : "E");
  }
  s += " ready:";
  for (long long t = lo; t < c->next_ticket; ++t) {
    const hipError_t e = hipEventQuery(c->ready[c->handoff_single ? 0 : t % kEventRing]);
    s += (e == hipSuccess ? "1" : e == hipErrorNotReady ? "0" : "E");
  }
  if (c->handoff_value) {
    s += " done_ctr=" + std::to_string(*c->done_ctr);
    for (auto& kv : c->ready_ctr)
      s += " ready_ctr=" + std::to_string(*kv.second.first) + "/" + std::to_string(kv.second.second);
  }
  hipError_t e = hipStreamQuery(c->stream);
  s += std::string(" stream=") + (e == hipSuccess ? "1" : e == hipErrorNotReady ? "0" : "E");
  (void)hipGetLastError();
  if (cap > 0) {
    strncpy(out, s.c_str(), (size_t)cap - 1);
    out[cap - 1] = 0;
  }
  return (int)s.size();
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
