// One-shot all-reduce for small fp32 messages over peer-mapped device memory (xGMI on an MI355X node).
//
// Every rank owns a fine-grained device region (inboxes + flags) that its peers map with an IPC handle.
// One launch of one workgroup per rank:
//   1. push: the rank stores its message into slot [parity][rank] of EVERY peer's inbox (one xGMI link
//      per peer, all of them busy at once);
//   2. signal: system-scope release store of the call's sequence number into each peer's flag
//      [parity][rank];
//   3. wait: bounded spin until the rank's own flags [parity][0..world) all carry the sequence number;
//   4. reduce: sum the world slots of the own inbox (rank order, so every rank gets the same bits).
// Two inbox/flag sets alternate with the parity of the sequence number: a rank can only enter call n+2
// after every rank has pushed call n+1, which each of them does after finishing its own call n, so slot
// set n%2 is never overwritten while somebody still reads it.
// This is the latency path (metric scalars, SyncBatchNorm statistics, a few KB): a bandwidth-sized
// reduction belongs to RCCL.  (Role in the reference: the NCCL all-reduce behind `dist.all_reduce` of
// 0-dim metric tensors, /root/reference/2.distributed.py:71-75,219-223.)
#include "common.h"
#include <string.h>

#define P2P_MAX_WORLD 16

struct P2PArgs {
  float* inbox[P2P_MAX_WORLD];     // peer-mapped: [2][world][slot_elems]
  unsigned* flags[P2P_MAX_WORLD];  // peer-mapped: [2][world]
  const float* in;
  float* out;
  int* err;                        // set to 1 when the wait timed out (results are then undefined)
  long long timeout_ticks;         // wall_clock64 ticks (100 MHz)
  float scale;
  int n, rank, world, slot_elems;
  unsigned seq;
};

__global__ void __launch_bounds__(512) p2p_allreduce_oneshot_kernel(P2PArgs a) {
  const int t = threadIdx.x, nt = blockDim.x;
  const int par = (int)(a.seq & 1u);
  const long long slot = (long long)(par * a.world + a.rank) * a.slot_elems;
  // 1. push (the own copy goes through the same path: slot [par][rank] of the own inbox)
  for (int p = 0; p < a.world; ++p) {
    float* __restrict__ dst = a.inbox[p] + slot;
    for (int i = t; i < a.n; i += nt) dst[i] = a.in[i];
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // system scope: the stores above are visible to the peers
  __syncthreads();
  // 2. signal + 3. wait: thread p talks to peer p
  if (t < a.world) {
    __hip_atomic_store(a.flags[t] + par * a.world + a.rank, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned* mine = a.flags[a.rank] + par * a.world + t;
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(mine, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != a.seq) {
      if (wall_clock64() - t0 > a.timeout_ticks) {  // every wave leaves the loop: a dead peer cannot hang the GPU
        *a.err = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(16);
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // the other threads read what the waiting threads acquired
  // 4. reduce
  const float* __restrict__ box = a.inbox[a.rank] + (long long)par * a.world * a.slot_elems;
  for (int i = t; i < a.n; i += nt) {
    float s = 0.f;
    for (int p = 0; p < a.world; ++p) s += box[(long long)p * a.slot_elems + i];
    a.out[i] = s * a.scale;
  }
}

static int p2p_err(hipError_t e) { return e == hipSuccess ? 0 : (int)e; }

// Fine-grained (cross-device coherent) device memory, zero-filled.
PDA_EXPORT int pda_p2p_alloc(long long bytes, void** ptr) {
  hipError_t e = hipExtMallocWithFlags(ptr, (size_t)bytes, hipDeviceMallocFinegrained);
  if (e != hipSuccess) return (int)e;
  e = hipMemset(*ptr, 0, (size_t)bytes);
  if (e != hipSuccess) return (int)e;
  return p2p_err(hipDeviceSynchronize());
}

PDA_EXPORT int pda_p2p_free(void* ptr) { return p2p_err(hipFree(ptr)); }

PDA_EXPORT int pda_p2p_handle_bytes() { return (int)sizeof(hipIpcMemHandle_t); }

PDA_EXPORT int pda_p2p_get_handle(void* ptr, unsigned char* out) {
  hipIpcMemHandle_t h;
  hipError_t e = hipIpcGetMemHandle(&h, ptr);
  if (e != hipSuccess) return (int)e;
  memcpy(out, &h, sizeof(h));
  return 0;
}

PDA_EXPORT int pda_p2p_open_handle(const unsigned char* in, void** ptr) {
  hipIpcMemHandle_t h;
  memcpy(&h, in, sizeof(h));
  return p2p_err(hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess));
}

PDA_EXPORT int pda_p2p_close_handle(void* ptr) { return p2p_err(hipIpcCloseMemHandle(ptr)); }

// bases[p]: the region of rank p as mapped into this process (own region for p == rank).
// Region layout: [2][world][slot_elems] floats, then [2][world] unsigned flags.
PDA_EXPORT int pda_p2p_allreduce(void* const* bases, const float* in, float* out, int n, int rank, int world,
                                 int slot_elems, unsigned seq, float scale, int* err, int timeout_ms,
                                 hipStream_t st) {
  if (world < 1 || world > P2P_MAX_WORLD || rank < 0 || rank >= world || n < 0 || n > slot_elems || seq == 0)
    return -1;
  if (n == 0) return 0;
  P2PArgs a;
  for (int p = 0; p < P2P_MAX_WORLD; ++p) {
    char* b = reinterpret_cast<char*>(bases[p < world ? p : 0]);
    a.inbox[p] = reinterpret_cast<float*>(b);
    a.flags[p] = reinterpret_cast<unsigned*>(b + (size_t)2 * world * slot_elems * sizeof(float));
  }
  a.in = in; a.out = out; a.err = err;
  a.timeout_ticks = (long long)timeout_ms * 100000LL;
  a.scale = scale; a.n = n; a.rank = rank; a.world = world; a.slot_elems = slot_elems; a.seq = seq;
  const int threads = n >= 512 ? 512 : (n > 64 ? ((n + 63) / 64) * 64 : 64);
  hipLaunchKernelGGL(p2p_allreduce_oneshot_kernel, dim3(1), dim3(threads), 0, st, a);
  return pda_check_launch();
}
