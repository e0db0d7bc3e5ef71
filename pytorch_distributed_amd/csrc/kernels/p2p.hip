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
// The sequence number lives in device memory (seq_state[0]; seq_state[1] counts the workgroups that are done):
// every workgroup of a launch reads it, uses seq + 1, and the last one to finish stores the new value - so a
// launch recorded into a hipGraph advances the protocol on every replay exactly as an eager launch does.
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
  unsigned* seq_state;             // device memory of this rank: {sequence number of the last call, done counter}
};

// next sequence number after s: never 0 (the flags' initial value), parity alternates
__device__ __forceinline__ unsigned p2p_next_seq(unsigned s) { return s == 0xffffffffu ? 2u : s + 1u; }

// last workgroup of a launch to get here publishes the sequence number the launch used
__device__ __forceinline__ void p2p_finish(unsigned* seq_state, unsigned seq) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned ticket = __hip_atomic_fetch_add(seq_state + 1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (ticket == gridDim.x - 1) {
      __hip_atomic_store(seq_state + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(seq_state, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

__global__ void __launch_bounds__(512) p2p_allreduce_oneshot_kernel(P2PArgs a) {
  const int t = threadIdx.x, nt = blockDim.x;
  a.seq = p2p_next_seq(__hip_atomic_load(a.seq_state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
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
  p2p_finish(a.seq_state, a.seq);
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
                                 int slot_elems, unsigned* seq_state, float scale, int* err, int timeout_ms,
                                 hipStream_t st) {
  if (world < 1 || world > P2P_MAX_WORLD || rank < 0 || rank >= world || n < 0 || n > slot_elems ||
      seq_state == nullptr)
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
  a.scale = scale; a.n = n; a.rank = rank; a.world = world; a.slot_elems = slot_elems; a.seq = 0;
  a.seq_state = seq_state;
  const int threads = n >= 512 ? 512 : (n > 64 ? ((n + 63) / 64) * 64 : 64);
  hipLaunchKernelGGL(p2p_allreduce_oneshot_kernel, dim3(1), dim3(threads), 0, st, a);
  return pda_check_launch();
}

// ------------------------------------------------------------------------------------------------
// Two-shot all-reduce: the bandwidth path over the xGMI mesh (direct reduce-scatter + direct all-gather).
//
// A ring all-reduce moves every byte of a GPU through ONE outbound link; on a fully connected 8-GPU mesh
// (7 links per GPU, no switch) the direct algorithm keeps all 7 busy and needs 2 hops instead of 2(N-1):
//   1. scatter : rank r stores shard j of its input straight into slot r of rank j's inbox (one link per
//                peer, all at once); shard r stays where it is;
//   2. reduce  : after the peers' flags arrive, rank r sums the world slots of shard r in fp32 (own shard
//                read from the input) and stores the rounded result into shard r of EVERY rank's result
//                buffer (again one link per peer) - each shard is reduced by exactly one rank, so all ranks
//                end up with identical bits;
//   3. gather  : after the second round of flags the complete result sits in the rank's own region and is
//                copied to the output tensor (a local HBM copy).
// The tensors themselves need no IPC registration: only the per-rank region (inbox + result + flags) is
// peer mapped.  `nblk` workgroups per rank (default 16: a comm kernel must leave the CUs to the backward
// pass it overlaps with) each own one slice of every shard and synchronise only with the same-numbered
// workgroup of the peers, so there is no grid-wide barrier.  Sequence-parity double buffering as in the
// one-shot kernel above; every wait is bounded by wall_clock64 and raises *err instead of hanging.
// (Role in the reference: the NCCL all-reduce of a gradient bucket during `loss.backward()`,
// /root/reference/2.distributed.py:231.)
#define P2P2_MAX_BLOCKS 64

struct P2P2Args {
  char* base[P2P_MAX_WORLD];       // peer-mapped regions
  const void* in;
  void* out;
  int* err;
  long long timeout_ticks;
  long long n;                     // elements in this call
  long long shard;                 // elements per shard (multiple of the 16-byte vector length)
  long long slice;                 // elements per (shard, workgroup) slice (multiple of the vector length)
  long long cap;                   // region capacity in elements (per parity)
  float scale;
  int rank, world;
  unsigned seq;
  unsigned* seq_state;             // as in P2PArgs
};

template <typename T> struct P2P2Layout {
  // region: inbox [2][cap] T (slot p of shard r at p * shard_cap) | result [2][cap] T | flag1 | flag2
  static __device__ __forceinline__ T* inbox(char* base, long long cap, int par) {
    return reinterpret_cast<T*>(base) + (long long)par * cap;
  }
  static __device__ __forceinline__ T* result(char* base, long long cap, int par) {
    return reinterpret_cast<T*>(base) + 2 * cap + (long long)par * cap;
  }
  static __device__ __forceinline__ unsigned* flags(char* base, long long cap, int which, int par, int blk, int world) {
    unsigned* f = reinterpret_cast<unsigned*>(base + 4 * cap * sizeof(T));
    return f + (((long long)which * 2 + par) * P2P2_MAX_BLOCKS + blk) * P2P_MAX_WORLD;
  }
};

template <typename T>
__device__ __forceinline__ void p2p2_copy(T* __restrict__ dst, const T* __restrict__ src, long long len, int t, int nt) {
  constexpr int V = 16 / sizeof(T);
  const long long nv = len / V;
  for (long long i = t; i < nv; i += nt) store16(dst + i * V, load16(src + i * V));
  for (long long i = nv * V + t; i < len; i += nt) dst[i] = src[i];
}

// thread p (< world) signals peer p and then waits for peer p's signal on the own region
template <typename T>
__device__ __forceinline__ void p2p2_handshake(const P2P2Args& a, int which, int par, int blk) {
  const int t = threadIdx.x;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");   // system scope: this thread's peer stores are visible first
  __syncthreads();
  if (t < a.world && t != a.rank) {
    __hip_atomic_store(P2P2Layout<T>::flags(a.base[t], a.cap, which, par, blk, a.world) + a.rank, a.seq,
                       __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned* mine = P2P2Layout<T>::flags(a.base[a.rank], a.cap, which, par, blk, a.world) + t;
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(mine, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != a.seq) {
      if (wall_clock64() - t0 > a.timeout_ticks) {   // every wave leaves: a dead peer cannot hang the GPU
        *a.err = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(8);
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
}

template <typename T>
__global__ void __launch_bounds__(512) p2p_allreduce_twoshot_kernel(P2P2Args a) {
  constexpr int V = 16 / sizeof(T);
  const int t = threadIdx.x, nt = blockDim.x, blk = blockIdx.x;
  a.seq = p2p_next_seq(__hip_atomic_load(a.seq_state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  const int par = (int)(a.seq & 1u);
  const T* __restrict__ in = static_cast<const T*>(a.in);
  T* __restrict__ out = static_cast<T*>(a.out);
  const long long s0 = (long long)blk * a.slice;   // this workgroup's slice inside every shard
  // slice length inside shard j (shards / slices at the end of the tensor may be short or empty)
  auto slice_len = [&](int j) -> long long {
    const long long beg = (long long)j * a.shard + s0;
    long long end = (long long)j * a.shard + (s0 + a.slice < a.shard ? s0 + a.slice : a.shard);
    if (end > a.n) end = a.n;
    return end > beg ? end - beg : 0;
  };
  // 1. scatter: shard j of the input -> slot `rank` of shard j's owner
  for (int k = 1; k < a.world; ++k) {
    const int j = (a.rank + k) % a.world;          // start at the next rank: peers hit different links first
    const long long len = slice_len(j);
    if (len > 0)
      p2p2_copy<T>(P2P2Layout<T>::inbox(a.base[j], a.cap, par) + (long long)a.rank * a.shard + s0,
                   in + (long long)j * a.shard + s0, len, t, nt);
  }
  p2p2_handshake<T>(a, 0, par, blk);
  // 2. reduce the own shard's slice in fp32 and broadcast the rounded result into every rank's result buffer
  {
    const long long len = slice_len(a.rank);
    const long long off = (long long)a.rank * a.shard + s0;
    const T* __restrict__ box = P2P2Layout<T>::inbox(a.base[a.rank], a.cap, par) + s0;
    const long long nv = len / V;
    for (long long i = t; i < nv; i += nt) {
      float acc[V];
      Vec16<T> v = load16(in + off + i * V);
#pragma unroll
      for (int e = 0; e < V; ++e) acc[e] = ElemTraits<T>::to_f32(v.v[e]);
      for (int p = 0; p < a.world; ++p) {
        if (p == a.rank) continue;
        v = load16(box + (long long)p * a.shard + i * V);
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] += ElemTraits<T>::to_f32(v.v[e]);
      }
#pragma unroll
      for (int e = 0; e < V; ++e) v.v[e] = ElemTraits<T>::from_f32(acc[e] * a.scale);
      for (int k = 0; k < a.world; ++k) {
        const int j = (a.rank + k) % a.world;
        store16(P2P2Layout<T>::result(a.base[j], a.cap, par) + off + i * V, v);
      }
    }
    for (long long i = nv * V + t; i < len; i += nt) {
      float acc = ElemTraits<T>::to_f32(in[off + i]);
      for (int p = 0; p < a.world; ++p)
        if (p != a.rank) acc += ElemTraits<T>::to_f32(box[(long long)p * a.shard + i]);
      const T r = ElemTraits<T>::from_f32(acc * a.scale);
      for (int j = 0; j < a.world; ++j) P2P2Layout<T>::result(a.base[j], a.cap, par)[off + i] = r;
    }
  }
  p2p2_handshake<T>(a, 1, par, blk);
  // 3. the reduced tensor is complete in the own result buffer: copy this workgroup's slices out
  const T* __restrict__ res = P2P2Layout<T>::result(a.base[a.rank], a.cap, par);
  for (int j = 0; j < a.world; ++j) {
    const long long len = slice_len(j);
    if (len > 0) p2p2_copy<T>(out + (long long)j * a.shard + s0, res + (long long)j * a.shard + s0, len, t, nt);
  }
  p2p_finish(a.seq_state, a.seq);
}

// Bytes of one rank's region for messages of up to cap_bytes.
PDA_EXPORT long long pda_p2p2_region_bytes(long long cap_bytes) {
  const long long flags = 2LL * 2 * P2P2_MAX_BLOCKS * P2P_MAX_WORLD * sizeof(unsigned);
  return (4 * cap_bytes + flags + 4095) / 4096 * 4096;
}

// bases[p]: region of rank p (pda_p2p2_region_bytes(cap_bytes) each, zero-initialised).  dtype: 0 fp32, 1 fp16,
// 2 bf16.  n * sizeof(T) <= cap_bytes; in/out 16-byte aligned; nblk workgroups (1..64).
PDA_EXPORT int pda_p2p_allreduce_twoshot(void* const* bases, int dtype, const void* in, void* out, long long n,
                                         int rank, int world, long long cap_bytes, unsigned* seq_state, float scale,
                                         int* err, int timeout_ms, int nblk, hipStream_t st) {
  const int esz = dtype == PDA_F32 ? 4 : 2;
  if (world < 1 || world > P2P_MAX_WORLD || rank < 0 || rank >= world || n < 0 || seq_state == nullptr || nblk < 1 ||
      nblk > P2P2_MAX_BLOCKS || dtype < 0 || dtype > 2 || n * esz > cap_bytes || cap_bytes % 16 != 0)
    return -1;
  if ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) return -2;
  if (n == 0) return 0;
  const int V = 16 / esz;
  P2P2Args a;
  for (int p = 0; p < P2P_MAX_WORLD; ++p) a.base[p] = static_cast<char*>(bases[p < world ? p : 0]);
  a.in = in; a.out = out; a.err = err;
  a.timeout_ticks = (long long)timeout_ms * 100000LL;
  a.n = n; a.cap = cap_bytes / esz;
  a.shard = ((n + world - 1) / world + V - 1) / V * V;
  a.slice = ((a.shard + nblk - 1) / nblk + V - 1) / V * V;
  a.scale = scale; a.rank = rank; a.world = world; a.seq = 0;
  a.seq_state = seq_state;
  if (a.shard * world > a.cap) return -3;   // the rounded-up shards must still fit the inbox slots
#define L_P2P2(T) hipLaunchKernelGGL((p2p_allreduce_twoshot_kernel<T>), dim3(nblk), dim3(512), 0, st, a)
  if (dtype == PDA_F32) L_P2P2(float);
  else if (dtype == PDA_F16) L_P2P2(f16);
  else L_P2P2(bf16);
#undef L_P2P2
  return pda_check_launch();
}
