// Single-process multi-GPU copies for DataParallel over xGMI peer access (the role of
// torch.cuda.comm.broadcast_coalesced / reduce_add_coalesced behind nn.DataParallel,
// /root/reference/1.dataparallel.py:109,222-233).
//
//   peer_bcast : ONE kernel on the source device reads the flat parameter buffer once from its HBM and stores
//                every 16-byte vector into up to 8 destination buffers on other devices - all outbound xGMI
//                links of the source GPU busy at once, instead of one host-driven copy per replica that
//                re-reads the source each time.
//   reduce-add : runs on the target device as `add_n` (elementwise.hip) with the replicas' gradient buffers
//                passed as peer pointers: every source is read once, straight over its link, with no staging
//                copy on the target.
// pda_peer_enable turns on peer access for an (accessing device, owner device) pair.
#include "common.h"

struct PeerBcastArgs {
  void* dst[8];
  int n;
};

__global__ void __launch_bounds__(256)
peer_bcast_kernel(const u32x4* __restrict__ src, PeerBcastArgs a, long long nvec, const unsigned char* __restrict__ srcb,
                  long long bytes) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nvec; i += (long long)gridDim.x * blockDim.x) {
    const u32x4 v = src[i];
#pragma unroll
    for (int d = 0; d < 8; ++d)
      if (d < a.n) reinterpret_cast<u32x4*>(a.dst[d])[i] = v;
  }
  if (blockIdx.x == 0) {  // tail bytes
    for (long long i = nvec * 16 + threadIdx.x; i < bytes; i += blockDim.x) {
      const unsigned char b = srcb[i];
      for (int d = 0; d < a.n; ++d) reinterpret_cast<unsigned char*>(a.dst[d])[i] = b;
    }
  }
}

// src and every dst 16-byte aligned; launched on `st`, a stream of the source device.
PDA_EXPORT int pda_peer_bcast(const void* src, void* const* dsts, int ndst, long long bytes, hipStream_t st) {
  if (ndst < 0 || ndst > 8 || bytes < 0) return -1;
  if (ndst == 0 || bytes == 0) return 0;
  if ((uintptr_t)src & 15) return -5;
  PeerBcastArgs a;
  a.n = ndst;
  for (int i = 0; i < 8; ++i) {
    a.dst[i] = i < ndst ? dsts[i] : nullptr;
    if (i < ndst && ((uintptr_t)dsts[i] & 15)) return -5;
  }
  const long long nvec = bytes / 16;
  long long nb = (nvec + 255) / 256;
  if (nb > 2048) nb = 2048;   // a copy kernel: enough waves to keep the links busy, not the whole chip
  if (nb < 1) nb = 1;
  hipLaunchKernelGGL(peer_bcast_kernel, dim3((unsigned)nb), dim3(256), 0, st, static_cast<const u32x4*>(src), a, nvec,
                     static_cast<const unsigned char*>(src), bytes);
  return pda_check_launch();
}

// 0 = `dev` can now address `peer`'s memory (or they are the same device), 1 = the pair has no peer path.
PDA_EXPORT int pda_peer_enable(int dev, int peer) {
  if (dev == peer) return 0;
  int can = 0;
  if (hipDeviceCanAccessPeer(&can, dev, peer) != hipSuccess || !can) return 1;
  int prev = 0;
  if (hipGetDevice(&prev) != hipSuccess) return -1;
  if (hipSetDevice(dev) != hipSuccess) return -1;
  const hipError_t e = hipDeviceEnablePeerAccess(peer, 0);
  (void)hipSetDevice(prev);
  if (e == hipSuccess) return 0;
  if (e == hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); return 0; }
  return (int)e;
}
