// Memory-bound element-wise kernels:
//  * input pipeline: NCHW (uint8 / fp32 / fp16) -> NHWC compute dtype with the
//    per-channel (x - mean) / std of the loader fused in, optional channel
//    padding (the ResNet stem wants 4 channels) and an optional zero border
//    (reference: `images.cuda(non_blocking=True)` + the data_prefetcher's
//    `.float().sub_(mean).div_(std)`, /root/reference/4.apex_distributed.py:119-122);
//  * stand-alone ReLU fwd/bwd, add, per-channel bias add (MNIST net);
//  * dropout / channel dropout with a counter-based Philox stream, so the mask
//    is regenerated in backward instead of stored
//    (reference: nn.Dropout2d / F.dropout, /root/reference/5.2.horovod_pytorch_mnist.py:41,47,50).
#include "common.h"

template <typename TI, typename TO>
__global__ void __launch_bounds__(256)
nchw_to_nhwc_kernel(const TI* __restrict__ in, TO* __restrict__ out, int N, int C, int H, int W, int Cp, int Hp,
                    int Wp, int oh, int ow, const float* __restrict__ mean, const float* __restrict__ inv_std,
                    float pre_scale) {
  // one thread per output pixel (all Cp channels): reads are strided by H*W but
  // consecutive threads read consecutive w, so each channel plane is coalesced.
  const long long total = (long long)N * Hp * Wp;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int wp = (int)(i % Wp);
    long long r = i / Wp;
    const int hp = (int)(r % Hp);
    const int n = (int)(r / Hp);
    const int h = hp - oh, w = wp - ow;
    const bool inside = (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;
    TO* o = out + i * Cp;
    for (int c = 0; c < Cp; ++c) {
      float v = 0.f;
      if (inside && c < C) {
        v = (float)in[(((long long)n * C + c) * H + h) * W + w] * pre_scale;
        if (mean) v = (v - mean[c]) * inv_std[c];
      }
      o[c] = ElemTraits<TO>::from_f32(v);
    }
  }
}

// ReLU and its clipped form (ReLU6): y = min(max(x, 0), cap), cap = +inf for the plain one; the gradient
// passes where 0 < y < cap (torch.nn.functional.hardtanh's rule at the upper end).  One 16-byte vector per
// lane; the last n % (16 / sizeof(T)) elements are handled one by one by the first lanes.
template <typename T>
__global__ void __launch_bounds__(256)
relu_cap_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, long long n, float cap, int vec) {
  constexpr int EPC = Vec16<T>::N;
  const long long nvec = vec ? n / EPC : 0;   // vec == 0: a pointer is not 16-byte aligned
  const long long tid = blockIdx.x * (long long)blockDim.x + threadIdx.x, nthr = (long long)gridDim.x * blockDim.x;
  for (long long i = tid; i < nvec; i += nthr) {
    Vec16<T> v = load16(x + i * EPC);
#pragma unroll
    for (int e = 0; e < EPC; ++e) v.v[e] = ElemTraits<T>::from_f32(fminf(fmaxf(ElemTraits<T>::to_f32(v.v[e]), 0.f), cap));
    store16(y + i * EPC, v);
  }
  for (long long i = nvec * EPC + tid; i < n; i += nthr)
    y[i] = ElemTraits<T>::from_f32(fminf(fmaxf(ElemTraits<T>::to_f32(x[i]), 0.f), cap));
}
template <typename T>
__global__ void __launch_bounds__(256)
relu_cap_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y, T* __restrict__ dx, long long n, float cap,
                    int vec) {
  constexpr int EPC = Vec16<T>::N;
  const long long nvec = vec ? n / EPC : 0;
  const long long tid = blockIdx.x * (long long)blockDim.x + threadIdx.x, nthr = (long long)gridDim.x * blockDim.x;
  for (long long i = tid; i < nvec; i += nthr) {
    Vec16<T> g = load16(dy + i * EPC);
    const Vec16<T> v = load16(y + i * EPC);
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      const float f = ElemTraits<T>::to_f32(v.v[e]);
      if (!(f > 0.f && f < cap)) g.v[e] = ElemTraits<T>::from_f32(0.f);
    }
    store16(dx + i * EPC, g);
  }
  for (long long i = nvec * EPC + tid; i < n; i += nthr) {
    const float f = ElemTraits<T>::to_f32(y[i]);
    dx[i] = (f > 0.f && f < cap) ? dy[i] : ElemTraits<T>::from_f32(0.f);
  }
}
template <typename T>
__global__ void __launch_bounds__(256)
add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ o, long long n, int relu) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float v = ElemTraits<T>::to_f32(a[i]) + ElemTraits<T>::to_f32(b[i]);
    if (relu) v = fmaxf(v, 0.f);
    o[i] = ElemTraits<T>::from_f32(v);
  }
}
// out = sum of up to 8 same-sized buffers in one pass (DataParallel's reduce-add of per-replica
// gradients: one read of every replica's flat gradient instead of N-1 read-modify-write passes).
struct AddNArgs {
  const void* src[8];
  int n;
};
template <typename T>
__global__ void __launch_bounds__(256)
add_n_kernel(AddNArgs a, T* __restrict__ o, long long nvec, long long count) {
  constexpr int EPC = 16 / sizeof(T);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nvec; i += (long long)gridDim.x * blockDim.x) {
    float acc[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (s < a.n) {
        const Vec16<T> v = load16(reinterpret_cast<const T*>(a.src[s]) + i * EPC);
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] += ElemTraits<T>::to_f32(v.v[e]);
      }
    }
    Vec16<T> r;
#pragma unroll
    for (int e = 0; e < EPC; ++e) r.v[e] = ElemTraits<T>::from_f32(acc[e]);
    store16(o + i * EPC, r);
  }
  // tail (count not a multiple of the vector width): first workgroup, scalar
  if (blockIdx.x == 0) {
    for (long long i = nvec * EPC + threadIdx.x; i < count; i += blockDim.x) {
      float acc = 0.f;
      for (int s = 0; s < a.n; ++s) acc += ElemTraits<T>::to_f32(reinterpret_cast<const T*>(a.src[s])[i]);
      o[i] = ElemTraits<T>::from_f32(acc);
    }
  }
}

template <typename T>
__global__ void __launch_bounds__(256)
bias_add_kernel(const T* __restrict__ x, const float* __restrict__ bias, T* __restrict__ y, long long n, int C,
                int relu) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float v = ElemTraits<T>::to_f32(x[i]) + bias[i % C];
    if (relu) v = fmaxf(v, 0.f);
    y[i] = ElemTraits<T>::from_f32(v);
  }
}
// C a multiple of the 16-byte vector and 16-byte aligned pointers: one vector per lane, its channel offset
// from a 32-bit remainder (vectors per row = C / EPC)
template <typename T>
__global__ void __launch_bounds__(256)
bias_add_vec_kernel(const T* __restrict__ x, const float* __restrict__ bias, T* __restrict__ y, long long nvec,
                    int vpr, int relu) {
  constexpr int EPC = Vec16<T>::N;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nvec;
       i += (long long)gridDim.x * blockDim.x) {
    const int c0 = (int)(i % vpr) * EPC;
    Vec16<T> v = load16(x + i * EPC);
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      float f = ElemTraits<T>::to_f32(v.v[e]) + bias[c0 + e];
      if (relu) f = fmaxf(f, 0.f);
      v.v[e] = ElemTraits<T>::from_f32(f);
    }
    store16(y + i * EPC, v);
  }
}
// dbias[c] = sum over rows of dy[m][c]   (any C; grid.x = 64-channel groups, grid.y = row slabs).
// One slab: plain store.  Several slabs (a first-layer conv has millions of rows and 64 channels - one
// workgroup per 64 channels would leave the chip idle): float atomics into the zero-filled output.
template <typename T>
__global__ void __launch_bounds__(256)
colsum_kernel(const T* __restrict__ dy, float* __restrict__ out, long long M, int C, long long slab) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int rl = threadIdx.x >> 6;
  const long long m0 = (long long)blockIdx.y * slab;
  const long long m1 = m0 + slab < M ? m0 + slab : M;
  float acc = 0.f;
  if (c < C)
    for (long long m = m0 + rl; m < m1; m += 4) acc += ElemTraits<T>::to_f32(dy[m * C + c]);
  red[rl][threadIdx.x & 63] = acc;
  __syncthreads();
  if (rl == 0 && c < C) {
    const float t = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    if (gridDim.y > 1) atomicAdd(out + c, t);
    else out[c] = t;
  }
}

// ---- Philox4x32-10 ---------------------------------------------------------
struct Philox {
  unsigned int k0, k1;
  __device__ Philox(unsigned long long seed) : k0((unsigned)seed), k1((unsigned)(seed >> 32)) {}
  __device__ void operator()(unsigned long long ctr_lo, unsigned long long ctr_hi, unsigned int (&out)[4]) const {
    unsigned int c0 = (unsigned)ctr_lo, c1 = (unsigned)(ctr_lo >> 32), c2 = (unsigned)ctr_hi, c3 = (unsigned)(ctr_hi >> 32);
    unsigned int a = k0, b = k1;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
      const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
      const unsigned int n0 = (unsigned)(p1 >> 32) ^ c1 ^ a;
      const unsigned int n1 = (unsigned)p1;
      const unsigned int n2 = (unsigned)(p0 >> 32) ^ c3 ^ b;
      const unsigned int n3 = (unsigned)p0;
      c0 = n0; c1 = n1; c2 = n2; c3 = n3;
      a += 0x9E3779B9u; b += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
  }
};

// y = x * mask / (1-p); mask element index = i (element mode) or (n, c) (channel
// mode, NHWC: every pixel of one image shares the channel's mask).  The same
// (seed, offset) regenerates the mask for backward.
template <typename T>
__global__ void __launch_bounds__(256)
dropout_kernel(const T* __restrict__ x, T* __restrict__ y, long long n, float p, unsigned long long seed,
               unsigned long long offset, int channel_mode, long long per_image, int C,
               const unsigned long long* __restrict__ replay) {
  // `replay`: a device-resident counter added to the host's offset.  A step recorded into a hipGraph has the host
  // offset baked in; the counter, advanced by the graph itself once per replay, gives every replay its own masks
  // (and the backward of that replay the same ones as its forward).
  if (replay != nullptr) offset += *replay;
  Philox rng(seed);
  const float keep_scale = p < 1.f ? 1.f / (1.f - p) : 0.f;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const long long id = channel_mode ? (i / per_image) * C + (i % C) : i;
    unsigned int r[4];
    rng((unsigned long long)(id >> 2), offset, r);
    const float u = (float)(r[id & 3] >> 8) * (1.0f / 16777216.0f);
    const float v = u >= p ? ElemTraits<T>::to_f32(x[i]) * keep_scale : 0.f;
    y[i] = ElemTraits<T>::from_f32(v);
  }
}

static inline int ew_blocks(long long n) {
  long long b = (n + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}
#define PDA_DISPATCH_T(dtype, CALL)                 \
  switch (dtype) {                                  \
    case PDA_F32: { typedef float T; CALL; } break; \
    case PDA_F16: { typedef f16 T; CALL; } break;   \
    case PDA_BF16: { typedef bf16 T; CALL; } break; \
    default: return -6;                             \
  }

// dt_in: 0 fp32, 1 fp16, 3 uint8
PDA_EXPORT int pda_nchw_to_nhwc(int dt_in, int dt_out, const void* in, void* out, int N, int C, int H, int W, int Cp,
                                int Hp, int Wp, int oh, int ow, const float* mean, const float* inv_std,
                                float pre_scale, hipStream_t st) {
  const long long total = (long long)N * Hp * Wp;
  if (total == 0) return 0;
  const int nb = ew_blocks(total);
#define L_CVT(TI) \
  PDA_DISPATCH_T(dt_out, hipLaunchKernelGGL((nchw_to_nhwc_kernel<TI, T>), dim3(nb), dim3(256), 0, st, (const TI*)in, \
                                            (T*)out, N, C, H, W, Cp, Hp, Wp, oh, ow, mean, inv_std, pre_scale))
  if (dt_in == 0) { L_CVT(float); }
  else if (dt_in == 1) { L_CVT(f16); }
  else if (dt_in == 3) { L_CVT(unsigned char); }
  else return -6;
#undef L_CVT
  return pda_check_launch();
}

static inline bool ew_aligned16(const void* a, const void* b, const void* c) {
  return ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c)) & 15) == 0;
}
PDA_EXPORT int pda_relu_cap_fwd(int dtype, const void* x, void* y, long long n, float cap, hipStream_t st) {
  if (n == 0) return 0;
  const int vec = ew_aligned16(x, y, nullptr);
  PDA_DISPATCH_T(dtype, hipLaunchKernelGGL((relu_cap_fwd_kernel<T>), dim3(ew_blocks(n / 4 + 1)), dim3(256), 0, st, (const T*)x, (T*)y, n, cap, vec));
  return pda_check_launch();
}
PDA_EXPORT int pda_relu_cap_bwd(int dtype, const void* dy, const void* y, void* dx, long long n, float cap,
                                hipStream_t st) {
  if (n == 0) return 0;
  const int vec = ew_aligned16(dy, y, dx);
  PDA_DISPATCH_T(dtype, hipLaunchKernelGGL((relu_cap_bwd_kernel<T>), dim3(ew_blocks(n / 4 + 1)), dim3(256), 0, st, (const T*)dy, (const T*)y, (T*)dx, n, cap, vec));
  return pda_check_launch();
}
PDA_EXPORT int pda_relu_fwd(int dtype, const void* x, void* y, long long n, hipStream_t st) {
  return pda_relu_cap_fwd(dtype, x, y, n, INFINITY, st);
}
PDA_EXPORT int pda_relu_bwd(int dtype, const void* dy, const void* y, void* dx, long long n, hipStream_t st) {
  return pda_relu_cap_bwd(dtype, dy, y, dx, n, INFINITY, st);
}
// srcs: array of nsrc (<= 8) device pointers, all 16-byte aligned like out; out may alias srcs[0].
PDA_EXPORT int pda_add_n(int dtype, const void* const* srcs, int nsrc, void* out, long long count, hipStream_t st) {
  if (nsrc < 1 || nsrc > 8) return -1;
  if (count <= 0) return 0;
  AddNArgs a;
  a.n = nsrc;
  for (int i = 0; i < 8; ++i) {
    a.src[i] = i < nsrc ? srcs[i] : nullptr;
    if (i < nsrc && ((uintptr_t)srcs[i] & 15)) return -5;
  }
  if ((uintptr_t)out & 15) return -5;
  const int es = dtype == PDA_F32 ? 4 : 2;
  const long long nvec = count / (16 / es);
  PDA_DISPATCH_T(dtype, hipLaunchKernelGGL((add_n_kernel<T>), dim3(ew_blocks(nvec > 0 ? nvec : 1)), dim3(256), 0, st, a,
                                           (T*)out, nvec, count));
  return pda_check_launch();
}

PDA_EXPORT int pda_add(int dtype, const void* a, const void* b, void* o, long long n, int relu, hipStream_t st) {
  if (n == 0) return 0;
  PDA_DISPATCH_T(dtype, hipLaunchKernelGGL((add_kernel<T>), dim3(ew_blocks(n)), dim3(256), 0, st, (const T*)a, (const T*)b, (T*)o, n, relu));
  return pda_check_launch();
}
PDA_EXPORT int pda_bias_add(int dtype, const void* x, const float* bias, void* y, long long n, int C, int relu,
                            hipStream_t st) {
  if (n == 0) return 0;
  const int epc = dtype == PDA_F32 ? 4 : 8;
  if (C % epc == 0 && n % C == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0) {
    const long long nvec = n / epc;
    PDA_DISPATCH_T(dtype, hipLaunchKernelGGL((bias_add_vec_kernel<T>), dim3(ew_blocks(nvec)), dim3(256), 0, st, (const T*)x, bias, (T*)y, nvec, C / epc, relu));
    return pda_check_launch();
  }
  PDA_DISPATCH_T(dtype, hipLaunchKernelGGL((bias_add_kernel<T>), dim3(ew_blocks(n)), dim3(256), 0, st, (const T*)x, bias, (T*)y, n, C, relu));
  return pda_check_launch();
}
// one_slab != 0: a single workgroup per 64 channels sums all rows (no atomics: bit-reproducible).
PDA_EXPORT int pda_colsum(int dtype, const void* dy, float* out, long long M, int C, int one_slab, hipStream_t st) {
  if (C == 0) return 0;
  const int gx = (C + 63) / 64;
  long long ns = 1;
  if (!one_slab) {
    ns = 2048 / gx;                           // ~8 workgroups per CU
    const long long cap = (M + 1023) / 1024;  // at least 256 rows per row lane
    if (ns > cap) ns = cap;
    if (ns < 1) ns = 1;
    if (ns > 65535) ns = 65535;
  }
  const long long slab = M > 0 ? (M + ns - 1) / ns : 1;
  const unsigned gy = M > 0 ? (unsigned)((M + slab - 1) / slab) : 1u;
  if (gy > 1) {
    hipError_t e = hipMemsetAsync(out, 0, sizeof(float) * (size_t)C, st);
    if (e != hipSuccess) return (int)e;
  }
  PDA_DISPATCH_T(dtype, hipLaunchKernelGGL((colsum_kernel<T>), dim3(gx, gy), dim3(256), 0, st, (const T*)dy, out, M, C, slab));
  return pda_check_launch();
}
PDA_EXPORT int pda_dropout(int dtype, const void* x, void* y, long long n, float p, unsigned long long seed,
                           unsigned long long offset, int channel_mode, long long per_image, int C,
                           const unsigned long long* replay, hipStream_t st) {
  if (n == 0) return 0;
  PDA_DISPATCH_T(dtype, hipLaunchKernelGGL((dropout_kernel<T>), dim3(ew_blocks(n)), dim3(256), 0, st, (const T*)x, (T*)y, n, p, seed, offset, channel_mode, per_image, C, replay));
  return pda_check_launch();
}
