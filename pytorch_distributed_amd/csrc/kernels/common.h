// Shared device helpers for the gfx950 (MI355X / CDNA4) kernel library.
//
// Every kernel in this directory is written for 64-lane wavefronts and is
// exported through a plain C ABI (`extern "C" int pda_*`), taking raw device
// pointers and a hipStream_t, so the Python side can drive it with ctypes on the
// stream PyTorch is currently using (which also makes every launch capturable
// in a hipGraph).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#define PDA_EXPORT extern "C" __attribute__((visibility("default")))

// dtype codes shared with python (ops/_lib.py)
enum PdaDType : int { PDA_F32 = 0, PDA_F16 = 1, PDA_BF16 = 2 };

typedef _Float16 f16;
typedef __bf16 bf16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define WAVE 64

template <typename T> struct ElemTraits;
template <> struct ElemTraits<float> {
  static constexpr int code = PDA_F32;
  static __device__ __forceinline__ float to_f32(float v) { return v; }
  static __device__ __forceinline__ float from_f32(float v) { return v; }
};
template <> struct ElemTraits<f16> {
  static constexpr int code = PDA_F16;
  static __device__ __forceinline__ float to_f32(f16 v) { return (float)v; }
  static __device__ __forceinline__ f16 from_f32(float v) { return (f16)v; }
};
template <> struct ElemTraits<bf16> {
  static constexpr int code = PDA_BF16;
  static __device__ __forceinline__ float to_f32(bf16 v) { return (float)v; }
  static __device__ __forceinline__ bf16 from_f32(float v) { return (bf16)v; }
};

// 16-byte vector of T (the unit every memory-bound kernel moves per lane).
template <typename T> struct Vec16 {
  static constexpr int N = 16 / sizeof(T);
  T v[N];
};
template <typename T>
__device__ __forceinline__ Vec16<T> load16(const T* p) {
  Vec16<T> r;
  *reinterpret_cast<u32x4*>(&r) = *reinterpret_cast<const u32x4*>(p);
  return r;
}
template <typename T>
__device__ __forceinline__ void store16(T* p, const Vec16<T>& v) {
  *reinterpret_cast<u32x4*>(p) = *reinterpret_cast<const u32x4*>(&v);
}

// Streaming forms for tensors far larger than the caches that are written once and read back much later, or
// read exactly once (activations and gradients of a layer at training batch sizes): non-temporal accesses do
// not displace the operands other kernels re-read from L2.  Measured on the ResNet-50 step (batch 400): the
// loads of the BatchNorm passes 2-3 %, the stores below 1 %.
template <typename T>
__device__ __forceinline__ void store16_stream(T* p, const Vec16<T>& v) {
  __builtin_nontemporal_store(*reinterpret_cast<const u32x4*>(&v), reinterpret_cast<u32x4*>(p));
}
// Outputs smaller than this stay on the default cache policy: the kernel that reads them next finds them in
// L2 / the memory-side cache (PDA_STREAM_MIN_MB, default 64).
static inline long long pda_stream_min_bytes() {
  static const long long v = [] {
    const char* e = getenv("PDA_STREAM_MIN_MB");
    return (e ? atoll(e) : 64LL) << 20;
  }();
  return v;
}
template <typename T>
__device__ __forceinline__ void store16_maybe_stream(T* p, const Vec16<T>& v, int stream) {
  if (stream) store16_stream(p, v);
  else store16(p, v);
}
template <typename T>
__device__ __forceinline__ Vec16<T> load16_stream(const T* p) {
  Vec16<T> r;
  *reinterpret_cast<u32x4*>(&r) = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
  return r;
}

// N-element vector of T: one 16-byte access when N*sizeof(T)==16, one 4-byte access when N*sizeof(T)==4 (an even
// count of 16-bit channels that does not fill 16-byte vectors), element-wise otherwise (N=1 is the
// any-channel-count fallback of the pooling kernels).
template <typename T, int N> struct VecN { T v[N]; };
template <typename T, int N>
__device__ __forceinline__ VecN<T, N> loadN(const T* p) {
  VecN<T, N> r;
  if constexpr (N * sizeof(T) == 16) {
    *reinterpret_cast<u32x4*>(&r) = *reinterpret_cast<const u32x4*>(p);
  } else if constexpr (N * sizeof(T) == 4 && N > 1) {
    *reinterpret_cast<unsigned*>(&r) = *reinterpret_cast<const unsigned*>(p);
  } else {
#pragma unroll
    for (int i = 0; i < N; ++i) r.v[i] = p[i];
  }
  return r;
}
template <typename T, int N>
__device__ __forceinline__ void storeN(T* p, const VecN<T, N>& v) {
  if constexpr (N * sizeof(T) == 16) {
    *reinterpret_cast<u32x4*>(p) = *reinterpret_cast<const u32x4*>(&v);
  } else if constexpr (N * sizeof(T) == 4 && N > 1) {
    *reinterpret_cast<unsigned*>(p) = *reinterpret_cast<const unsigned*>(&v);
  } else {
#pragma unroll
    for (int i = 0; i < N; ++i) p[i] = v.v[i];
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Block-wide sum for blocks of up to 1024 threads; `scratch` needs 16 floats.
__device__ __forceinline__ float block_sum(float v, float* scratch) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) scratch[wid] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < nw; ++i) r += scratch[i];
  return r;
}

static inline int pda_check_launch() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

// Bijective XCD-aware remap of a linear workgroup id: workgroups are dealt
// round-robin over the 8 XCDs, so id%8 labels the XCD; give every XCD a
// contiguous run of tiles so neighbouring tiles share one L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + (bid >> 3);
}
