// Multi-tensor-apply kernels: one launch covers a whole list of tensors
// (the reference reaches apex's
// amp_C.multi_tensor_scale / multi_tensor_axpby / multi_tensor_l2norm through
// `amp.scale_loss`, /root/reference/4.apex_distributed2.py:289-291, and
// torch.optim.SGD.step through `optimizer.step()`, /root/reference/2.distributed.py:232).
//
// Layout of the device tables (built once per tensor list by ops/multi_tensor.py):
//   meta : int64 [ntensors][L+1]  = L data pointers (one per list) + numel
//   blk  : int32 [nblocks][2]     = (tensor index, chunk index), chunk = 32768 elements
// Every block walks one chunk with 4-element vectors (16 B of fp32 / 8 B of
// fp16) when all of the tensor's pointers allow it.
//
//   scale : out = in * scale, sets *noop_flag if any input is inf/nan or a value
//           leaves the range of a narrower output type (grad unscale + overflow
//           check in one pass; with scale=1 it is the cast / flatten / unflatten
//           kernel: out pointers may be views of a bucket).  Bit 1 of `invert`
//           makes it a pure check: nothing is stored, so the pass cannot write a
//           stale value over a gradient that another stream is replacing with its
//           all-reduced form
//   axpby : out = a*x + b*y with the same finite check
//   l2norm: per-tensor and global sum of squares
//   sgd   : unscale + overflow-skip + weight decay + momentum (+nesterov) + update
//           of the fp32 master weight + optional half-precision model copy
#include "common.h"

#define MT_CHUNK 32768

template <typename T> __device__ __forceinline__ void ld4(const T* p, long long i, long long n, bool al, float (&v)[4]) {
  if (al && i + 4 <= n) {
    if constexpr (sizeof(T) == 4) {
      f32x4 t = *reinterpret_cast<const f32x4*>(p + i);
      v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
    } else {
      struct alignas(8) P { T e[4]; };
      P t = *reinterpret_cast<const P*>(p + i);
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = ElemTraits<T>::to_f32(t.e[k]);
    }
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = (i + k < n) ? ElemTraits<T>::to_f32(p[i + k]) : 0.f;
  }
}
template <typename T> __device__ __forceinline__ void st4(T* p, long long i, long long n, bool al, const float (&v)[4]) {
  if (al && i + 4 <= n) {
    if constexpr (sizeof(T) == 4) {
      f32x4 t = {v[0], v[1], v[2], v[3]};
      *reinterpret_cast<f32x4*>(p + i) = t;
    } else {
      struct alignas(8) P { T e[4]; };
      P t;
#pragma unroll
      for (int k = 0; k < 4; ++k) t.e[k] = ElemTraits<T>::from_f32(v[k]);
      *reinterpret_cast<P*>(p + i) = t;
    }
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (i + k < n) p[i + k] = ElemTraits<T>::from_f32(v[k]);
  }
}
template <typename T> __device__ __forceinline__ bool al4(const void* p) {
  return (reinterpret_cast<uintptr_t>(p) & (sizeof(T) * 4 - 1)) == 0;
}

template <typename TI, typename TO>
__global__ void __launch_bounds__(256)
mt_scale_kernel(const long long* __restrict__ meta, const int* __restrict__ blk, float scale,
                const float* __restrict__ dev_scale, int invert, int* __restrict__ noop_flag) {
  const int ti = blk[blockIdx.x * 2], ci = blk[blockIdx.x * 2 + 1];
  const TI* in = reinterpret_cast<const TI*>(meta[ti * 3 + 0]);
  TO* out = reinterpret_cast<TO*>(meta[ti * 3 + 1]);
  const long long n = meta[ti * 3 + 2];
  float sc = scale;
  const bool check_only = (invert & 2) != 0;
  if (dev_scale) sc *= (invert & 1) ? 1.f / *dev_scale : *dev_scale;
  const bool al = al4<TI>(in) && al4<TO>(out);
  const long long beg = (long long)ci * MT_CHUNK;
  const long long end = beg + MT_CHUNK < n ? beg + MT_CHUNK : n;
  bool bad = false;
  for (long long i = beg + threadIdx.x * 4; i < end; i += 256 * 4) {
    float v[4];
    ld4<TI>(in, i, end, al, v);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v[k] *= sc;
      // a finite fp32 value beyond the fp16 range becomes inf on the store: flag that as well
      bad = bad || !isfinite(sizeof(TO) < sizeof(TI) ? ElemTraits<TO>::to_f32(ElemTraits<TO>::from_f32(v[k])) : v[k]);
    }
    if (!check_only) st4<TO>(out, i, end, al, v);
  }
  if (noop_flag && __any(bad)) {
    if ((threadIdx.x & 63) == 0) *noop_flag = 1;
  }
}

template <typename TX, typename TY, typename TO>
__global__ void __launch_bounds__(256)
mt_axpby_kernel(const long long* __restrict__ meta, const int* __restrict__ blk, float a, float b,
                int* __restrict__ noop_flag) {
  const int ti = blk[blockIdx.x * 2], ci = blk[blockIdx.x * 2 + 1];
  const TX* x = reinterpret_cast<const TX*>(meta[ti * 4 + 0]);
  const TY* y = reinterpret_cast<const TY*>(meta[ti * 4 + 1]);
  TO* out = reinterpret_cast<TO*>(meta[ti * 4 + 2]);
  const long long n = meta[ti * 4 + 3];
  const bool al = al4<TX>(x) && al4<TY>(y) && al4<TO>(out);
  const long long beg = (long long)ci * MT_CHUNK;
  const long long end = beg + MT_CHUNK < n ? beg + MT_CHUNK : n;
  bool bad = false;
  for (long long i = beg + threadIdx.x * 4; i < end; i += 256 * 4) {
    float vx[4], vy[4], vo[4];
    ld4<TX>(x, i, end, al, vx);
    ld4<TY>(y, i, end, al, vy);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      bad = bad || !isfinite(vx[k]) || !isfinite(vy[k]);
      vo[k] = a * vx[k] + b * vy[k];
    }
    st4<TO>(out, i, end, al, vo);
  }
  if (noop_flag && __any(bad)) {
    if ((threadIdx.x & 63) == 0) *noop_flag = 1;
  }
}

// sumsq[0] += global sum of squares; sumsq[1 + ti] += per-tensor sum of squares
template <typename T>
__global__ void __launch_bounds__(256)
mt_l2norm_kernel(const long long* __restrict__ meta, const int* __restrict__ blk, float* __restrict__ sumsq,
                 int per_tensor) {
  __shared__ float scratch[16];
  const int ti = blk[blockIdx.x * 2], ci = blk[blockIdx.x * 2 + 1];
  const T* x = reinterpret_cast<const T*>(meta[ti * 2 + 0]);
  const long long n = meta[ti * 2 + 1];
  const bool al = al4<T>(x);
  const long long beg = (long long)ci * MT_CHUNK;
  const long long end = beg + MT_CHUNK < n ? beg + MT_CHUNK : n;
  float acc = 0.f;
  for (long long i = beg + threadIdx.x * 4; i < end; i += 256 * 4) {
    float v[4];
    ld4<T>(x, i, end, al, v);
#pragma unroll
    for (int k = 0; k < 4; ++k) acc += v[k] * v[k];
  }
  const float tot = block_sum(acc, scratch);
  if (threadIdx.x == 0) {
    atomicAdd(sumsq, tot);
    if (per_tensor) atomicAdd(sumsq + 1 + ti, tot);
  }
}

__global__ void sqrt_inplace_kernel(float* x, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = sqrtf(x[i]);
}

// hp = {lr, momentum, dampening, weight_decay, nesterov}
// lists: grad (TG), param fp32, momentum buffer fp32, [model copy TM]
template <typename TG, typename TM, bool HAS_MODEL>
__global__ void __launch_bounds__(256)
mt_sgd_kernel(const long long* __restrict__ meta, const int* __restrict__ blk, const float* __restrict__ hp,
              const float* __restrict__ dev_scale, const int* __restrict__ noop_flag, int first_run, float gscale) {
  if (noop_flag && *noop_flag) return;  // overflow somewhere: skip the whole step
  constexpr int L = HAS_MODEL ? 4 : 3;
  const int ti = blk[blockIdx.x * 2], ci = blk[blockIdx.x * 2 + 1];
  const TG* g = reinterpret_cast<const TG*>(meta[ti * (L + 1) + 0]);
  float* p = reinterpret_cast<float*>(meta[ti * (L + 1) + 1]);
  float* mbuf = reinterpret_cast<float*>(meta[ti * (L + 1) + 2]);
  TM* pm = HAS_MODEL ? reinterpret_cast<TM*>(meta[ti * (L + 1) + 3]) : nullptr;
  const long long n = meta[ti * (L + 1) + L];
  const float lr = hp[0], mom = hp[1], damp = hp[2], wd = hp[3];
  const bool nesterov = hp[4] != 0.f;
  // gscale: what is still owed to the gradient besides the loss scale (1/world of a SUM all-reduce
  // whose result is consumed straight from the wire buffer)
  const float inv = dev_scale ? gscale / *dev_scale : gscale;
  bool al = al4<TG>(g) && al4<float>(p) && al4<float>(mbuf);
  if (HAS_MODEL) al = al && al4<TM>(pm);
  const long long beg = (long long)ci * MT_CHUNK;
  const long long end = beg + MT_CHUNK < n ? beg + MT_CHUNK : n;
  for (long long i = beg + threadIdx.x * 4; i < end; i += 256 * 4) {
    float vg[4], vp[4], vm[4];
    ld4<TG>(g, i, end, al, vg);
    ld4<float>(p, i, end, al, vp);
    if (mom != 0.f && !first_run) ld4<float>(mbuf, i, end, al, vm);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float gr = vg[k] * inv + wd * vp[k];
      if (mom != 0.f) {
        vm[k] = first_run ? gr : mom * vm[k] + (1.f - damp) * gr;
        gr = nesterov ? gr + mom * vm[k] : vm[k];
      }
      vp[k] -= lr * gr;
    }
    st4<float>(p, i, end, al, vp);
    if (mom != 0.f) st4<float>(mbuf, i, end, al, vm);
    if (HAS_MODEL) st4<TM>(pm, i, end, al, vp);
  }
}

// Dynamic loss-scale state machine, on device: state = {scale,
// growth_tracker, skipped_steps, last_overflow}.  Halve on overflow, double after
// `interval` clean steps, clamp to [min_scale, max_scale]; clears the flag.
__global__ void scaler_update_kernel(float* __restrict__ state, int* __restrict__ noop_flag, float growth,
                                     float backoff, int interval, float min_scale, float max_scale) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const int overflow = *noop_flag;
  float scale = state[0];
  float tracker = state[1];
  if (overflow) {
    scale = fmaxf(scale * backoff, min_scale);
    tracker = 0.f;
    state[2] += 1.f;
  } else {
    tracker += 1.f;
    if (tracker >= (float)interval) {
      scale = fminf(scale * growth, max_scale);
      tracker = 0.f;
    }
  }
  state[0] = scale;
  state[1] = tracker;
  state[3] = overflow ? 1.f : 0.f;
  *noop_flag = 0;
}

// ------------------------------------------------------------- host side ----
#define MT_SWITCH2(di, dout, NAME, ...)                                                     \
  if (di == PDA_F32 && dout == PDA_F32) { NAME(float, float, __VA_ARGS__); }                  \
  else if (di == PDA_F32 && dout == PDA_F16) { NAME(float, f16, __VA_ARGS__); }               \
  else if (di == PDA_F32 && dout == PDA_BF16) { NAME(float, bf16, __VA_ARGS__); }             \
  else if (di == PDA_F16 && dout == PDA_F32) { NAME(f16, float, __VA_ARGS__); }               \
  else if (di == PDA_BF16 && dout == PDA_F32) { NAME(bf16, float, __VA_ARGS__); }             \
  else if (di == PDA_F16 && dout == PDA_F16) { NAME(f16, f16, __VA_ARGS__); }                 \
  else if (di == PDA_BF16 && dout == PDA_BF16) { NAME(bf16, bf16, __VA_ARGS__); }             \
  else return -6;

PDA_EXPORT int pda_mt_chunk() { return MT_CHUNK; }

PDA_EXPORT int pda_mt_scale(int dt_in, int dt_out, const long long* meta, const int* blk, int nblocks, float scale,
                            const float* dev_scale, int invert, int* noop_flag, hipStream_t st) {
  if (nblocks == 0) return 0;
#define L_SCALE(TI, TO, ...) \
  hipLaunchKernelGGL((mt_scale_kernel<TI, TO>), dim3(nblocks), dim3(256), 0, st, meta, blk, scale, dev_scale, invert, noop_flag)
  MT_SWITCH2(dt_in, dt_out, L_SCALE, 0)
#undef L_SCALE
  return pda_check_launch();
}

PDA_EXPORT int pda_mt_axpby(int dt_x, int dt_y, int dt_out, const long long* meta, const int* blk, int nblocks,
                            float a, float b, int* noop_flag, hipStream_t st) {
  if (nblocks == 0) return 0;
#define L_AX(TX, TY, TO) \
  hipLaunchKernelGGL((mt_axpby_kernel<TX, TY, TO>), dim3(nblocks), dim3(256), 0, st, meta, blk, a, b, noop_flag)
  if (dt_x == PDA_F32 && dt_y == PDA_F32 && dt_out == PDA_F32) L_AX(float, float, float);
  else if (dt_x == PDA_F16 && dt_y == PDA_F32 && dt_out == PDA_F32) L_AX(f16, float, float);
  else if (dt_x == PDA_BF16 && dt_y == PDA_F32 && dt_out == PDA_F32) L_AX(bf16, float, float);
  else if (dt_x == PDA_F16 && dt_y == PDA_F16 && dt_out == PDA_F16) L_AX(f16, f16, f16);
  else if (dt_x == PDA_BF16 && dt_y == PDA_BF16 && dt_out == PDA_BF16) L_AX(bf16, bf16, bf16);
  else return -6;
#undef L_AX
  return pda_check_launch();
}

PDA_EXPORT int pda_mt_l2norm(int dtype, const long long* meta, const int* blk, int nblocks, float* sumsq,
                             int per_tensor, hipStream_t st) {
  if (nblocks == 0) return 0;
  if (dtype == PDA_F32) hipLaunchKernelGGL((mt_l2norm_kernel<float>), dim3(nblocks), dim3(256), 0, st, meta, blk, sumsq, per_tensor);
  else if (dtype == PDA_F16) hipLaunchKernelGGL((mt_l2norm_kernel<f16>), dim3(nblocks), dim3(256), 0, st, meta, blk, sumsq, per_tensor);
  else if (dtype == PDA_BF16) hipLaunchKernelGGL((mt_l2norm_kernel<bf16>), dim3(nblocks), dim3(256), 0, st, meta, blk, sumsq, per_tensor);
  else return -6;
  return pda_check_launch();
}

PDA_EXPORT int pda_sqrt_inplace(float* x, int n, hipStream_t st) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(sqrt_inplace_kernel, dim3((n + 255) / 256), dim3(256), 0, st, x, n);
  return pda_check_launch();
}

PDA_EXPORT int pda_mt_sgd(int dt_grad, int dt_model, const long long* meta, const int* blk, int nblocks,
                          const float* hp, const float* dev_scale, const int* noop_flag, int first_run,
                          float gscale, hipStream_t st) {
  if (nblocks == 0) return 0;
#define L_SGD(TG, TM, HM) \
  hipLaunchKernelGGL((mt_sgd_kernel<TG, TM, HM>), dim3(nblocks), dim3(256), 0, st, meta, blk, hp, dev_scale, noop_flag, first_run, gscale)
  if (dt_model < 0) {
    if (dt_grad == PDA_F32) L_SGD(float, float, false);
    else if (dt_grad == PDA_F16) L_SGD(f16, float, false);
    else if (dt_grad == PDA_BF16) L_SGD(bf16, float, false);
    else return -6;
  } else if (dt_model == PDA_F16) {
    if (dt_grad == PDA_F32) L_SGD(float, f16, true);
    else if (dt_grad == PDA_F16) L_SGD(f16, f16, true);
    else return -6;
  } else if (dt_model == PDA_BF16) {
    if (dt_grad == PDA_F32) L_SGD(float, bf16, true);
    else if (dt_grad == PDA_BF16) L_SGD(bf16, bf16, true);
    else return -6;
  } else return -6;
#undef L_SGD
  return pda_check_launch();
}

PDA_EXPORT int pda_scaler_update(float* state, int* noop_flag, float growth, float backoff, int interval,
                                 float min_scale, float max_scale, hipStream_t st) {
  hipLaunchKernelGGL(scaler_update_kernel, dim3(1), dim3(64), 0, st, state, noop_flag, growth, backoff, interval,
                     min_scale, max_scale);
  return pda_check_launch();
}
