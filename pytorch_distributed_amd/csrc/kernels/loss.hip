// Softmax cross-entropy, log-softmax / NLL and accuracy counting, fused and
// device-resident (reference: nn.CrossEntropyLoss +
// accuracy() at /root/reference/2.distributed.py:118,214,344-364, the true top-k
// variant at /root/reference/6.distributed_slurm_main.py:335-349 and
// F.log_softmax/F.nll_loss at /root/reference/5.2.horovod_pytorch_mnist.py:52,64,90).
//
// One wave (64 lanes) per row: online max / sum-exp in registers, then a second
// sweep writes d(logits) = (softmax - onehot) * grad_scale in the same launch.
// Per-row losses and top-1 / top-5 hits are accumulated into a 4-float metrics
// vector {loss_sum, correct@1, correct@5, rows} with one atomic per wave, so the
// training loop never has to read a scalar back between forward and backward.
#include "common.h"

template <typename T>
__global__ void __launch_bounds__(256)
softmax_ce_kernel(const T* __restrict__ logits, const long long* __restrict__ target, int B, int NC,
                  float grad_scale, T* __restrict__ dlogits, float* __restrict__ row_loss,
                  float* __restrict__ metrics, T* __restrict__ logprob) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= B) return;
  const T* z = logits + (long long)row * NC;
  float m = -INFINITY, s = 0.f;
  for (int c = lane; c < NC; c += 64) {
    const float v = ElemTraits<T>::to_f32(z[c]);
    if (v > m) { s = s * __expf(m - v); m = v; }
    s += __expf(v - m);
  }
  // combine (m, s) across lanes
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
    const float mn = fmaxf(m, m2);
    s = (m == -INFINITY ? 0.f : s * __expf(m - mn)) + (m2 == -INFINITY ? 0.f : s2 * __expf(m2 - mn));
    m = mn;
  }
  const float lse = m + __logf(s);
  const long long tg = target[row];
  const bool tvalid = tg >= 0 && tg < NC;
  const float zt = tvalid ? ElemTraits<T>::to_f32(z[tg]) : 0.f;
  // rank of the target logit = number of classes strictly greater (ties broken
  // by lower index first, matching argmax/topk's first-occurrence rule)
  int greater = 0;
  for (int c = lane; c < NC; c += 64) {
    const float v = ElemTraits<T>::to_f32(z[c]);
    greater += (v > zt || (v == zt && c < tg)) ? 1 : 0;
    const float p = __expf(v - lse);
    if (dlogits) {
      const float g = (p - ((tvalid && c == tg) ? 1.f : 0.f)) * grad_scale;
      dlogits[(long long)row * NC + c] = ElemTraits<T>::from_f32(tvalid ? g : 0.f);
    }
    if (logprob) logprob[(long long)row * NC + c] = ElemTraits<T>::from_f32(v - lse);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) greater += __shfl_xor(greater, o, 64);
  if (lane == 0) {
    const float loss = tvalid ? (lse - zt) : 0.f;
    if (row_loss) row_loss[row] = loss;
    if (metrics) {
      atomicAdd(metrics + 0, loss);
      atomicAdd(metrics + 1, (tvalid && greater < 1) ? 1.f : 0.f);
      atomicAdd(metrics + 2, (tvalid && greater < 5) ? 1.f : 0.f);
      atomicAdd(metrics + 3, 1.f);
    }
  }
}

// NLL over precomputed log-probabilities: loss rows + gradient wrt log-probs.
template <typename T>
__global__ void nll_kernel(const T* __restrict__ logprob, const long long* __restrict__ target, int B, int NC,
                           float* __restrict__ row_loss, float* __restrict__ metrics) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= B) return;
  const long long tg = target[row];
  const float l = (tg >= 0 && tg < NC) ? -ElemTraits<T>::to_f32(logprob[(long long)row * NC + tg]) : 0.f;
  if (row_loss) row_loss[row] = l;
  if (metrics) atomicAdd(metrics, l);
}

// dlogits = dlogits_unit * upstream (upstream = d loss_total / d loss_mean, a device scalar)
template <typename T>
__global__ void scale_by_devscalar_kernel(T* __restrict__ x, const float* __restrict__ s, long long n) {
  const float f = *s;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    x[i] = ElemTraits<T>::from_f32(ElemTraits<T>::to_f32(x[i]) * f);
}

#define PDA_DISPATCH_T(dtype, CALL)                 \
  switch (dtype) {                                  \
    case PDA_F32: { typedef float T; CALL; } break; \
    case PDA_F16: { typedef f16 T; CALL; } break;   \
    case PDA_BF16: { typedef bf16 T; CALL; } break; \
    default: return -6;                             \
  }

// metrics (4 floats) is accumulated into, not overwritten: zero it per step/epoch.
PDA_EXPORT int pda_softmax_ce(int dtype, const void* logits, const long long* target, int B, int NC,
                              float grad_scale, void* dlogits, float* row_loss, float* metrics, void* logprob,
                              hipStream_t st) {
  if (B == 0) return 0;
  const int rows_per_blk = 4;
  PDA_DISPATCH_T(dtype, hipLaunchKernelGGL((softmax_ce_kernel<T>), dim3((B + rows_per_blk - 1) / rows_per_blk),
                                           dim3(64 * rows_per_blk), 0, st, (const T*)logits, target, B, NC,
                                           grad_scale, (T*)dlogits, row_loss, metrics, (T*)logprob));
  return pda_check_launch();
}

PDA_EXPORT int pda_nll(int dtype, const void* logprob, const long long* target, int B, int NC, float* row_loss,
                       float* metrics, hipStream_t st) {
  if (B == 0) return 0;
  PDA_DISPATCH_T(dtype, hipLaunchKernelGGL((nll_kernel<T>), dim3((B + 255) / 256), dim3(256), 0, st,
                                           (const T*)logprob, target, B, NC, row_loss, metrics));
  return pda_check_launch();
}

PDA_EXPORT int pda_scale_by_devscalar(int dtype, void* x, const float* s, long long n, hipStream_t st) {
  if (n == 0) return 0;
  long long b = (n + 255) / 256;
  if (b > 2048) b = 2048;
  PDA_DISPATCH_T(dtype, hipLaunchKernelGGL((scale_by_devscalar_kernel<T>), dim3((int)b), dim3(256), 0, st, (T*)x, s, n));
  return pda_check_launch();
}
