// Direct (non-MFMA) convolution for shapes the implicit-GEMM kernels do not
// cover: channel counts that are not a multiple of the 16-byte chunk (the MNIST
// net's 1->10->20 channel 5x5 convs, /root/reference/5.2.horovod_pytorch_mnist.py:39-40)
// or non-power-of-two strides in dgrad.  NHWC activations, KRSC weights, fp32
// accumulation, any dtype.  These are correctness fallbacks for tiny layers, not
// a performance path.
#include "common.h"

struct DConv {
  int N, Hi, Wi, Ci, Ho, Wo, Co, R, S, stride_h, stride_w, pad_h, pad_w;
};

template <typename T>
__global__ void __launch_bounds__(256)
dconv_fprop_kernel(const T* __restrict__ x, const T* __restrict__ w, const float* __restrict__ bias,
                   T* __restrict__ y, DConv d) {
  const long long total = (long long)d.N * d.Ho * d.Wo * d.Co;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(i % d.Co);
    long long r_ = i / d.Co;
    const int wo = (int)(r_ % d.Wo); r_ /= d.Wo;
    const int ho = (int)(r_ % d.Ho);
    const int n = (int)(r_ / d.Ho);
    float acc = bias ? bias[k] : 0.f;
    for (int r = 0; r < d.R; ++r) {
      const int hi = ho * d.stride_h - d.pad_h + r;
      if ((unsigned)hi >= (unsigned)d.Hi) continue;
      for (int s = 0; s < d.S; ++s) {
        const int wi = wo * d.stride_w - d.pad_w + s;
        if ((unsigned)wi >= (unsigned)d.Wi) continue;
        const T* xp = x + (((long long)n * d.Hi + hi) * d.Wi + wi) * d.Ci;
        const T* wp = w + (((long long)k * d.R + r) * d.S + s) * d.Ci;
        for (int c = 0; c < d.Ci; ++c) acc += ElemTraits<T>::to_f32(xp[c]) * ElemTraits<T>::to_f32(wp[c]);
      }
    }
    y[i] = ElemTraits<T>::from_f32(acc);
  }
}

template <typename T>
__global__ void __launch_bounds__(256)
dconv_dgrad_kernel(const T* __restrict__ dy, const T* __restrict__ w, T* __restrict__ dx, DConv d) {
  const long long total = (long long)d.N * d.Hi * d.Wi * d.Ci;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % d.Ci);
    long long r_ = i / d.Ci;
    const int wi = (int)(r_ % d.Wi); r_ /= d.Wi;
    const int hi = (int)(r_ % d.Hi);
    const int n = (int)(r_ / d.Hi);
    float acc = 0.f;
    for (int r = 0; r < d.R; ++r) {
      const int hn = hi + d.pad_h - r;
      if (hn < 0 || hn % d.stride_h != 0) continue;
      const int ho = hn / d.stride_h;
      if (ho >= d.Ho) continue;
      for (int s = 0; s < d.S; ++s) {
        const int wn = wi + d.pad_w - s;
        if (wn < 0 || wn % d.stride_w != 0) continue;
        const int wo = wn / d.stride_w;
        if (wo >= d.Wo) continue;
        const T* gp = dy + (((long long)n * d.Ho + ho) * d.Wo + wo) * d.Co;
        for (int k = 0; k < d.Co; ++k)
          acc += ElemTraits<T>::to_f32(gp[k]) * ElemTraits<T>::to_f32(w[(((long long)k * d.R + r) * d.S + s) * d.Ci + c]);
      }
    }
    dx[i] = ElemTraits<T>::from_f32(acc);
  }
}

// one block per weight element (k, r, s, c): block-reduce over all output pixels
template <typename T>
__global__ void __launch_bounds__(256)
dconv_wgrad_kernel(const T* __restrict__ dy, const T* __restrict__ x, float* __restrict__ dw, DConv d) {
  __shared__ float scratch[16];
  const long long e = blockIdx.x;
  const int c = (int)(e % d.Ci);
  long long r_ = e / d.Ci;
  const int s = (int)(r_ % d.S); r_ /= d.S;
  const int r = (int)(r_ % d.R);
  const int k = (int)(r_ / d.R);
  const long long M = (long long)d.N * d.Ho * d.Wo;
  float acc = 0.f;
  for (long long m = threadIdx.x; m < M; m += blockDim.x) {
    const int wo = (int)(m % d.Wo);
    long long q = m / d.Wo;
    const int ho = (int)(q % d.Ho);
    const int n = (int)(q / d.Ho);
    const int hi = ho * d.stride_h - d.pad_h + r, wi = wo * d.stride_w - d.pad_w + s;
    if ((unsigned)hi < (unsigned)d.Hi && (unsigned)wi < (unsigned)d.Wi)
      acc += ElemTraits<T>::to_f32(dy[m * d.Co + k]) *
             ElemTraits<T>::to_f32(x[(((long long)n * d.Hi + hi) * d.Wi + wi) * d.Ci + c]);
  }
  const float tot = block_sum(acc, scratch);
  if (threadIdx.x == 0) dw[e] += tot;
}

static inline int ew_blocks(long long n) {
  long long b = (n + 255) / 256;
  if (b > 8192) b = 8192;
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

// mode: 0 fprop (a=x, b=w, out=y, bias optional), 1 dgrad (a=dy, b=w, out=dx), 2 wgrad (a=dy, b=x, out=dw fp32, accumulated)
PDA_EXPORT int pda_direct_conv(int dtype, int mode, const void* a, const void* b, const float* bias, void* out, int N,
                               int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int R, int S, int stride_h,
                               int stride_w, int pad_h, int pad_w, hipStream_t st) {
  DConv d{N, Hi, Wi, Ci, Ho, Wo, Co, R, S, stride_h, stride_w, pad_h, pad_w};
  if (mode == 0) {
    const long long total = (long long)N * Ho * Wo * Co;
    if (total == 0) return 0;
    PDA_DISPATCH_T(dtype, hipLaunchKernelGGL((dconv_fprop_kernel<T>), dim3(ew_blocks(total)), dim3(256), 0, st,
                                             (const T*)a, (const T*)b, bias, (T*)out, d));
  } else if (mode == 1) {
    const long long total = (long long)N * Hi * Wi * Ci;
    if (total == 0) return 0;
    PDA_DISPATCH_T(dtype, hipLaunchKernelGGL((dconv_dgrad_kernel<T>), dim3(ew_blocks(total)), dim3(256), 0, st,
                                             (const T*)a, (const T*)b, (T*)out, d));
  } else if (mode == 2) {
    const long long total = (long long)Co * R * S * Ci;
    if (total == 0) return 0;
    if (total > 0x7fffffffLL) return -1;
    PDA_DISPATCH_T(dtype, hipLaunchKernelGGL((dconv_wgrad_kernel<T>), dim3((unsigned)total), dim3(256), 0, st,
                                             (const T*)a, (const T*)b, (float*)out, d));
  } else return -2;
  return pda_check_launch();
}
