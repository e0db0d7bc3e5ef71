// Pooling for NHWC activations: max-pool (any window/stride/pad; ResNet stem
// 3x3 s2 p1 and the 2x2 pools of the MNIST net) with a saved arg-max byte so the
// backward is a gather, and global average pooling (reference
// triggers: torchvision ResNet maxpool/avgpool under
// /root/reference/2.distributed.py:213 and F.max_pool2d in
// /root/reference/5.2.horovod_pytorch_mnist.py:46-47).
#include "common.h"

// Both max-pool kernels give one image row to a workgroup: the (image, row) split and the window rows are scalar
// work done once, a thread's only division is its element index by the vectors per pixel (a shift when that is a
// power of two).  With one flat index per thread the seven integer divisions of the index arithmetic cost more
// VALU time than the kernel's memory traffic takes (ResNet-50 stem at batch 400: 377 us backward for a 180 us floor).
__device__ __forceinline__ int div_pow2_or(int v, int d, int shift) { return shift >= 0 ? v >> shift : v / d; }

// KW: compile-time window size (0 = any, from `k`): all taps of a window are then requested before the first
// comparison instead of one load latency after the other.
template <typename T, int EPC, int KW>
__global__ void __launch_bounds__(512)
maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ out, unsigned char* __restrict__ idx, int N, int H,
                   int W, int C, int Ho, int Wo, int k, int s, int p, const float* __restrict__ scale,
                   const float* __restrict__ shift, int vshift) {
  // scale/shift given: x is a pre-normalisation tensor and the pooled value is relu(x*scale+shift), rounded
  // to T exactly as a separate BatchNorm pass would have stored it (the ResNet stem: the 112x112 activation
  // is never written)
  const int vpr = C / EPC;
  const long long row = blockIdx.x;            // n * Ho + ho
  const int ho = (int)(row % Ho);
  const long long n = row / Ho;
  const int h0 = ho * s - p;
  const int a_lo = h0 < 0 ? -h0 : 0, a_hi = (h0 + k > H ? H - h0 : k);   // window rows inside the image
  const T* __restrict__ xn = x + n * (long long)H * W * C;
  // the workgroup size is a multiple of the vectors per pixel whenever it can be (host): a thread then keeps its
  // channels from one element to the next and loads their coefficients once
  const bool fixed_cv = blockDim.x % vpr == 0;
  float sc[EPC], sf[EPC];
  if (scale && fixed_cv) {
    const int cv = threadIdx.x % vpr;
#pragma unroll
    for (int e = 0; e < EPC; ++e) { sc[e] = scale[cv * EPC + e]; sf[e] = shift[cv * EPC + e]; }
  }
  for (int j = threadIdx.x; j < Wo * vpr; j += blockDim.x) {
    const int wo = div_pow2_or(j, vpr, vshift);
    const int cv = j - wo * vpr;
    float best[EPC];
    unsigned char bi[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) { best[e] = -INFINITY; bi[e] = 0; }
    if (scale && !fixed_cv) {
#pragma unroll
      for (int e = 0; e < EPC; ++e) { sc[e] = scale[cv * EPC + e]; sf[e] = shift[cv * EPC + e]; }
    }
    const int w0 = wo * s - p;
    const int b_lo = w0 < 0 ? -w0 : 0, b_hi = (w0 + k > W ? W - w0 : k);
    auto take = [&](const VecN<T, EPC>& v, int a, int b) {
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        float f = ElemTraits<T>::to_f32(v.v[e]);
        if (scale) f = ElemTraits<T>::to_f32(ElemTraits<T>::from_f32(fmaxf(f * sc[e] + sf[e], 0.f)));
        if (f > best[e] || f != f) { best[e] = f; bi[e] = (unsigned char)(a * k + b); }
      }
    };
    if constexpr (KW > 0) {
      VecN<T, EPC> v[KW * KW];
      bool ok[KW * KW];
#pragma unroll
      for (int a = 0; a < KW; ++a)
#pragma unroll
        for (int b = 0; b < KW; ++b) {
          ok[a * KW + b] = a >= a_lo && a < a_hi && b >= b_lo && b < b_hi;
          const long long o = ok[a * KW + b] ? ((long long)(h0 + a) * W + (w0 + b)) * C + cv * EPC : (long long)cv * EPC;
          v[a * KW + b] = loadN<T, EPC>(xn + o);
        }
#pragma unroll
      for (int a = 0; a < KW; ++a)
#pragma unroll
        for (int b = 0; b < KW; ++b)
          if (ok[a * KW + b]) take(v[a * KW + b], a, b);
    } else {
      for (int a = a_lo; a < a_hi; ++a)
        for (int b = b_lo; b < b_hi; ++b)
          take(loadN<T, EPC>(xn + ((long long)(h0 + a) * W + (w0 + b)) * C + cv * EPC), a, b);
    }
    const long long i = row * Wo * vpr + j;
    VecN<T, EPC> o;
#pragma unroll
    for (int e = 0; e < EPC; ++e) o.v[e] = ElemTraits<T>::from_f32(best[e]);
    storeN<T, EPC>(out + i * EPC, o);
    if (idx) {
      if constexpr (EPC == 8) {   // one 8-byte store instead of eight byte stores
        u32x2 pk = {0u, 0u};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          pk[0] |= (unsigned)bi[e] << (8 * e);
          pk[1] |= (unsigned)bi[4 + e] << (8 * e);
        }
        *reinterpret_cast<u32x2*>(idx + i * EPC) = pk;
      } else if constexpr (EPC == 4) {
        unsigned pk = 0u;
#pragma unroll
        for (int e = 0; e < 4; ++e) pk |= (unsigned)bi[e] << (8 * e);
        *reinterpret_cast<unsigned*>(idx + i * EPC) = pk;
      } else {
#pragma unroll
        for (int e = 0; e < EPC; ++e) idx[i * EPC + e] = bi[e];
      }
    }
  }
}

// One thread per input pixel-vector: sum the output gradients of every window
// whose arg-max byte points back at this pixel (no atomics, deterministic).
template <typename T, int EPC>
__global__ void __launch_bounds__(512)
maxpool_bwd_kernel(const T* __restrict__ dout, const unsigned char* __restrict__ idx, T* __restrict__ dx, int N,
                   int H, int W, int C, int Ho, int Wo, int k, int s, int p, int vshift, int sshift) {
  const int vpr = C / EPC;
  const long long row = blockIdx.x;            // n * H + h
  const int h = (int)(row % H);
  const long long n = row / H;
  // windows ho with ho*s - p <= h <= ho*s - p + k - 1
  int ho_lo = h + p - k + 1 + s - 1;
  ho_lo = ho_lo < 0 ? 0 : ho_lo / s;
  int ho_hi = (h + p) / s;
  if (ho_hi > Ho - 1) ho_hi = Ho - 1;
  for (int j = threadIdx.x; j < W * vpr; j += blockDim.x) {
    const int w = div_pow2_or(j, vpr, vshift);
    const int cv = j - w * vpr;
    int wo_lo = w + p - k + 1 + s - 1;
    wo_lo = wo_lo < 0 ? 0 : div_pow2_or(wo_lo, s, sshift);
    int wo_hi = div_pow2_or(w + p, s, sshift);
    if (wo_hi > Wo - 1) wo_hi = Wo - 1;
    float acc[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
    struct Cand { VecN<T, EPC> gv; unsigned char ib[EPC]; };
    auto fetch = [&](int ho, int wo) {
      Cand c;
      const long long o = (((n * Ho + ho) * Wo + wo) * vpr + cv) * EPC;
      c.gv = loadN<T, EPC>(dout + o);
      if constexpr (EPC == 8) {
        *reinterpret_cast<u32x2*>(c.ib) = *reinterpret_cast<const u32x2*>(idx + o);
      } else if constexpr (EPC == 4) {
        *reinterpret_cast<unsigned int*>(c.ib) = *reinterpret_cast<const unsigned int*>(idx + o);
      } else {
#pragma unroll
        for (int e = 0; e < EPC; ++e) c.ib[e] = idx[o + e];
      }
      return c;
    };
    auto take = [&](const Cand& c, int ho, int wo) {
      const int a = h - (ho * s - p), b = w - (wo * s - p);
      const unsigned char me = (unsigned char)(a * k + b);
#pragma unroll
      for (int e = 0; e < EPC; ++e)
        if (c.ib[e] == me) acc[e] += ElemTraits<T>::to_f32(c.gv.v[e]);
    };
    // (requesting all four candidate windows of a k <= 2 s pool up front measured slower, 306 vs 249 us on the
    // ResNet stem: a pixel lies in 2.25 windows on average and the extra requests cost more than the latency)
    for (int ho = ho_lo; ho <= ho_hi; ++ho)
      for (int wo = wo_lo; wo <= wo_hi; ++wo) take(fetch(ho, wo), ho, wo);
    VecN<T, EPC> o;
#pragma unroll
    for (int e = 0; e < EPC; ++e) o.v[e] = ElemTraits<T>::from_f32(acc[e]);
    storeN<T, EPC>(dx + (row * W * vpr + j) * EPC, o);
  }
}

// Average pooling, padding counted in the divisor (torch's count_include_pad=True default): Inception's
// 3x3 s1 p1 branch pools, the 5x5 s3 pool of its auxiliary head, DenseNet's 2x2 s2 transitions.
template <typename T, int EPC>
__global__ void __launch_bounds__(256)
avgpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ out, int N, int H, int W, int C, int Ho, int Wo, int k,
                   int s, int p) {
  const int vpr = C / EPC;
  const long long total = (long long)N * Ho * Wo * vpr;
  const float inv = 1.f / (float)(k * k);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int cv = (int)(i % vpr);
    long long r = i / vpr;
    const int wo = (int)(r % Wo); r /= Wo;
    const int ho = (int)(r % Ho);
    const int n = (int)(r / Ho);
    float acc[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
    for (int a = 0; a < k; ++a) {
      const int h = ho * s - p + a;
      if ((unsigned)h >= (unsigned)H) continue;
      for (int b = 0; b < k; ++b) {
        const int w = wo * s - p + b;
        if ((unsigned)w >= (unsigned)W) continue;
        const VecN<T, EPC> v = loadN<T, EPC>(x + (((long long)n * H + h) * W + w) * C + cv * EPC);
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] += ElemTraits<T>::to_f32(v.v[e]);
      }
    }
    VecN<T, EPC> o;
#pragma unroll
    for (int e = 0; e < EPC; ++e) o.v[e] = ElemTraits<T>::from_f32(acc[e] * inv);
    storeN<T, EPC>(out + i * EPC, o);
  }
}

// One thread per input pixel-vector: the sum of the output gradients of every window that covers it.
template <typename T, int EPC>
__global__ void __launch_bounds__(256)
avgpool_bwd_kernel(const T* __restrict__ dout, T* __restrict__ dx, int N, int H, int W, int C, int Ho, int Wo, int k,
                   int s, int p) {
  const int vpr = C / EPC;
  const long long total = (long long)N * H * W * vpr;
  const float inv = 1.f / (float)(k * k);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int cv = (int)(i % vpr);
    long long r = i / vpr;
    const int w = (int)(r % W); r /= W;
    const int h = (int)(r % H);
    const long long n = r / H;
    // windows ho with ho*s - p <= h <= ho*s - p + k - 1
    int ho_lo = h + p - k + 1 + s - 1;
    ho_lo = ho_lo < 0 ? 0 : ho_lo / s;
    int ho_hi = (h + p) / s;
    if (ho_hi > Ho - 1) ho_hi = Ho - 1;
    int wo_lo = w + p - k + 1 + s - 1;
    wo_lo = wo_lo < 0 ? 0 : wo_lo / s;
    int wo_hi = (w + p) / s;
    if (wo_hi > Wo - 1) wo_hi = Wo - 1;
    float acc[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
    for (int ho = ho_lo; ho <= ho_hi; ++ho)
      for (int wo = wo_lo; wo <= wo_hi; ++wo) {
        const VecN<T, EPC> g = loadN<T, EPC>(dout + (((n * Ho + ho) * Wo + wo) * vpr + cv) * EPC);
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] += ElemTraits<T>::to_f32(g.v[e]);
      }
    VecN<T, EPC> o;
#pragma unroll
    for (int e = 0; e < EPC; ++e) o.v[e] = ElemTraits<T>::from_f32(acc[e] * inv);
    storeN<T, EPC>(dx + i * EPC, o);
  }
}

// out[n][c] = mean over HW of x[n][hw][c]   (fp32 accumulate)
template <typename T, int EPC>
__global__ void __launch_bounds__(256)
gap_fwd_kernel(const T* __restrict__ x, T* __restrict__ out, int N, int HW, int C) {
  const int vpr = C / EPC;
  const long long total = (long long)N * vpr;
  const float inv = 1.f / (float)HW;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int cv = (int)(i % vpr);
    const long long n = i / vpr;
    float acc[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
    for (int p = 0; p < HW; ++p) {
      VecN<T, EPC> v = loadN<T, EPC>(x + (n * HW + p) * C + cv * EPC);
#pragma unroll
      for (int e = 0; e < EPC; ++e) acc[e] += ElemTraits<T>::to_f32(v.v[e]);
    }
    VecN<T, EPC> o;
#pragma unroll
    for (int e = 0; e < EPC; ++e) o.v[e] = ElemTraits<T>::from_f32(acc[e] * inv);
    storeN<T, EPC>(out + i * EPC, o);
  }
}

template <typename T, int EPC>
__global__ void __launch_bounds__(256)
gap_bwd_kernel(const T* __restrict__ dout, T* __restrict__ dx, int N, int HW, int C) {
  const int vpr = C / EPC;
  const long long total = (long long)N * HW * vpr;
  const float inv = 1.f / (float)HW;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int cv = (int)(i % vpr);
    const long long n = i / ((long long)HW * vpr);
    VecN<T, EPC> g = loadN<T, EPC>(dout + (n * vpr + cv) * EPC);
    VecN<T, EPC> o;
#pragma unroll
    for (int e = 0; e < EPC; ++e) o.v[e] = ElemTraits<T>::from_f32(ElemTraits<T>::to_f32(g.v[e]) * inv);
    storeN<T, EPC>(dx + i * EPC, o);
  }
}

static inline int ew_blocks(long long n) {
  long long b = (n + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}
// VEC = elements per lane: a full 16-byte vector when C allows it, else 1.
#define PDA_DISPATCH_TV(dtype, vec, CALL)                                        \
  switch (dtype) {                                                               \
    case PDA_F32: { typedef float T; if (vec) { constexpr int EPC = 4; CALL; } else { constexpr int EPC = 1; CALL; } } break; \
    case PDA_F16: { typedef f16 T; if (vec) { constexpr int EPC = 8; CALL; } else { constexpr int EPC = 1; CALL; } } break;   \
    case PDA_BF16: { typedef bf16 T; if (vec) { constexpr int EPC = 8; CALL; } else { constexpr int EPC = 1; CALL; } } break; \
    default: return -6;                                                          \
  }

static int log2_or_minus1(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return (v > 0 && (1 << l) == v) ? l : -1;
}
// threads of a one-row workgroup: whole waves; the whole row at once up to 512 threads, else 256 per pass
static int pool_row_threads(long long elems) {
  const long long t = ((elems + 63) / 64) * 64;
  return (int)(t < 64 ? 64 : (t > 512 ? 256 : t));
}

PDA_EXPORT int pda_maxpool_fwd(int dtype, const void* x, void* out, unsigned char* idx, int N, int H, int W, int C,
                               int Ho, int Wo, int k, int s, int p, const float* scale, const float* shift,
                               hipStream_t st) {
  int epc = dtype == PDA_F32 ? 4 : 8;
  const bool vec = (C % epc) == 0;
  if (!vec) epc = 1;
  if (k * k > 255) return -1;
  const long long total = (long long)N * Ho * Wo * (C / epc);
  if (total == 0) return 0;
  const long long rows = (long long)N * Ho;
  if (rows > 0x7fffffffLL) return -4;
  const int vpr = C / epc, nt = pool_row_threads((long long)Wo * vpr);
#define PDA_MAXPOOL_FWD(KW_)                                                                                         \
  PDA_DISPATCH_TV(dtype, vec, hipLaunchKernelGGL((maxpool_fwd_kernel<T, EPC, KW_>), dim3((unsigned)rows), dim3(nt), 0, st, \
                                           (const T*)x, (T*)out, idx, N, H, W, C, Ho, Wo, k, s, p, scale, shift,      \
                                           log2_or_minus1(vpr)))
  if (k == 3) { PDA_MAXPOOL_FWD(3); }
  else if (k == 2) { PDA_MAXPOOL_FWD(2); }
  else { PDA_MAXPOOL_FWD(0); }
#undef PDA_MAXPOOL_FWD
  return pda_check_launch();
}

PDA_EXPORT int pda_maxpool_bwd(int dtype, const void* dout, const unsigned char* idx, void* dx, int N, int H, int W,
                               int C, int Ho, int Wo, int k, int s, int p, hipStream_t st) {
  int epc = dtype == PDA_F32 ? 4 : 8;
  const bool vec = (C % epc) == 0;
  if (!vec) epc = 1;
  const long long total = (long long)N * H * W * (C / epc);
  if (total == 0) return 0;
  const long long rows = (long long)N * H;
  if (rows > 0x7fffffffLL) return -4;
  const int vpr = C / epc, nt = pool_row_threads((long long)W * vpr);
PDA_DISPATCH_TV(dtype, vec, hipLaunchKernelGGL((maxpool_bwd_kernel<T, EPC>), dim3((unsigned)rows), dim3(nt), 0, st,
                                           (const T*)dout, idx, (T*)dx, N, H, W, C, Ho, Wo, k, s, p,
                                           log2_or_minus1(vpr), log2_or_minus1(s)));
  return pda_check_launch();
}

// dir 0: a = x [N,H,W,C] -> b = out [N,Ho,Wo,C];  dir 1: a = dout [N,Ho,Wo,C] -> b = dx [N,H,W,C]
PDA_EXPORT int pda_avgpool(int dtype, int dir, const void* a, void* b, int N, int H, int W, int C, int Ho, int Wo,
                           int k, int s, int p, hipStream_t st) {
  if (k < 1 || s < 1 || p < 0 || 2 * p > k || H + 2 * p < k || W + 2 * p < k) return -1;
  if (Ho != (H + 2 * p - k) / s + 1 || Wo != (W + 2 * p - k) / s + 1) return -2;
  int epc = dtype == PDA_F32 ? 4 : 8;
  const bool vec = (C % epc) == 0;
  if (!vec) epc = 1;
  const long long total = (long long)N * (dir == 0 ? (long long)Ho * Wo : (long long)H * W) * (C / epc);
  if (total == 0) return 0;
  if (dir == 0) {
    PDA_DISPATCH_TV(dtype, vec, hipLaunchKernelGGL((avgpool_fwd_kernel<T, EPC>), dim3(ew_blocks(total)), dim3(256), 0, st,
                                             (const T*)a, (T*)b, N, H, W, C, Ho, Wo, k, s, p));
  } else {
    PDA_DISPATCH_TV(dtype, vec, hipLaunchKernelGGL((avgpool_bwd_kernel<T, EPC>), dim3(ew_blocks(total)), dim3(256), 0, st,
                                             (const T*)a, (T*)b, N, H, W, C, Ho, Wo, k, s, p));
  }
  return pda_check_launch();
}

PDA_EXPORT int pda_gap_fwd(int dtype, const void* x, void* out, int N, int HW, int C, hipStream_t st) {
  int epc = dtype == PDA_F32 ? 4 : 8;
  const bool vec = (C % epc) == 0;
  if (!vec) epc = 1;
  const long long total = (long long)N * (C / epc);
  if (total == 0) return 0;
  PDA_DISPATCH_TV(dtype, vec, hipLaunchKernelGGL((gap_fwd_kernel<T, EPC>), dim3(ew_blocks(total)), dim3(256), 0, st,
                                           (const T*)x, (T*)out, N, HW, C));
  return pda_check_launch();
}

PDA_EXPORT int pda_gap_bwd(int dtype, const void* dout, void* dx, int N, int HW, int C, hipStream_t st) {
  int epc = dtype == PDA_F32 ? 4 : 8;
  const bool vec = (C % epc) == 0;
  if (!vec) epc = 1;
  const long long total = (long long)N * HW * (C / epc);
  if (total == 0) return 0;
  PDA_DISPATCH_TV(dtype, vec, hipLaunchKernelGGL((gap_bwd_kernel<T, EPC>), dim3(ew_blocks(total)), dim3(256), 0, st,
                                           (const T*)dout, (T*)dx, N, HW, C));
  return pda_check_launch();
}
