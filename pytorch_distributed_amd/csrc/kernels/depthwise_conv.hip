// Depthwise convolution (one filter per channel: MobileNet / MNASNet / ShuffleNet blocks reached through
// the reference's `--arch` switch, /root/reference/1.dataparallel.py:23-24,97-102) for NHWC activations.
//
// A depthwise layer does R*S multiply-adds per loaded element, so it is bound by HBM, not by the matrix
// cores: these kernels are vector-memory kernels.  Channels are the contiguous dimension, so one lane
// owns a 16-byte channel vector (8 x 16-bit or 4 x fp32) of one pixel and consecutive lanes read
// consecutive vectors - every access of a wave is a run of full cache lines.  Weights are kept tap-major
// ([R][S][C], prepared by the caller) so the per-tap filter values are one 16-byte load as well.
//   fprop: one lane per output pixel-vector, R*S taps gathered from x.
//   dgrad: one lane per input pixel-vector, gathers the output gradients whose window covers it.
//   wgrad: dW[r][s][c] = sum over pixels of dy * x(shifted): a workgroup owns 8 channel vectors and a
//          slab of output pixels; 32 pixel lanes per vector accumulate privately, meet in LDS, and one
//          float atomic per (tap, channel, slab) lands in the fp32 gradient.
#include "common.h"

struct DwGeom {
  int N, Hi, Wi, C, Ho, Wo, R, S, sh, sw, ph, pw;
};

template <typename T, int EPC>
__global__ void __launch_bounds__(256)
dw_fprop_kernel(const T* __restrict__ x, const T* __restrict__ w, const float* __restrict__ bias,
                T* __restrict__ y, DwGeom g) {
  const int vpr = g.C / EPC;
  const long long total = (long long)g.N * g.Ho * g.Wo * vpr;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int cv = (int)(i % vpr);
    long long q = i / vpr;
    const int wo = (int)(q % g.Wo); q /= g.Wo;
    const int ho = (int)(q % g.Ho);
    const int n = (int)(q / g.Ho);
    float acc[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc[e] = bias ? bias[cv * EPC + e] : 0.f;
    for (int r = 0; r < g.R; ++r) {
      const int hi = ho * g.sh - g.ph + r;
      if ((unsigned)hi >= (unsigned)g.Hi) continue;
      for (int s = 0; s < g.S; ++s) {
        const int wi = wo * g.sw - g.pw + s;
        if ((unsigned)wi >= (unsigned)g.Wi) continue;
        const VecN<T, EPC> xv = loadN<T, EPC>(x + (((long long)n * g.Hi + hi) * g.Wi + wi) * g.C + cv * EPC);
        const VecN<T, EPC> wv = loadN<T, EPC>(w + (long long)(r * g.S + s) * g.C + cv * EPC);
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] += ElemTraits<T>::to_f32(xv.v[e]) * ElemTraits<T>::to_f32(wv.v[e]);
      }
    }
    VecN<T, EPC> o;
#pragma unroll
    for (int e = 0; e < EPC; ++e) o.v[e] = ElemTraits<T>::from_f32(acc[e]);
    storeN<T, EPC>(y + i * EPC, o);
  }
}

template <typename T, int EPC>
__global__ void __launch_bounds__(256)
dw_dgrad_kernel(const T* __restrict__ dy, const T* __restrict__ w, T* __restrict__ dx, DwGeom g) {
  const int vpr = g.C / EPC;
  const long long total = (long long)g.N * g.Hi * g.Wi * vpr;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int cv = (int)(i % vpr);
    long long q = i / vpr;
    const int wi = (int)(q % g.Wi); q /= g.Wi;
    const int hi = (int)(q % g.Hi);
    const int n = (int)(q / g.Hi);
    float acc[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
    for (int r = 0; r < g.R; ++r) {
      const int hn = hi + g.ph - r;
      if (hn < 0 || hn % g.sh != 0) continue;
      const int ho = hn / g.sh;
      if (ho >= g.Ho) continue;
      for (int s = 0; s < g.S; ++s) {
        const int wn = wi + g.pw - s;
        if (wn < 0 || wn % g.sw != 0) continue;
        const int wo = wn / g.sw;
        if (wo >= g.Wo) continue;
        const VecN<T, EPC> gv = loadN<T, EPC>(dy + (((long long)n * g.Ho + ho) * g.Wo + wo) * g.C + cv * EPC);
        const VecN<T, EPC> wv = loadN<T, EPC>(w + (long long)(r * g.S + s) * g.C + cv * EPC);
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] += ElemTraits<T>::to_f32(gv.v[e]) * ElemTraits<T>::to_f32(wv.v[e]);
      }
    }
    VecN<T, EPC> o;
#pragma unroll
    for (int e = 0; e < EPC; ++e) o.v[e] = ElemTraits<T>::from_f32(acc[e]);
    storeN<T, EPC>(dx + i * EPC, o);
  }
}

// blockDim = 256 = 8 channel vectors (threadIdx % 8) x 32 pixel lanes (threadIdx / 8).
// grid.x = ceil(vpr / 8), grid.y = slabs of `slab` output pixels.
#define DW_CVB 8
#define DW_PXL 32
template <typename T, int EPC>
__global__ void __launch_bounds__(256)
dw_wgrad_kernel(const T* __restrict__ dy, const T* __restrict__ x, float* __restrict__ dw, DwGeom g,
                long long slab, int use_atomic) {
  __shared__ float red[DW_PXL][DW_CVB * EPC + 1];
  const int vpr = g.C / EPC;
  const int tx = threadIdx.x % DW_CVB, ty = threadIdx.x / DW_CVB;
  const int cv = blockIdx.x * DW_CVB + tx;
  const bool live = cv < vpr;
  const long long M = (long long)g.N * g.Ho * g.Wo;
  const long long m0 = (long long)blockIdx.y * slab;
  const long long m1 = m0 + slab < M ? m0 + slab : M;
  for (int r = 0; r < g.R; ++r)
    for (int s = 0; s < g.S; ++s) {
      float acc[EPC];
#pragma unroll
      for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
      if (live) {
        for (long long m = m0 + ty; m < m1; m += DW_PXL) {
          const int wo = (int)(m % g.Wo);
          const long long q = m / g.Wo;
          const int ho = (int)(q % g.Ho);
          const long long n = q / g.Ho;
          const int hi = ho * g.sh - g.ph + r, wi = wo * g.sw - g.pw + s;
          if ((unsigned)hi >= (unsigned)g.Hi || (unsigned)wi >= (unsigned)g.Wi) continue;
          const VecN<T, EPC> gv = loadN<T, EPC>(dy + m * g.C + cv * EPC);
          const VecN<T, EPC> xv = loadN<T, EPC>(x + ((n * g.Hi + hi) * g.Wi + wi) * g.C + cv * EPC);
#pragma unroll
          for (int e = 0; e < EPC; ++e) acc[e] += ElemTraits<T>::to_f32(gv.v[e]) * ElemTraits<T>::to_f32(xv.v[e]);
        }
      }
#pragma unroll
      for (int e = 0; e < EPC; ++e) red[ty][tx * EPC + e] = acc[e];
      __syncthreads();
      if (threadIdx.x < DW_CVB * EPC) {
        float t = 0.f;
#pragma unroll
        for (int p = 0; p < DW_PXL; ++p) t += red[p][threadIdx.x];
        const int c = blockIdx.x * DW_CVB * EPC + threadIdx.x;
        if (c < g.C) {
          float* dst = dw + (long long)(r * g.S + s) * g.C + c;
          if (use_atomic) atomicAdd(dst, t);
          else *dst += t;   // one slab: this workgroup is the only writer of the element
        }
      }
      __syncthreads();
    }
}

// Same decomposition, a whole filter row per pass: the S taps of row r share one load of dy and sit in
// registers (S x EPC accumulators), so dy is read R times instead of R*S times (3x3 and 5x5 filters).
template <typename T, int EPC, int SW>
__global__ void __launch_bounds__(256)
dw_wgrad_row_kernel(const T* __restrict__ dy, const T* __restrict__ x, float* __restrict__ dw, DwGeom g,
                    long long slab, int use_atomic) {
  __shared__ float red[DW_PXL][DW_CVB * EPC + 1];
  const int vpr = g.C / EPC;
  const int tx = threadIdx.x % DW_CVB, ty = threadIdx.x / DW_CVB;
  const int cv = blockIdx.x * DW_CVB + tx;
  const bool live = cv < vpr;
  const long long M = (long long)g.N * g.Ho * g.Wo;
  const long long m0 = (long long)blockIdx.y * slab;
  const long long m1 = m0 + slab < M ? m0 + slab : M;
  for (int r = 0; r < g.R; ++r) {
    float acc[SW][EPC];
#pragma unroll
    for (int s = 0; s < SW; ++s)
#pragma unroll
      for (int e = 0; e < EPC; ++e) acc[s][e] = 0.f;
    if (live) {
      for (long long m = m0 + ty; m < m1; m += DW_PXL) {
        const int wo = (int)(m % g.Wo);
        const long long q = m / g.Wo;
        const int ho = (int)(q % g.Ho);
        const long long n = q / g.Ho;
        const int hi = ho * g.sh - g.ph + r;
        if ((unsigned)hi >= (unsigned)g.Hi) continue;
        const VecN<T, EPC> gv = loadN<T, EPC>(dy + m * g.C + cv * EPC);
        const T* xrow = x + ((n * g.Hi + hi) * g.Wi) * g.C + cv * EPC;
        const int wi0 = wo * g.sw - g.pw;
#pragma unroll
        for (int s = 0; s < SW; ++s) {
          const int wi = wi0 + s;
          if ((unsigned)wi >= (unsigned)g.Wi) continue;
          const VecN<T, EPC> xv = loadN<T, EPC>(xrow + (long long)wi * g.C);
#pragma unroll
          for (int e = 0; e < EPC; ++e) acc[s][e] += ElemTraits<T>::to_f32(gv.v[e]) * ElemTraits<T>::to_f32(xv.v[e]);
        }
      }
    }
#pragma unroll
    for (int s = 0; s < SW; ++s) {
#pragma unroll
      for (int e = 0; e < EPC; ++e) red[ty][tx * EPC + e] = acc[s][e];
      __syncthreads();
      if (threadIdx.x < DW_CVB * EPC) {
        float t = 0.f;
#pragma unroll
        for (int p = 0; p < DW_PXL; ++p) t += red[p][threadIdx.x];
        const int c = blockIdx.x * DW_CVB * EPC + threadIdx.x;
        if (c < g.C) {
          float* dst = dw + (long long)(r * SW + s) * g.C + c;
          if (use_atomic) atomicAdd(dst, t);
          else *dst += t;
        }
      }
      __syncthreads();
    }
  }
}

static inline int dw_blocks(long long n) {
  long long b = (n + 255) / 256;
  if (b > 16384) b = 16384;
  if (b < 1) b = 1;
  return (int)b;
}
// epc: channels per lane access - 8 (4 for fp32) when the channel count fills 16-byte vectors, 2 for an even count of
// 16-bit channels that does not (ShuffleNetV2's 58 / 116: one 4-byte access), 1 otherwise
// (variadic: the launch macro expands to <<<a, b, c, d>>>, whose commas must not split the argument on the way down)
#define DW_DISPATCH_16(epc, ...)                                         \
  if ((epc) == 8) { constexpr int EPC = 8; __VA_ARGS__; }                \
  else if ((epc) == 2) { constexpr int EPC = 2; __VA_ARGS__; }           \
  else { constexpr int EPC = 1; __VA_ARGS__; }
#define DW_DISPATCH(dtype, epc, ...)                                                           \
  switch (dtype) {                                                                             \
    case PDA_F32: { typedef float T; if ((epc) == 4) { constexpr int EPC = 4; __VA_ARGS__; } else { constexpr int EPC = 1; __VA_ARGS__; } } break; \
    case PDA_F16: { typedef f16 T; DW_DISPATCH_16(epc, __VA_ARGS__) } break;                   \
    case PDA_BF16: { typedef bf16 T; DW_DISPATCH_16(epc, __VA_ARGS__) } break;                 \
    default: return -6;                                                                        \
  }

// mode 0: fprop  (a = x [N,Hi,Wi,C],  b = w [R,S,C] in the activation dtype, out = y [N,Ho,Wo,C], bias fp32 or null)
// mode 1: dgrad  (a = dy [N,Ho,Wo,C], b = w [R,S,C],                          out = dx [N,Hi,Wi,C])
// mode 2: wgrad  (a = dy,             b = x,                                  out = dw fp32 [R,S,C], accumulated)
// slabs (wgrad only): number of pixel slabs; 0 = enough to fill the chip, 1 = no atomics (bit-reproducible).
PDA_EXPORT int pda_depthwise_conv(int dtype, int mode, const void* a, const void* b, const float* bias, void* out,
                                  int N, int Hi, int Wi, int C, int Ho, int Wo, int R, int S, int sh, int sw,
                                  int ph, int pw, int slabs, hipStream_t st) {
  if (N < 0 || Hi < 1 || Wi < 1 || C < 1 || Ho < 1 || Wo < 1 || R < 1 || S < 1 || sh < 1 || sw < 1 || ph < 0 ||
      pw < 0)
    return -1;
  // the output size must be the one the geometry implies: the kernels index with it
  if (Ho != (Hi + 2 * ph - R) / sh + 1 || Wo != (Wi + 2 * pw - S) / sw + 1 || Hi + 2 * ph < R || Wi + 2 * pw < S)
    return -2;
  const uintptr_t ptrs = reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(out);
  int epc = dtype == PDA_F32 ? 4 : 8;
  if (C % epc != 0 || (ptrs % 16) != 0) epc = (dtype != PDA_F32 && C % 2 == 0 && (ptrs % 4) == 0) ? 2 : 1;
  DwGeom g{N, Hi, Wi, C, Ho, Wo, R, S, sh, sw, ph, pw};
  if (N == 0) return 0;
  if (mode == 0) {
    const long long total = (long long)N * Ho * Wo * (C / epc);
    DW_DISPATCH(dtype, epc, hipLaunchKernelGGL((dw_fprop_kernel<T, EPC>), dim3(dw_blocks(total)), dim3(256), 0, st,
                                               (const T*)a, (const T*)b, bias, (T*)out, g));
  } else if (mode == 1) {
    const long long total = (long long)N * Hi * Wi * (C / epc);
    DW_DISPATCH(dtype, epc, hipLaunchKernelGGL((dw_dgrad_kernel<T, EPC>), dim3(dw_blocks(total)), dim3(256), 0, st,
                                               (const T*)a, (const T*)b, (T*)out, g));
  } else if (mode == 2) {
    const int e = epc;
    const int gx = (C / e + DW_CVB - 1) / DW_CVB;
    const long long M = (long long)N * Ho * Wo;
    long long ns = slabs;
    if (ns <= 0) {
      ns = 2048 / gx;                       // ~8 workgroups per CU
      const long long cap = (M + 255) / 256;  // at least 8 pixels per pixel lane
      if (ns > cap) ns = cap;
      if (ns < 1) ns = 1;
    }
    if (ns > 65535) ns = 65535;
    const long long slab = (M + ns - 1) / ns;
    const unsigned gy = (unsigned)((M + slab - 1) / slab);
    const int use_atomic = gy > 1;
    if (S == 3) {
      DW_DISPATCH(dtype, epc, hipLaunchKernelGGL((dw_wgrad_row_kernel<T, EPC, 3>), dim3(gx, gy), dim3(256), 0, st,
                                                 (const T*)a, (const T*)b, (float*)out, g, slab, use_atomic));
    } else if (S == 5) {
      DW_DISPATCH(dtype, epc, hipLaunchKernelGGL((dw_wgrad_row_kernel<T, EPC, 5>), dim3(gx, gy), dim3(256), 0, st,
                                                 (const T*)a, (const T*)b, (float*)out, g, slab, use_atomic));
    } else {
      DW_DISPATCH(dtype, epc, hipLaunchKernelGGL((dw_wgrad_kernel<T, EPC>), dim3(gx, gy), dim3(256), 0, st,
                                                 (const T*)a, (const T*)b, (float*)out, g, slab, use_atomic));
    }
  } else {
    return -3;
  }
  return pda_check_launch();
}
