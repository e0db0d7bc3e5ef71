// Gradient of a max-pool INPUT element, gathered from the pooled gradient and the saved arg-max bytes:
// the sum over every window that covers the pixel and whose arg-max points back at it (no atomics,
// deterministic).
#pragma once
#include "common.h"

struct PoolGeom {
  int H, W, Ho, Wo, k, s, p;  // input / output size, window, stride, padding
};

// r = (n*H + h)*W + w (input pixel row), cv = channel vector, vpr = vectors per pixel.
template <typename T, int EPC>
__device__ __forceinline__ void pool_grad_gather(const T* __restrict__ dpool, const unsigned char* __restrict__ idx,
                                                 const PoolGeom& g, long long r, int cv, int vpr, float (&acc)[EPC]) {
  const int w = (int)(r % g.W);
  const long long r2 = r / g.W;
  const int h = (int)(r2 % g.H);
  const long long n = r2 / g.H;
#pragma unroll
  for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
  // windows ho with ho*s - p <= h <= ho*s - p + k - 1
  int ho_lo = h + g.p - g.k + 1 + g.s - 1;
  ho_lo = ho_lo < 0 ? 0 : ho_lo / g.s;
  int ho_hi = (h + g.p) / g.s;
  if (ho_hi > g.Ho - 1) ho_hi = g.Ho - 1;
  int wo_lo = w + g.p - g.k + 1 + g.s - 1;
  wo_lo = wo_lo < 0 ? 0 : wo_lo / g.s;
  int wo_hi = (w + g.p) / g.s;
  if (wo_hi > g.Wo - 1) wo_hi = g.Wo - 1;
  for (int ho = ho_lo; ho <= ho_hi; ++ho)
    for (int wo = wo_lo; wo <= wo_hi; ++wo) {
      const int a = h - (ho * g.s - g.p), b = w - (wo * g.s - g.p);
      const unsigned char me = (unsigned char)(a * g.k + b);
      const long long o = (((n * g.Ho + ho) * g.Wo + wo) * vpr + cv) * EPC;
      VecN<T, EPC> gv = loadN<T, EPC>(dpool + o);
      unsigned char ib[EPC];
      if constexpr (EPC == 8) {
        *reinterpret_cast<u32x2*>(ib) = *reinterpret_cast<const u32x2*>(idx + o);
      } else if constexpr (EPC == 4) {
        *reinterpret_cast<unsigned int*>(ib) = *reinterpret_cast<const unsigned int*>(idx + o);
      } else {
#pragma unroll
        for (int e = 0; e < EPC; ++e) ib[e] = idx[o + e];
      }
#pragma unroll
      for (int e = 0; e < EPC; ++e)
        if (ib[e] == me) acc[e] += ElemTraits<T>::to_f32(gv.v[e]);
    }
}
