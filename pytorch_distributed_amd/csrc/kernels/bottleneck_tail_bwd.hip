// Backward of the tail of a ResNet bottleneck, "BatchNorm3 backward apply + conv3 dgrad + conv3 wgrad" in ONE pass, for
// the first stage's geometry (conv3 = 1x1, 64 -> 256 channels) in fp16 / bf16 on gfx950.
//
// On the 56x56 stage at batch 400 a 256-channel tensor is 642 MB.  The separate kernels move five of them per block:
// the BatchNorm backward reads dz (gradient of the block output, already masked) and y3 (conv3's output) and writes
// dy3 = A*dz + B*y3 + K, then conv3's dgrad and wgrad each read dy3 again - 0.76 ms per block for work that is bound
// by nothing but those bytes.  Here a workgroup takes 64 pixels at a time, forms their dy3 rows in LDS (fp32
// arithmetic and one rounding to the 16-bit type, as bn_bwd_apply_kernel stores them) and uses the tile twice while
// it is on chip:
//   * dX[64][64] = dy3[64][256] * W3[256][64] on v_mfma_f32_16x16x32 (the weight fragments of a wave's 16 output
//     columns live in 32 registers for the whole kernel), masked by conv3's normalise-on-load input
//     a2 = relu(y2 * scale + shift) > 0, with the two sums BatchNorm2's backward needs (sum dX, sum dX * y2) taken
//     while the tile is in registers - what the dgrad kernel's fused epilogue does;
//   * dW3[256][64] += dy3^T * a2 on v_mfma_f32_32x32x16 through transposing LDS reads, accumulated in registers over
//     all tiles of the (persistent) workgroup and added to the gradient once at the end.
// Two tensors are read (dz, y3) plus the 64-channel y2, one 64-channel tensor is written: 2.25 passes instead of 5.75.
// The loads of the next tile are requested before the MFMAs of the current one.
// (Reference trigger: `loss.backward()` through torchvision's Bottleneck, /root/reference/2.distributed.py:231.)
#include "common.h"

namespace {

struct TailArgs {
  const void* dz;        // [M][256]
  const void* y3;        // [M][256]
  const float* coef;     // [>=3][256]: A, B, K of BatchNorm3's backward
  const void* x2;        // [M][64] conv2's raw output (before BatchNorm2)
  const void* in_scale;  // [64] of x2's type: a2 = relu(x2 * in_scale + in_shift); null: x2 is the activation itself,
  const void* in_shift;  // dX is not masked and no sums are taken (a shortcut's 1x1 conv + BatchNorm)
  const void* wt;        // [64][256] transposed weight (CRSK)
  void* dx;              // [M][64]
  float* dw;             // [256][64] fp32, accumulated into (kernels with the weight gradient)
  void* dy;              // [M][CO]: dy3 stored for a separate weight-gradient kernel (kernels without it)
  float* sums;           // [2][rows][64]: sum dX, sum dX * x2 per workgroup
  long long M;
  int ntiles;
};

template <typename T> struct TailMfma;
template <> struct TailMfma<f16> {
  typedef f16x8 frag;
  static __device__ __forceinline__ f32x4 m16(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x16 m32(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};
template <> struct TailMfma<bf16> {
  typedef bf16x8 frag;
  static __device__ __forceinline__ f32x4 m16(frag a, frag b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x16 m32(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};

__device__ __forceinline__ f16x8 tail_affine_relu(f16x8 v, f16x8 sc, f16x8 sf) {
  const f16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
  return __builtin_elementwise_max(__builtin_elementwise_fma(v, sc, sf), zero);
}
__device__ __forceinline__ bf16x8 tail_affine_relu(bf16x8 v, bf16x8 sc, bf16x8 sf) {
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (bf16)fmaxf(fmaf((float)v[e], (float)sc[e], (float)sf[e]), 0.f);
  return o;
}

template <typename T>
__device__ __forceinline__ typename TailMfma<T>::frag tail_tr_read(const unsigned char* p0, const unsigned char* p1) {
  typedef s16x4 __attribute__((address_space(3))) * lp;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(p0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(p1));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(typename TailMfma<T>::frag, v);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also fences global memory: every wave would wait
// for the tile's dX / dy3 stores (a trip to memory) before the next tile's images may be written.
__device__ __forceinline__ void tail_lds_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_waitcnt(15 | (3 << 14) | (7 << 4) | (0 << 8));   // lgkmcnt(0) only
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}

// CO / CI: conv3's output / input channels (256 / 64 or 512 / 128).  WGRAD: the weight gradient is part of the pass;
// without it (the 128 -> 512 stage: 512 x 128 fp32 accumulators plus the weight fragments do not fit a workgroup's
// registers) dy3 is also stored and the ring weight-gradient kernel reads it - still one pass over dz / y3 less.
template <typename T, int CO, int CI, bool WGRAD>
__global__ void __launch_bounds__(512)
tail_bwd_kernel(TailArgs a) {
  constexpr int BM = 64, NT = 512;
  static_assert((CO == 256 && CI == 64) || (CO == 512 && CI == 128 && !WGRAD), "geometries this kernel is laid out for");
  // LDS images [pixel][channel].  Pitches: a transposing read touches 4 pixel rows x 32 bytes per 16-lane group, so
  // consecutive rows must start 32 bytes (8 banks) apart modulo 256: 2 * CO + 32, 2 * CI + 32.
  constexpr int PD = CO * 2 + 32, PA = CI * 2 + 32, PS = CI * 2 + 16;
  constexpr int OFF_A = BM * PD, OFF_S = OFF_A + (WGRAD ? BM * PA : 0), OFF_C = OFF_S + BM * PS;
  constexpr int LDS_BYTES = OFF_C + 3 * CO * 4;              // + the A, B, K table of BatchNorm3's backward
  constexpr int VD = CO / 8, VX = CI / 8;                    // 16-byte vectors per pixel row
  constexpr int NVD = BM * VD / NT, RSD = NT / VD;           // dy3 vectors per thread, pixel-row step between them
  constexpr int NVX = BM * VX / NT, RSX = NT / VX;           // x2 / dX vectors per thread
  static_assert(NVD >= 1 && NVX >= 1 && NVD * RSD == BM && NVX * RSX == BM, "element-wise roles tile the 64 pixel rows");
  static_assert(OFF_C >= 16 * NVX * NT * 4, "the final reduction of the sums reuses the images");
  constexpr int KS = CO / 32;                                // 32-deep dgrad steps
  constexpr int NCB = CI / 16, WPC = 8 / NCB, RBW = 4 / WPC; // dgrad: column blocks, waves per column block, row blocks per wave
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
  typedef typename TailMfma<T>::frag frag;
  const int t = threadIdx.x, lane = t & 63;
  const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
  const T* __restrict__ DZ = reinterpret_cast<const T*>(a.dz);
  const T* __restrict__ Y3 = reinterpret_cast<const T*>(a.y3);
  const T* __restrict__ X2 = reinterpret_cast<const T*>(a.x2);
  T* __restrict__ DX = reinterpret_cast<T*>(a.dx);
  T* __restrict__ DY = reinterpret_cast<T*>(a.dy);

  // ---- element-wise roles.  dy3: thread -> 8-channel group t % VD of pixel rows t / VD + RSD u;
  // x2 / dX: thread -> channel group t % VX of pixel rows t / VX + RSX v
  const int cvd = t % VD, rowd = t / VD;
  const int cvx = t % VX, rowx = t / VX;
  // (the coefficients of a thread's eight channels are re-read from LDS per tile: 24 registers this kernel does not have)
  float* ctab = reinterpret_cast<float*>(lds + OFF_C);
  for (int i = t; i < 3 * CO; i += NT) ctab[i] = a.coef[i];
  const bool plain = a.in_scale == nullptr;   // (uniform)
  frag xsc = {}, xsf = {};
  if (!plain) {
    xsc = *reinterpret_cast<const frag*>(reinterpret_cast<const T*>(a.in_scale) + cvx * 8);
    xsf = *reinterpret_cast<const frag*>(reinterpret_cast<const T*>(a.in_shift) + cvx * 8);
  }

  // ---- MFMA roles.  dgrad (16x16x32): wave -> output columns cb * 16 .. + 15, pixel-row blocks rb0 .. rb0 + RBW - 1;
  // operand layout: lane -> row / column lane & 15, k = 8 * (lane >> 4) .. + 7 of a 32-deep step
  const int cb = wid % NCB, rb0 = (wid / NCB) * RBW;
  const int l15 = lane & 15, l4 = lane >> 4;
  frag wfr[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
    wfr[ks] = *reinterpret_cast<const frag*>(reinterpret_cast<const T*>(a.wt) + (cb * 16 + l15) * CO + ks * 32 + 8 * l4);
  // wgrad (32x32x16): wave -> dW rows (conv3 output channels) wid * 32 .. + 31, all 64 columns; fragments by
  // transposing reads: 16-lane group lg -> channel sub-block 16 * (lg & 1), pixel rows 8 * (lg >> 1) + {0..3}, {4..7}
  const int li = lane & 15, lg = lane >> 4;
  const int prow0 = 8 * (lg >> 1) + (li >> 2);
  const int csub = 16 * (lg & 1) + 4 * (li & 3);
  f32x16 wacc[WGRAD ? 2 : 1];
#pragma unroll
  for (int j = 0; j < (WGRAD ? 2 : 1); ++j)
#pragma unroll
    for (int g = 0; g < 16; ++g) wacc[j][g] = 0.f;
  float s1[NVX][8], s2[NVX][8];
#pragma unroll
  for (int v = 0; v < NVX; ++v)
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[v][e] = 0.f; s2[v][e] = 0.f; }

  unsigned char* imgD = lds;
  unsigned char* imgA = lds + OFF_A;
  unsigned char* stg = lds + OFF_S;

  Vec16<T> dzv[NVD], y3v[NVD], x2v[NVX];
  auto fetch = [&](int tile) {
    const long long m0 = (long long)tile * BM;
#pragma unroll
    for (int u = 0; u < NVD; ++u) {
      const long long m = m0 + rowd + RSD * u;
      const long long o = (m < a.M ? m : 0) * CO + cvd * 8;   // rows past M read row 0 and are zeroed below
      dzv[u] = load16(DZ + o);
      y3v[u] = load16(Y3 + o);
    }
#pragma unroll
    for (int v = 0; v < NVX; ++v) {
      const long long mx = m0 + rowx + RSX * v;
      x2v[v] = load16(X2 + (mx < a.M ? mx : 0) * CI + cvx * 8);
    }
  };

  int tile = blockIdx.x;
  if (tile < a.ntiles) fetch(tile);
  __syncthreads();   // coefficient table
  for (; tile < a.ntiles; tile += gridDim.x) {
    const long long m0 = (long long)tile * BM;
    // ---- dy3 rows and a2 into LDS (fp32 arithmetic, one rounding: as bn_bwd_apply_kernel)
    float cA[8], cB[8], cK[8];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const f32x4 va = *reinterpret_cast<const f32x4*>(ctab + cvd * 8 + 4 * h);
      const f32x4 vb = *reinterpret_cast<const f32x4*>(ctab + CO + cvd * 8 + 4 * h);
      const f32x4 vk = *reinterpret_cast<const f32x4*>(ctab + 2 * CO + cvd * 8 + 4 * h);
#pragma unroll
      for (int e = 0; e < 4; ++e) { cA[4 * h + e] = va[e]; cB[4 * h + e] = vb[e]; cK[4 * h + e] = vk[e]; }
    }
#pragma unroll
    for (int u = 0; u < NVD; ++u) {
      const bool ok = m0 + rowd + RSD * u < a.M;
      Vec16<T> o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float v = cA[e] * ElemTraits<T>::to_f32(dzv[u].v[e]) + cB[e] * ElemTraits<T>::to_f32(y3v[u].v[e]) + cK[e];
        o.v[e] = ElemTraits<T>::from_f32(ok ? v : 0.f);
      }
      *reinterpret_cast<Vec16<T>*>(imgD + (rowd + RSD * u) * PD + cvd * 16) = o;
      if (!WGRAD && ok) store16(DY + (m0 + rowd + RSD * u) * CO + cvd * 8, o);
    }
    Vec16<T> x2raw[NVX];
    frag a2[NVX];
    bool okx[NVX];
#pragma unroll
    for (int v = 0; v < NVX; ++v) {
      okx[v] = m0 + rowx + RSX * v < a.M;
      x2raw[v] = x2v[v];
      a2[v] = plain ? __builtin_bit_cast(frag, x2v[v]) : tail_affine_relu(__builtin_bit_cast(frag, x2v[v]), xsc, xsf);
      if (!okx[v]) {
#pragma unroll
        for (int e = 0; e < 8; ++e) a2[v][e] = 0;
      }
      if (WGRAD) *reinterpret_cast<frag*>(imgA + (rowx + RSX * v) * PA + cvx * 16) = a2[v];
    }
    tail_lds_barrier();
    // ---- the next tile's operands travel while this one computes
    if (tile + (int)gridDim.x < a.ntiles) fetch(tile + gridDim.x);

    // ---- dgrad: RBW 16 x 16 blocks of dX over K = CO
    f32x4 dacc[RBW];
#pragma unroll
    for (int r = 0; r < RBW; ++r) dacc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int r = 0; r < RBW; ++r) {
        const frag af = *reinterpret_cast<const frag*>(imgD + ((rb0 + r) * 16 + l15) * PD + (ks * 32 + 8 * l4) * 2);
        dacc[r] = TailMfma<T>::m16(af, wfr[ks], dacc[r]);
      }
    }
    if constexpr (WGRAD) {
      // ---- wgrad: dW rows wid * 32 .. + 31 over the tile's 64 pixels
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int pr = kk * 16 + prow0;
        const frag fa = tail_tr_read<T>(imgD + pr * PD + (wid * 32 + csub) * 2, imgD + (pr + 4) * PD + (wid * 32 + csub) * 2);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const frag fb = tail_tr_read<T>(imgA + pr * PA + (j * 32 + csub) * 2, imgA + (pr + 4) * PA + (j * 32 + csub) * 2);
          wacc[j] = TailMfma<T>::m32(fa, fb, wacc[j]);
        }
      }
    }
    // ---- dX tile to row vectors through LDS: C/D map of the 16x16 MFMA: column lane & 15, rows 4 * (lane >> 4) + e
#pragma unroll
    for (int r = 0; r < RBW; ++r)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        *reinterpret_cast<T*>(stg + ((rb0 + r) * 16 + 4 * l4 + e) * PS + (cb * 16 + l15) * 2) = ElemTraits<T>::from_f32(dacc[r][e]);
    tail_lds_barrier();   // also: every wave is done reading this tile's images
#pragma unroll
    for (int v = 0; v < NVX; ++v) {
      Vec16<T> dxv = *reinterpret_cast<const Vec16<T>*>(stg + (rowx + RSX * v) * PS + cvx * 16);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        // ReLU backward of the layer that produced a2: the mask is decided by the value the forward multiplied in
        const bool pos = plain || ElemTraits<T>::to_f32(a2[v][e]) > 0.f;
        if (!pos) dxv.v[e] = ElemTraits<T>::from_f32(0.f);
        const float g = ElemTraits<T>::to_f32(dxv.v[e]);
        s1[v][e] += g;
        s2[v][e] += g * ElemTraits<T>::to_f32(x2raw[v].v[e]);
      }
      if (okx[v]) store16(DX + (m0 + rowx + RSX * v) * CI + cvx * 8, dxv);
    }
  }

  if constexpr (WGRAD) {
    // ---- weight gradient: C/D map of the 32x32 MFMA: column lane & 31, rows (g & 3) + 8 * (g >> 2) + 4 * (lane >> 5)
    const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const int co = wid * 32 + (g & 3) + 8 * (g >> 2) + 4 * lh;
        atomicAdd(a.dw + co * CI + j * 32 + l31, wacc[j][g]);
      }
  }
  // ---- the producing BatchNorm's two sums: one row per workgroup
  if (plain) return;   // (uniform)
  __syncthreads();
  float* red = reinterpret_cast<float*>(lds);
#pragma unroll
  for (int v = 0; v < NVX; ++v)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[((0 * NVX + v) * 8 + e) * NT + t] = s1[v][e];
      red[((1 * NVX + v) * 8 + e) * NT + t] = s2[v][e];
    }
  __syncthreads();
  if (t < 2 * CI) {
    const int which = t / CI, c = t % CI;
    const int e = c & 7, cv = c >> 3;
    float acc = 0.f;
#pragma unroll
    for (int v = 0; v < NVX; ++v)
#pragma unroll 8
      for (int r = 0; r < RSX; ++r) acc += red[((which * NVX + v) * 8 + e) * NT + r * VX + cv];
    a.sums[((long long)which * gridDim.x + blockIdx.x) * CI + c] = acc;
  }
#endif
}

}  // namespace

// Workgroups (= rows of `sums`) the fused kernel uses for M pixels.
PDA_EXPORT int pda_bottleneck_tail_bwd_rows(long long M) {
  const long long tiles = (M + 63) / 64;
  return (int)(tiles < 512 ? tiles : 512);
}

// See the head of the file.  (Ci, Co) = (64, 256): dw (fp32 [256][64]) is accumulated into with float atomics (not for
// the run-to-run reproducible mode) and dy may be null.  (Ci, Co) = (128, 512): dy ([M][512]) receives the BatchNorm
// gradient for a separate weight-gradient kernel and dw is not touched.  Other geometries: <0, the caller runs the
// separate kernels.  sums: [2][pda_bottleneck_tail_bwd_rows(M)][Ci].  in_scale == in_shift == null: x2 is the
// activation itself (no transform, no mask on dx, no sums; sums may be null): a shortcut's 1x1 conv + BatchNorm.
PDA_EXPORT int pda_bottleneck_tail_bwd(int dtype, const void* dz, const void* y3, const float* coef, const void* x2,
                                       const void* in_scale, const void* in_shift, const void* wt, void* dx, float* dw,
                                       void* dy, float* sums, long long M, int Ci, int Co, hipStream_t stream) {
  if (dtype != PDA_F16 && dtype != PDA_BF16) return -6;
  const bool small = Ci == 64 && Co == 256, mid = Ci == 128 && Co == 512;
  if (!small && !mid) return -1;
  if (M <= 0) return 0;
  if (M > 0x7fffffffLL * 32) return -4;
  if ((((uintptr_t)dz | (uintptr_t)y3 | (uintptr_t)x2 | (uintptr_t)in_scale | (uintptr_t)in_shift | (uintptr_t)wt |
        (uintptr_t)dx | (uintptr_t)dy) & 15) != 0)
    return -5;
  if (!dz || !y3 || !coef || !x2 || !wt || !dx) return -11;
  if ((in_scale == nullptr) != (in_shift == nullptr) || (in_scale != nullptr && !sums)) return -11;
  if ((small && !dw) || (mid && !dy)) return -11;
  TailArgs a;
  a.dz = dz; a.y3 = y3; a.coef = coef; a.x2 = x2; a.in_scale = in_scale; a.in_shift = in_shift; a.wt = wt;
  a.dx = dx; a.dw = dw; a.dy = dy; a.sums = sums; a.M = M; a.ntiles = (int)((M + 63) / 64);
  const int nwg = pda_bottleneck_tail_bwd_rows(M);
  if (small) {
    if (dtype == PDA_F16) hipLaunchKernelGGL((tail_bwd_kernel<f16, 256, 64, true>), dim3(nwg), dim3(512), 0, stream, a);
    else hipLaunchKernelGGL((tail_bwd_kernel<bf16, 256, 64, true>), dim3(nwg), dim3(512), 0, stream, a);
  } else {
    if (dtype == PDA_F16) hipLaunchKernelGGL((tail_bwd_kernel<f16, 512, 128, false>), dim3(nwg), dim3(512), 0, stream, a);
    else hipLaunchKernelGGL((tail_bwd_kernel<bf16, 512, 128, false>), dim3(nwg), dim3(512), 0, stream, a);
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
