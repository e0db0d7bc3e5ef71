// Backward of the ResNet stem's BatchNorm + convolution in one pass: dy0 = A*g + B*y0 + K with g = dact masked by the
// stem's ReLU (recomputed from y0), and dW += dy0^T * X, where X is the stem's implicit-GEMM operand (7 row taps x 8
// pixels x 4 padded channels of the zero-bordered input image), for 64 output channels in fp16 / bf16 on gfx950.
//
// The stem has no input gradient, so its backward is the BatchNorm backward apply (read dact and y0, write dy0: three
// passes over a 112x112x64 tensor, 642 MB at batch 400) followed by a weight gradient that reads dy0 once more.  Here
// a workgroup forms the dy0 rows of 64 pixels in LDS, gathers the same pixels' operand rows next to them and runs the
// weight-gradient MFMAs (v_mfma_f32_32x32x16 through transposing LDS reads) on the pair; the 64 x 224 accumulators
// of the persistent workgroup go to memory once at the end.  dy0 is never stored: two passes instead of four.
// (Reference trigger: `loss.backward()` reaching torchvision's conv1 / bn1, /root/reference/2.distributed.py:231.)
#include "common.h"

namespace {

struct StemArgs {
  const void* dact;    // [M][64] gradient of relu(bn(y0)) (the max-pool's input gradient)
  const void* y0;      // [M][64] conv output
  const float* coef;   // [5][64]: A, B, K, scale, shift
  const void* xp;      // zero-bordered NHWC4 image
  float* dw;           // [64][R * 32] fp32, accumulated into
  long long M;
  long long sn, sh, sw;   // element strides of xp per image / row / output column
  int Ho, Wo, R, stride_h;
  int ntiles;
};

template <typename T> struct StemMfma;
template <> struct StemMfma<f16> {
  typedef f16x8 frag;
  static __device__ __forceinline__ f32x16 m32(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};
template <> struct StemMfma<bf16> {
  typedef bf16x8 frag;
  static __device__ __forceinline__ f32x16 m32(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};

template <typename T>
__device__ __forceinline__ typename StemMfma<T>::frag stem_tr_read(const unsigned char* p0, const unsigned char* p1) {
  typedef s16x4 __attribute__((address_space(3))) * lp;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(p0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(p1));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(typename StemMfma<T>::frag, v);
}

// R: filter rows (taps of 32 contiguous values each), R <= 8
// 256 threads per workgroup: one wave per SIMD, so that two or three workgroups share a compute unit and one's
// element-wise phase overlaps another's MFMA phase (the phases of one workgroup are separated by barriers; with 512
// threads and one workgroup per compute unit the kernel took 580 us instead of 450).
// NSET: tiles in flight per workgroup (register sets); CLDS: BatchNorm coefficients read from an LDS table per tile
// instead of held in 40 registers (occupancy)
template <typename T, int R, int NSET, bool CLDS>
__global__ void __launch_bounds__(256)
stem_bwd_kernel(StemArgs a) {
  constexpr int CO = 64, BM = 64, NT = 256;
  constexpr int NVD = BM * 8 / NT;               // dact / y0 vectors per thread
  constexpr int RPP = NT / 8;                    // pixel rows per element-wise pass
  constexpr int CBW = 8 / (NT / 64);             // 32-column blocks of dW per wave
  // dy image [pixel][channel]: consecutive pixel rows 32 bytes apart modulo 256 (transposing reads: see
  // bottleneck_tail_bwd.hip).  X is kept as the R raw input rows of the tile, BM + 3 chunks of 16 bytes (one output
  // column's 4 padded channels x stride 2) each: pixel p's 32 operand values of tap row r are the 64 bytes at chunk p
  // of row r, so the operand matrix's rows overlap, 16 bytes apart, and are never materialised (a quarter of the LDS
  // writes and of the gather requests of a [pixel][R * 32] image).
  constexpr int PD = CO * 2 + 32;
  constexpr int XCH = BM + 3, PXR = (XCH * 16 + 63) / 64 * 64;
  constexpr int NVX = (R * XCH + NT - 1) / NT;   // X chunks per thread
  constexpr int OFF_X = BM * PD, OFF_C = OFF_X + R * PXR, LDS_BYTES = OFF_C + (CLDS ? 5 * CO * 4 : 0);
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
  typedef typename StemMfma<T>::frag frag;
  const int t = threadIdx.x, lane = t & 63;
  const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
  const T* __restrict__ DA = reinterpret_cast<const T*>(a.dact);
  const T* __restrict__ Y0 = reinterpret_cast<const T*>(a.y0);
  const T* __restrict__ XP = reinterpret_cast<const T*>(a.xp);
  unsigned char* imgD = lds;
  unsigned char* imgX = lds + OFF_X;
  const float* ctab = reinterpret_cast<const float*>(lds + OFF_C);
  if (CLDS)
    for (int i = t; i < 5 * CO; i += NT) reinterpret_cast<float*>(lds + OFF_C)[i] = a.coef[i];

  // element-wise role: thread -> channel group t & 7 of pixel rows t >> 3 (+ RPP)
  const int cv = t & 7, row = t >> 3;
  float cA[8], cB[8], cK[8], sc[8], sf[8];
  if (!CLDS) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      cA[e] = a.coef[cv * 8 + e];
      cB[e] = a.coef[CO + cv * 8 + e];
      cK[e] = a.coef[2 * CO + cv * 8 + e];
      sc[e] = a.coef[3 * CO + cv * 8 + e];
      sf[e] = a.coef[4 * CO + cv * 8 + e];
    }
  }
  // A tile is a 64-pixel segment of ONE output row: every address of a tile is a scalar base plus a per-thread
  // constant (with tiles of 64 consecutive pixels the row / image carries and 64-bit products per gathered vector
  // made the kernel VALU-bound: 2.2 us per tile per CU for 16 KB of new data).
  // gather role: chunk v = t + NT * u of the tile -> tap row v / XCH, chunk v % XCH
  int gch[NVX], goff[NVX], gdst[NVX];
#pragma unroll
  for (int u = 0; u < NVX; ++u) {
    const int v = t + NT * u;
    const int r = v / XCH, c = v - r * XCH;
    gch[u] = v < R * XCH ? c : 1 << 20;                                    // (never inside a row)
    goff[u] = r * (int)a.sh + c * 8;                                       // elements
    gdst[u] = r * PXR + c * 16;                                            // bytes
  }
  const int tpr = (a.Wo + BM - 1) / BM;                                    // tiles per output row
  // wgrad role (32x32x16): wave -> dW columns (wid * CBW + j) * 32 .. + 31 (the last block is padding), both 32-row blocks
  const int li = lane & 15, lg = lane >> 4;
  const int prow0 = 8 * (lg >> 1) + (li >> 2);
  const int csub = 16 * (lg & 1) + 4 * (li & 3);
  f32x16 wacc[CBW][2];
#pragma unroll
  for (int j = 0; j < CBW; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < 16; ++g) wacc[j][i][g] = 0.f;

  Vec16<T> davs[NSET][NVD], y0vs[NSET][NVD], xvs[NSET][NVX];
  auto fetch = [&](int tile, Vec16<T>(&dav)[NVD], Vec16<T>(&y0v)[NVD], Vec16<T>(&xv)[NVX]) {
    // scalar: output row (image * Ho + row) and first column of the tile
    const int orow = __builtin_amdgcn_readfirstlane(tile / tpr);
    const int wo0 = __builtin_amdgcn_readfirstlane((tile - orow * tpr) * BM);
    const int n = __builtin_amdgcn_readfirstlane(orow / a.Ho);
    const int ho = orow - n * a.Ho;
    const int left = a.Wo - wo0;                                           // valid pixels of this tile
    const T* __restrict__ da = DA + ((long long)orow * a.Wo + wo0) * CO;
    const T* __restrict__ yy = Y0 + ((long long)orow * a.Wo + wo0) * CO;
    const T* __restrict__ xx = XP + (long long)n * a.sn + (long long)ho * a.stride_h * a.sh + (long long)wo0 * a.sw;
#pragma unroll
    for (int u = 0; u < NVD; ++u) {
      const int r = row + RPP * u;
      const int o = (r < left ? r : 0) * CO + cv * 8;                      // pixels past the row read pixel 0 and are zeroed
      dav[u] = load16(da + o);
      y0v[u] = load16(yy + o);
    }
    // (chunks past the row's last pixel read chunk 0: finite values, multiplied by dy rows that are zero)
#pragma unroll
    for (int u = 0; u < NVX; ++u) xv[u] = load16(xx + (gch[u] < left + 3 ? goff[u] : 0));
  };

  auto process = [&](int tile, bool live, Vec16<T>(&dav)[NVD], Vec16<T>(&y0v)[NVD], Vec16<T>(&xv)[NVX]) {
    const int left = live ? a.Wo - (tile - (tile / tpr) * tpr) * BM : 0;   // valid pixels of this tile (none: all zeros)
#pragma unroll
    for (int u = 0; u < NVX; ++u)
      if (gch[u] < XCH) *reinterpret_cast<Vec16<T>*>(imgX + gdst[u]) = xv[u];
    if (CLDS) {
      // (the offset is made opaque per tile: otherwise the reads are hoisted out of the loop and the coefficients
      // occupy their 40 registers for the whole kernel again)
      int c8 = cv * 8;
      asm volatile("" : "+v"(c8));
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const f32x4 va = *reinterpret_cast<const f32x4*>(ctab + c8 + h * 4);
        const f32x4 vb = *reinterpret_cast<const f32x4*>(ctab + CO + c8 + h * 4);
        const f32x4 vk = *reinterpret_cast<const f32x4*>(ctab + 2 * CO + c8 + h * 4);
        const f32x4 vs = *reinterpret_cast<const f32x4*>(ctab + 3 * CO + c8 + h * 4);
        const f32x4 vf = *reinterpret_cast<const f32x4*>(ctab + 4 * CO + c8 + h * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          cA[h * 4 + e] = va[e]; cB[h * 4 + e] = vb[e]; cK[h * 4 + e] = vk[e]; sc[h * 4 + e] = vs[e]; sf[h * 4 + e] = vf[e];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < NVD; ++u) {
      const int r = row + RPP * u;
      const bool ok = r < left;
      Vec16<T> o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float yv = ElemTraits<T>::to_f32(y0v[u].v[e]);
        float gd = ElemTraits<T>::to_f32(dav[u].v[e]);
        gd = (yv * sc[e] + sf[e] > 0.f) ? gd : 0.f;          // ReLU backward, mask recomputed as bn_bwd_apply_kernel does
        const float v = cA[e] * gd + cB[e] * yv + cK[e];
        o.v[e] = ElemTraits<T>::from_f32(ok ? v : 0.f);
      }
      *reinterpret_cast<Vec16<T>*>(imgD + r * PD + cv * 16) = o;
    }
    __syncthreads();
    // this register set is free again.  The request is unconditional (past the end: the last tile once more, unused):
    // behind a branch the compiler cannot count the younger requests at the top of the loop and waits for all of them,
    // which serialises every tile behind its own memory latency
    fetch(min(tile + NSET * (int)gridDim.x, a.ntiles - 1), dav, y0v, xv);
    __builtin_amdgcn_sched_barrier(0);   // (otherwise the requests sink to the end of the MFMA phase)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int pr = kk * 16 + prow0;
      frag fa[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        fa[i] = stem_tr_read<T>(imgD + pr * PD + (i * 32 + csub) * 2, imgD + (pr + 4) * PD + (i * 32 + csub) * 2);
#pragma unroll
      for (int j = 0; j < CBW; ++j) {
        // tap row = 32-column block of dW; the block past the last tap repeats it and is dropped at the end (a
        // branch here keeps the compiler from issuing a tile's LDS reads ahead of its MFMAs)
        const int r = min(wid * CBW + j, R - 1);
        const unsigned char* xr = imgX + r * PXR + csub * 2;
        const frag fb = stem_tr_read<T>(xr + pr * 16, xr + (pr + 4) * 16);
#pragma unroll
        for (int i = 0; i < 2; ++i) wacc[j][i] = StemMfma<T>::m32(fa[i], fb, wacc[j][i]);
      }
    }
    __syncthreads();   // every wave is done reading this tile's images
  };

  // tiles blockIdx.x + k * gridDim.x; a workgroup always runs a multiple of NSET of them (the last ones possibly
  // repeats of the final tile with `live` false: their dy image is zeros) so that the loop body is branch-free
  // Workgroups are dealt round-robin to the 8 XCDs, each with an L2 of its own: the ones of an XCD take neighbouring
  // tiles, so that the 5 of 7 input rows shared by consecutive output rows are read into one L2 only.
  int tile = (gridDim.x & 7) == 0 ? (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3) : blockIdx.x;
#pragma unroll
  for (int s = 0; s < NSET; ++s) fetch(min(tile + s * (int)gridDim.x, a.ntiles - 1), davs[s], y0vs[s], xvs[s]);
  __syncthreads();   // coefficient table
  for (; tile < a.ntiles; tile += NSET * gridDim.x) {
#pragma unroll
    for (int s = 0; s < NSET; ++s)
      process(tile + s * (int)gridDim.x, tile + s * (int)gridDim.x < a.ntiles, davs[s], y0vs[s], xvs[s]);
  }

  // C/D map of the 32x32 MFMA: column lane & 31, rows (g & 3) + 8 * (g >> 2) + 4 * (lane >> 5)
  const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int j = 0; j < CBW; ++j) {
    const int col = (wid * CBW + j) * 32 + l31;
    if (col < R * 32) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          const int co = i * 32 + (g & 3) + 8 * (g >> 2) + 4 * lh;
          atomicAdd(a.dw + co * (R * 32) + col, wacc[j][i][g]);
        }
    }
  }
#endif
}

template <typename T, int NSET, bool CLDS>
void stem_launch(const StemArgs& a, int nwg, hipStream_t stream) {
  hipLaunchKernelGGL((stem_bwd_kernel<T, 7, NSET, CLDS>), dim3(nwg), dim3(256), 0, stream, a);
}

template <typename T>
void stem_launch_variant(int variant, const StemArgs& a, int nwg, hipStream_t stream) {
  switch (variant) {
    case 0: stem_launch<T, 2, false>(a, nwg, stream); break;
    case 1: stem_launch<T, 1, false>(a, nwg, stream); break;
    case 2: stem_launch<T, 2, true>(a, nwg, stream); break;
    case 3: stem_launch<T, 1, true>(a, nwg, stream); break;
    default: stem_launch<T, 3, false>(a, nwg, stream); break;
  }
}

}  // namespace

// dw[64][R][1][32] (fp32, accumulated into with atomics) += the stem's weight gradient for dy0 = BatchNorm backward of
// (dact, y0) with coef = [5][64] (A, B, K, scale, shift; the ReLU mask is recomputed from y0).  xp / strides: the
// gather geometry of the stem's implicit GEMM (ops/fused.py::_stem_geometry): element strides per image, per image
// row and per output column; stride_h: the conv's row stride.  <0: geometry not covered.
PDA_EXPORT int pda_stem_bwd_fused(int dtype, const void* dact, const void* y0, const float* coef, const void* xp,
                                  float* dw, long long M, int Co, int R, int Ho, int Wo, long long sn, long long sh,
                                  long long sw, int stride_h, hipStream_t stream) {
  if (dtype != PDA_F16 && dtype != PDA_BF16) return -6;
  if (Co != 64 || R != 7 || sw != 8) return -1;
  if (M <= 0) return 0;
  if (Ho <= 0 || Wo <= 0 || M % ((long long)Ho * Wo) != 0) return -3;
  if ((((uintptr_t)dact | (uintptr_t)y0 | (uintptr_t)xp) & 15) != 0 || (sn % 8) || (sh % 8) || (sw % 8)) return -5;
  if (!dact || !y0 || !coef || !xp || !dw) return -11;
  if (sh > 0x7fffffffLL / 8) return -4;
  StemArgs a;
  a.dact = dact; a.y0 = y0; a.coef = coef; a.xp = xp; a.dw = dw; a.M = M; a.sn = sn; a.sh = sh; a.sw = sw;
  a.Ho = Ho; a.Wo = Wo; a.R = R; a.stride_h = stride_h;
  const long long tiles = (M / Wo) * ((Wo + 63) / 64);   // 64-pixel segments of output rows
  if (tiles > 0x7fffffffLL || (long long)63 * sw + 6 * sh + 32 > 0x7fffffffLL) return -4;
  a.ntiles = (int)tiles;
  static const int max_wg = [] { const char* e = getenv("PDA_STEM_BWD_WGS"); return e ? atoi(e) : 768; }();
  static const int variant = [] { const char* e = getenv("PDA_STEM_BWD_VARIANT"); return e ? atoi(e) : 1; }();
  const int nwg = (int)(tiles < max_wg ? tiles : max_wg);
  if (dtype == PDA_F16) stem_launch_variant<f16>(variant, a, nwg, stream);
  else stem_launch_variant<bf16>(variant, a, nwg, stream);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
