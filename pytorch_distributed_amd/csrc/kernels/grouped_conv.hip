// Grouped 3x3 convolution (ResNeXt: 32 groups of 4..32 channels, as many output as input channels) in fp16 / bf16 on
// gfx950: forward, input gradient and weight gradient on v_mfma_f32_32x32x16.
//
// A group is narrower than one MFMA tile, so the channels are cut into chunks of 32 (32 / Cg whole groups each) and
// every chunk is a dense 32 -> 32 convolution whose [32][9 taps][32] filter holds its groups on the diagonal: the MFMA
// work is 32 / Cg times the useful FLOPs (8x for 4-channel groups, none for 32-channel groups) instead of `groups`
// times for a block-diagonal filter over all channels, and the layer is bound by its memory traffic.
//
// fprop and the dgrad of stride-1 layers: gconv_lds_kernel (halo tile in LDS, see there).  The dgrad of stride-2 layers
// and PDA_GCONV_GATHER=1: no LDS.  A wave owns one chunk and keeps the chunk's whole filter in registers (18 K-steps x 4 VGPRs,
// in MFMA operand layout); per tile of 32 output pixels each lane gathers, for its pixel and each of the 9 taps, the
// 2 x 16 bytes that are its MFMA operand - buffer loads whose out-of-range offsets return the zero padding - and the
// tile is 18 MFMAs.  The operands are swapped (D = W * X^T) so that a lane ends up with 4 consecutive output channels
// of one pixel per register quad: 8-byte stores.  The four waves of a workgroup take four neighbouring chunks of the
// same pixels, so that the 128-byte lines they share are fetched once per compute unit.
//
// wgrad: the GEMM's K dimension is pixels, so both operands go through LDS as [pixel][channel] images and are read
// with the transposing ds_read_b64_tr_b16; the input is staged once per tile as its halo (rows x columns around up
// to 64 output pixels) and the 9 taps are 9 shifted views of it.  A wave accumulates its chunk's 9 x (32 x 32)
// gradients in 144 accumulator registers over all the tiles of its persistent workgroup and adds them to memory
// once at the end.
// (Reference role: torchvision's resnext50_32x4d / resnext101_32x8d behind `--arch`,
// /root/reference/1.dataparallel.py:23-24,97-102.)
#include "common.h"

namespace {

struct GArgs {
  const void* x;   // gathered tensor [N][Hi][Wi][C] (fprop: input; dgrad: dy)
  const void* w;   // [C/32][32 rows of the result][9][32 reduced channels]
  void* y;         // [N][Ho][Wo][C]
  long long x_bytes;
  int N, Hi, Wi, Ho, Wo, C;
  int stride_log2, pad;
  int M;                      // N * Ho * Wo
  unsigned long long mWo, mHo;   // floor(2^40 / d) + 1
  int ntiles;
};

template <typename T> struct GMfma;
template <> struct GMfma<f16> {
  typedef f16x8 frag;
  static __device__ __forceinline__ f32x16 m32(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};
template <> struct GMfma<bf16> {
  typedef bf16x8 frag;
  static __device__ __forceinline__ f32x16 m32(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};

// p / d for p < 2^31, d < 2^9 with m = floor(2^40 / d) + 1
__device__ __forceinline__ unsigned div_magic(unsigned p, unsigned long long m) {
  return (unsigned)(((unsigned long long)p * m) >> 40);
}

template <typename T, bool DGRAD>
__global__ void __launch_bounds__(256)
gconv_kernel(GArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef typename GMfma<T>::frag frag;
  const int lane = threadIdx.x & 63, l31 = lane & 31, lh = lane >> 5;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nchunks = a.C >> 5;
  const int ncg = (nchunks + 3) >> 2;
  const int cg = blockIdx.x % ncg, slot = blockIdx.x / ncg, nslots = gridDim.x / ncg;
  const int chunk = cg * 4 + wid;
  if (chunk >= nchunks) return;   // (the kernel has no barriers)
  const T* __restrict__ W = reinterpret_cast<const T*>(a.w) + ((long long)(chunk * 32 + l31) * 9) * 32 + lh * 8;
  frag wf[18];
#pragma unroll
  for (int t = 0; t < 18; ++t) {
    const Vec16<T> v = load16(W + (t >> 1) * 32 + (t & 1) * 16);
    wf[t] = __builtin_bit_cast(frag, v);
  }
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, (int)a.x_bytes, 0x00020000);
  const int cofs = (chunk * 32 + lh * 8) * 2;   // bytes
  const int st = 1 << a.stride_log2, stm = st - 1;
  T* __restrict__ Y = reinterpret_cast<T*>(a.y);

  for (int tile = slot; tile < a.ntiles; tile += nslots) {
    const int p = tile * 32 + l31;
    const bool valid = p < a.M;
    const unsigned pp = valid ? (unsigned)p : 0u;
    const unsigned nh = div_magic(pp, a.mWo), wo = pp - nh * a.Wo;
    const unsigned n = div_magic(nh, a.mHo), ho = nh - n * a.Ho;
    frag xf[18];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      int hi;
      bool okh;
      if (!DGRAD) {
        hi = (int)(ho << a.stride_log2) - a.pad + r;
        okh = valid && (unsigned)hi < (unsigned)a.Hi;
      } else {
        const int th = (int)ho + a.pad - r;
        hi = th >> a.stride_log2;
        okh = valid && th >= 0 && (th & stm) == 0 && hi < a.Hi;
      }
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        int wi;
        bool ok;
        if (!DGRAD) {
          wi = (int)(wo << a.stride_log2) - a.pad + s;
          ok = okh && (unsigned)wi < (unsigned)a.Wi;
        } else {
          const int tw = (int)wo + a.pad - s;
          wi = tw >> a.stride_log2;
          ok = okh && tw >= 0 && (tw & stm) == 0 && wi < a.Wi;
        }
        // (an offset past the tensor returns zeros: the padding, and the lanes past the last pixel)
        const int off = ok ? (int)((((unsigned)n * a.Hi + hi) * a.Wi + wi) * (unsigned)a.C * 2u) + cofs : -1;
        xf[(r * 3 + s) * 2] = __builtin_bit_cast(frag, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
        xf[(r * 3 + s) * 2 + 1] = __builtin_bit_cast(frag, __builtin_amdgcn_raw_buffer_load_b128(rsrc, ok ? off + 32 : -1, 0, 0));
      }
    }
    f32x16 acc;
#pragma unroll
    for (int g = 0; g < 16; ++g) acc[g] = 0.f;
#pragma unroll
    for (int t = 0; t < 18; ++t) acc = GMfma<T>::m32(wf[t], xf[t], acc);
    // C/D map of the 32x32 MFMA: column (pixel) lane & 31, rows (channels) (g & 3) + 8 * (g >> 2) + 4 * (lane >> 5)
    if (valid) {
      T* __restrict__ yo = Y + (long long)p * a.C + chunk * 32 + 4 * lh;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        VecN<T, 4> o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o.v[e] = ElemTraits<T>::from_f32(acc[q * 4 + e]);
        *reinterpret_cast<u32x2*>(yo + q * 8) = *reinterpret_cast<const u32x2*>(&o);
      }
    }
  }
#endif
}

// Forward-type convolution with the input staged through LDS: a workgroup loads the halo of up to 64 output pixels
// (TR rows x TW columns) for its 4 chunks once - each input value leaves L2 once per tile instead of once per tap -
// and a lane's MFMA operand of tap (r, s) is one ds_read_b128 at its pixel's halo position shifted by the tap.
// Pixel pitch 272 bytes: the 16 lanes of a quarter-wave read 16 consecutive pixels 4 banks apart, 16 bytes each.
// Serves fprop (stride 1 and 2) and, with each chunk's filter rotated by 180 degrees and transposed, the input gradient
// of a stride-1 layer.
struct GLArgs {
  const void* x;   // [N][Hi][Wi][C]
  const void* w;   // [C/32][32 result channels][9][32 reduced channels]
  void* y;         // [N][Ho][Wo][C]
  int N, Hi, Wi, Ho, Wo, C;
  int stride, pad;
  int TW, TR, HW, HR, tiles_w, tiles_h;
  unsigned mTW, mHW;
  int ntiles;
};

// Workgroup barrier that orders LDS traffic only (__syncthreads() also fences global memory: every wave would wait
// for its output stores of the tile before the next halo may be written).
__device__ __forceinline__ void gconv_lds_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_waitcnt(15 | (3 << 14) | (7 << 4) | (0 << 8));   // lgkmcnt(0) only
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}

template <typename T>
__global__ void __launch_bounds__(256)
gconv_lds_kernel(GLArgs a) {
  constexpr int NT = 256;
  constexpr int PP = 128 * 2 + 16;
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  typedef typename GMfma<T>::frag frag;
  const int t = threadIdx.x, lane = t & 63, l31 = lane & 31, lh = lane >> 5;
  const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
  const int nchunks = a.C >> 5;
  const int ncg = (nchunks + 3) >> 2;
  const int cg = blockIdx.x % ncg, slot = blockIdx.x / ncg, nslots = gridDim.x / ncg;
  const int chunk = cg * 4 + wid;
  const bool wave_live = chunk < nchunks;
  const int c0 = cg * 128;
  const int cvalid = min(128, a.C - c0) >> 3;
  const T* __restrict__ X = reinterpret_cast<const T*>(a.x);
  T* __restrict__ Y = reinterpret_cast<T*>(a.y);
  frag wf[18];
  if (wave_live) {
    const T* __restrict__ W = reinterpret_cast<const T*>(a.w) + ((long long)(chunk * 32 + l31) * 9) * 32 + lh * 8;
#pragma unroll
    for (int k = 0; k < 18; ++k) {
      const Vec16<T> v = load16(W + (k >> 1) * 32 + (k & 1) * 16);
      wf[k] = __builtin_bit_cast(frag, v);
    }
  } else {
#pragma unroll
    for (int k = 0; k < 18; ++k) wf[k] = frag{};
  }
  // the lane's two output pixels (one per 32-pixel block) inside a tile
  int hb[2], ptr[2], ptw[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int q = b * 32 + l31;
    ptr[b] = (int)(((unsigned)q * a.mTW) >> 20);
    ptw[b] = q - ptr[b] * a.TW;
    const bool inside = ptr[b] < a.TR;
    hb[b] = (inside ? ((ptr[b] * a.stride) * a.HW + ptw[b] * a.stride) * PP : 0) + (wid * 32 + lh * 8) * 2;
    if (!inside) ptr[b] = 1 << 20;   // (never a pixel of the tensor)
  }
  const int tiles_img = a.tiles_w * a.tiles_h;
  const int nvx = a.HR * a.HW * 16;
  for (int tile = slot; tile < a.ntiles; tile += nslots) {
    const int n = tile / tiles_img, ti = tile - n * tiles_img;
    const int th = ti / a.tiles_w, ho0 = th * a.TR, wo0 = (ti - th * a.tiles_w) * a.TW;
    const int hi0 = ho0 * a.stride - a.pad, wi0 = wo0 * a.stride - a.pad;
    for (int v0 = t; v0 < nvx; v0 += 4 * NT) {
      Vec16<T> val[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int v = v0 + u * NT;
        const int hp = v >> 4, c16 = v & 15;
        const int hr = (int)(((unsigned)hp * a.mHW) >> 20), hw = hp - hr * a.HW;
        const int hi = hi0 + hr, wi = wi0 + hw;
        if (v < nvx && (unsigned)hi < (unsigned)a.Hi && (unsigned)wi < (unsigned)a.Wi && c16 < cvalid)
          val[u] = load16(X + (((long long)n * a.Hi + hi) * a.Wi + wi) * a.C + c0 + c16 * 8);
        else {
#pragma unroll
          for (int e = 0; e < 8; ++e) val[u].v[e] = ElemTraits<T>::from_f32(0.f);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int v = v0 + u * NT;
        if (v < nvx) *reinterpret_cast<Vec16<T>*>(lds + (v >> 4) * PP + (v & 15) * 16) = val[u];
      }
    }
    gconv_lds_barrier();
    if (wave_live) {
      f32x16 acc[2];
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int g = 0; g < 16; ++g) acc[b][g] = 0.f;
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          const int tap = (r * a.HW + s) * PP;
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
              const frag xb = *reinterpret_cast<const frag*>(lds + hb[b] + tap + h * 32);
              acc[b] = GMfma<T>::m32(wf[(r * 3 + s) * 2 + h], xb, acc[b]);
            }
        }
      // C/D map of the 32x32 MFMA: column (pixel) lane & 31, rows (channels) (g & 3) + 8 * (g >> 2) + 4 * (lane >> 5)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int ho = ho0 + ptr[b], wo = wo0 + ptw[b];
        if (ho < a.Ho && wo < a.Wo) {
          T* __restrict__ yo = Y + (((long long)n * a.Ho + ho) * a.Wo + wo) * a.C + chunk * 32 + 4 * lh;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            VecN<T, 4> o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o.v[e] = ElemTraits<T>::from_f32(acc[b][q * 4 + e]);
            *reinterpret_cast<u32x2*>(yo + q * 8) = *reinterpret_cast<const u32x2*>(&o);
          }
        }
      }
    }
    gconv_lds_barrier();   // every wave is done reading this tile's halo
  }
#endif
}

// Chunk filters of a grouped 3x3 layer in one launch: from the logical weight w[K][Cg][9] (fp32 or the compute type)
// the forward layout f[K/32][32 output][9][32 input] (the chunk's groups on the diagonal, zeros elsewhere) and, when
// d is given, the input-gradient layout d[K/32][32 input][9][32 output] with the taps rotated by 180 degrees (`rot`:
// a stride-1 layer, whose input gradient runs on the forward kernel) or in place.
template <typename T, typename WT>
__global__ void __launch_bounds__(256)
gconv_filters_kernel(const WT* __restrict__ w, T* __restrict__ f, T* __restrict__ d, int K, int Cg, int rot) {
  const long long n = (long long)K * 9 * 32;
  const long long i = blockIdx.x * 256LL + threadIdx.x;
  if (i >= n) return;
  const int in = (int)(i & 31), tap = (int)((i >> 5) % 9), k = (int)(i / (9 * 32));
  const int o = k & 31, base = (o / Cg) * Cg;
  const int c = in - base;
  T v = ElemTraits<T>::from_f32(0.f);
  if (c >= 0 && c < Cg) v = ElemTraits<T>::from_f32((float)w[((long long)k * Cg + c) * 9 + tap]);
  f[i] = v;
  if (d != nullptr) {
    const int chunk = k >> 5;
    d[(((long long)chunk * 32 + in) * 9 + (rot ? 8 - tap : tap)) * 32 + o] = v;
  }
}

// dw[K][Cg][9] (fp32 or the compute type) from the chunk-wise gradient dwc[K/32][32][9][32]: the diagonal blocks.
template <typename WT>
__global__ void __launch_bounds__(256)
gconv_diag_kernel(const float* __restrict__ dwc, WT* __restrict__ dw, int K, int Cg) {
  const long long n = (long long)K * Cg * 9;
  const long long i = blockIdx.x * 256LL + threadIdx.x;
  if (i >= n) return;
  const int tap = (int)(i % 9), c = (int)((i / 9) % Cg), k = (int)(i / (9LL * Cg));
  const int o = k & 31, base = (o / Cg) * Cg;
  dw[i] = (WT)dwc[((long long)k * 9 + tap) * 32 + base + c];
}

struct GWArgs {
  const void* dy;  // [N][Ho][Wo][C]
  const void* x;   // [N][Hi][Wi][C]
  float* dw;       // [C/32][32 out][9][32 in] fp32, accumulated into
  int N, Hi, Wi, Ho, Wo, C;
  int stride, pad;
  int TW, TR;      // output columns / rows per tile, TW * TR <= 64
  int HW, HR;      // halo columns / rows: (TW - 1) * stride + 3, (TR - 1) * stride + 3
  int tiles_w, tiles_h;   // tiles per output row / per image column of tiles
  unsigned mTW, mHW;      // ceil(2^20 / d)
  int ntiles;
};

template <typename T>
__device__ __forceinline__ typename GMfma<T>::frag g_tr_read(const unsigned char* p0, const unsigned char* p1) {
  typedef s16x4 __attribute__((address_space(3))) * lp;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(p0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(p1));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(typename GMfma<T>::frag, v);
}

// One workgroup: 4 chunks (128 channels) of the tiles slot, slot + nslots, ...
template <typename T>
__global__ void __launch_bounds__(256, 2)
gconv_wgrad_kernel(GWArgs a) {
  constexpr int BM = 64, NT = 256;
  constexpr int PP = 128 * 2 + 32;   // bytes per pixel row of both images: 32 bytes apart modulo 256
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  typedef typename GMfma<T>::frag frag;
  const int t = threadIdx.x, lane = t & 63;
  const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
  const int nchunks = a.C >> 5;
  const int ncg = (nchunks + 3) >> 2;
  const int cg = blockIdx.x % ncg, slot = blockIdx.x / ncg, nslots = gridDim.x / ncg;
  const int chunk = cg * 4 + wid;
  const bool wave_live = chunk < nchunks;
  const int c0 = cg * 128;                       // first channel of the workgroup
  const int cvalid = min(128, a.C - c0) >> 3;     // 16-byte vectors per pixel that exist
  unsigned char* imgD = lds;
  unsigned char* imgX = lds + BM * PP;
  const T* __restrict__ DY = reinterpret_cast<const T*>(a.dy);
  const T* __restrict__ X = reinterpret_cast<const T*>(a.x);

  // wgrad role (32x32x16): A = dy^T (rows: the chunk's 32 output channels), B = x at one tap (columns: its 32 input
  // channels), K = 16 pixels per step.  A lane reads the pixel rows q0 = kk * 16 + prow0 and q0 + 4.
  const int li = lane & 15, lg = lane >> 4;
  const int prow0 = 8 * (lg >> 1) + (li >> 2);
  const int csub = 16 * (lg & 1) + 4 * (li & 3);
  int hb[8];   // halo offset of output pixel q (tap 0, 0), bytes
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int q = (j >> 1) * 16 + prow0 + (j & 1) * 4;
    const int tr = (int)(((unsigned)q * a.mTW) >> 20), tw = q - tr * a.TW;
    hb[j] = tr < a.TR ? ((tr * a.stride) * a.HW + tw * a.stride) * PP + (wid * 32 + csub) * 2 : (wid * 32 + csub) * 2;
  }
  f32x16 wacc[9];
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int g = 0; g < 16; ++g) wacc[k][g] = 0.f;

  const int tiles_img = a.tiles_w * a.tiles_h;
  static_assert(BM * 16 == 4 * NT, "a thread stages four dy vectors per tile");
  const int nvx = a.HR * a.HW * 16;
  for (int tile = slot; tile < a.ntiles; tile += nslots) {
    const int n = tile / tiles_img, ti = tile - n * tiles_img;
    const int th = ti / a.tiles_w, ho0 = th * a.TR, wo0 = (ti - th * a.tiles_w) * a.TW;
    // dy image: [q][128 channels], zero for pixels outside the tensor (four requests in flight per thread)
    {
      Vec16<T> val[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int v = t + u * NT;
        const int q = v >> 4, c16 = v & 15;
        const int tr = (int)(((unsigned)q * a.mTW) >> 20), tw = q - tr * a.TW;
        const int ho = ho0 + tr, wo = wo0 + tw;
        if (tr < a.TR && ho < a.Ho && wo < a.Wo && c16 < cvalid)
          val[u] = load16(DY + (((long long)n * a.Ho + ho) * a.Wo + wo) * a.C + c0 + c16 * 8);
        else {
#pragma unroll
          for (int e = 0; e < 8; ++e) val[u].v[e] = ElemTraits<T>::from_f32(0.f);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int v = t + u * NT;
        *reinterpret_cast<Vec16<T>*>(imgD + (v >> 4) * PP + (v & 15) * 16) = val[u];
      }
    }
    // x halo: [hr][hw][128 channels], zero outside the tensor (the padding)
    const int hi0 = ho0 * a.stride - a.pad, wi0 = wo0 * a.stride - a.pad;
    for (int v0 = t; v0 < nvx; v0 += 4 * NT) {
      Vec16<T> val[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int v = v0 + u * NT;
        const int hp = v >> 4, c16 = v & 15;
        const int hr = (int)(((unsigned)hp * a.mHW) >> 20), hw = hp - hr * a.HW;
        const int hi = hi0 + hr, wi = wi0 + hw;
        if (v < nvx && (unsigned)hi < (unsigned)a.Hi && (unsigned)wi < (unsigned)a.Wi && c16 < cvalid)
          val[u] = load16(X + (((long long)n * a.Hi + hi) * a.Wi + wi) * a.C + c0 + c16 * 8);
        else {
#pragma unroll
          for (int e = 0; e < 8; ++e) val[u].v[e] = ElemTraits<T>::from_f32(0.f);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int v = v0 + u * NT;
        if (v < nvx) *reinterpret_cast<Vec16<T>*>(imgX + (v >> 4) * PP + (v & 15) * 16) = val[u];
      }
    }
    __syncthreads();
    if (wave_live) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int q0 = kk * 16 + prow0;
        const frag fa = g_tr_read<T>(imgD + q0 * PP + (wid * 32 + csub) * 2, imgD + (q0 + 4) * PP + (wid * 32 + csub) * 2);
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int s = 0; s < 3; ++s) {
            const int tap = (r * a.HW + s) * PP;
            const frag fb = g_tr_read<T>(imgX + hb[kk * 2] + tap, imgX + hb[kk * 2 + 1] + tap);
            wacc[r * 3 + s] = GMfma<T>::m32(fa, fb, wacc[r * 3 + s]);
          }
      }
    }
    __syncthreads();   // every wave is done reading this tile's images
  }

  // C/D map of the 32x32 MFMA: column (input channel) lane & 31, rows (output channels) (g & 3) + 8 * (g >> 2) + 4 * (lane >> 5)
  if (wave_live) {
    const int l31 = lane & 31, lh = lane >> 5;
    float* __restrict__ dwc = a.dw + (long long)chunk * 32 * 9 * 32;
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const int co = (g & 3) + 8 * (g >> 2) + 4 * lh;
        atomicAdd(dwc + (co * 9 + k) * 32 + l31, wacc[k][g]);
      }
  }
#endif
}

int log2_exact(int v) {
  for (int i = 0; i < 16; ++i)
    if ((1 << i) == v) return i;
  return -1;
}

int gconv_workgroups(int ncg, long long tiles, int per_cu) {
  static const int env = [] { const char* e = getenv("PDA_GCONV_WGS"); return e ? atoi(e) : 0; }();
  const int total = env > 0 ? env : 256 * per_cu;
  long long slots = total / ncg;
  if (slots < 1) slots = 1;
  if (slots > tiles) slots = tiles;
  return (int)slots * ncg;
}

}  // namespace

// y[N][Ho][Wo][C] = 3x3 convolution of x[N][Hi][Wi][C] per 32-channel chunk with the chunk filters
// w[C/32][32][9][32] ([result channel][tap][reduced channel], the groups on the diagonal).
// mode 0: fprop (x is the input; Ho = (Hi + 2 pad - 3) / stride + 1).
// mode 1: dgrad (x is dy [N][Hi][Wi][C], y is dx [N][Ho][Wo][C]; stride / pad are the forward convolution's;
//         w is [C/32][32 input channels][9][32 output channels]).
// <0: not covered (-1: geometry, -4: tensor too large for 32-bit offsets, -5: alignment, -6: dtype).
PDA_EXPORT int pda_grouped_conv3x3(int dtype, int mode, const void* x, const void* w, void* y, int N, int Hi, int Wi,
                                   int Ho, int Wo, int C, int stride, int pad, hipStream_t stream) {
  if (dtype != PDA_F16 && dtype != PDA_BF16) return -6;
  if (mode != 0 && mode != 1) return -1;
  const int sl = log2_exact(stride);
  if (C <= 0 || C % 32 != 0 || sl < 0 || pad < 0 || pad > 2) return -1;
  if (N <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0 || Wo >= 512 || Ho >= 512) return -1;
  if (!x || !w || !y) return -11;
  if ((((uintptr_t)x | (uintptr_t)w | (uintptr_t)y) & 15) != 0) return -5;
  const long long xb = (long long)N * Hi * Wi * C * 2, M = (long long)N * Ho * Wo;
  if (xb >= 0x7fffffffLL || M * 1 >= 0x7fffffffLL - 64) return -4;
  static const bool gather_only = [] { const char* e = getenv("PDA_GCONV_GATHER"); return e && atoi(e) == 1; }();
  if (mode == 0 && !gather_only) {
    GLArgs g;
    g.x = x; g.w = w; g.y = y; g.N = N; g.Hi = Hi; g.Wi = Wi; g.Ho = Ho; g.Wo = Wo; g.C = C; g.stride = stride; g.pad = pad;
    // up to 64 output pixels per tile, as square as the row length allows (a smaller halo per pixel)
    const int parts = (Wo + 31) / 32;
    g.TW = (Wo + parts - 1) / parts;
    g.TR = 64 / g.TW;
    if (g.TR > Ho) g.TR = Ho;
    g.HW = (g.TW - 1) * stride + 3; g.HR = (g.TR - 1) * stride + 3;
    g.tiles_w = (Wo + g.TW - 1) / g.TW; g.tiles_h = (Ho + g.TR - 1) / g.TR;
    g.mTW = ((1u << 20) + g.TW - 1) / g.TW; g.mHW = ((1u << 20) + g.HW - 1) / g.HW;
    const long long tiles = (long long)N * g.tiles_w * g.tiles_h;
    const int lds_bytes = g.HR * g.HW * (128 * 2 + 16);
    if (tiles < 0x7fffffffLL && lds_bytes <= 160 * 1024 && stride <= 2) {
      g.ntiles = (int)tiles;
      static const hipError_t attr = [] {
        hipError_t r = hipFuncSetAttribute(reinterpret_cast<const void*>(gconv_lds_kernel<f16>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (r != hipSuccess) return r;
        return hipFuncSetAttribute(reinterpret_cast<const void*>(gconv_lds_kernel<bf16>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      }();
      if (attr != hipSuccess) return (int)attr;
      const int ncg = (C / 32 + 3) / 4;
      int per_cu = 160 * 1024 / lds_bytes;
      if (per_cu > 3) per_cu = 3;
      const int nwg = gconv_workgroups(ncg, tiles, per_cu);
      if (dtype == PDA_F16) hipLaunchKernelGGL((gconv_lds_kernel<f16>), dim3(nwg), dim3(256), lds_bytes, stream, g);
      else hipLaunchKernelGGL((gconv_lds_kernel<bf16>), dim3(nwg), dim3(256), lds_bytes, stream, g);
      hipError_t e = hipGetLastError();
      return e == hipSuccess ? 0 : (int)e;
    }
  }
  GArgs a;
  a.x = x; a.w = w; a.y = y; a.x_bytes = xb; a.N = N; a.Hi = Hi; a.Wi = Wi; a.Ho = Ho; a.Wo = Wo; a.C = C;
  a.stride_log2 = sl; a.pad = pad; a.M = (int)M;
  a.mWo = (1ULL << 40) / (unsigned)Wo + 1; a.mHo = (1ULL << 40) / (unsigned)Ho + 1;
  a.ntiles = (int)((M + 31) / 32);
  const int ncg = (C / 32 + 3) / 4;
  const int nwg = gconv_workgroups(ncg, a.ntiles, 3);
  if (dtype == PDA_F16) {
    if (mode == 0) hipLaunchKernelGGL((gconv_kernel<f16, false>), dim3(nwg), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((gconv_kernel<f16, true>), dim3(nwg), dim3(256), 0, stream, a);
  } else {
    if (mode == 0) hipLaunchKernelGGL((gconv_kernel<bf16, false>), dim3(nwg), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((gconv_kernel<bf16, true>), dim3(nwg), dim3(256), 0, stream, a);
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

// dw[C/32][32 output channels][9][32 input channels] (fp32, accumulated into with float atomics) += the chunk-wise
// weight gradient of the 3x3 convolution y = conv(x) with dy[N][Ho][Wo][C], x[N][Hi][Wi][C].
PDA_EXPORT int pda_grouped_conv3x3_wgrad(int dtype, const void* dy, const void* x, float* dw, int N, int Hi, int Wi,
                                         int Ho, int Wo, int C, int stride, int pad, hipStream_t stream) {
  if (dtype != PDA_F16 && dtype != PDA_BF16) return -6;
  if (C <= 0 || C % 32 != 0 || (stride != 1 && stride != 2) || pad < 0 || pad > 2) return -1;
  if (N <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0) return -1;
  if (!dy || !x || !dw) return -11;
  if ((((uintptr_t)dy | (uintptr_t)x) & 15) != 0) return -5;
  GWArgs a;
  a.dy = dy; a.x = x; a.dw = dw; a.N = N; a.Hi = Hi; a.Wi = Wi; a.Ho = Ho; a.Wo = Wo; a.C = C;
  a.stride = stride; a.pad = pad;
  const int parts = (Wo + 31) / 32;   // up to 64 output pixels per tile, as square as the row length allows
  a.TW = (Wo + parts - 1) / parts;
  a.TR = 64 / a.TW;
  if (a.TR > Ho) a.TR = Ho;
  a.HW = (a.TW - 1) * stride + 3; a.HR = (a.TR - 1) * stride + 3;
  a.tiles_w = (Wo + a.TW - 1) / a.TW; a.tiles_h = (Ho + a.TR - 1) / a.TR;
  a.mTW = ((1u << 20) + a.TW - 1) / a.TW; a.mHW = ((1u << 20) + a.HW - 1) / a.HW;
  const long long tiles = (long long)N * a.tiles_w * a.tiles_h;
  if (tiles >= 0x7fffffffLL) return -4;
  a.ntiles = (int)tiles;
  const int lds_bytes = (64 + a.HR * a.HW) * (128 * 2 + 32);
  if (lds_bytes > 160 * 1024) return -1;
  const int ncg = (C / 32 + 3) / 4;
  const int per_cu = lds_bytes <= 80 * 1024 ? 2 : 1;
  const int nwg = gconv_workgroups(ncg, tiles, per_cu);
  // (the dynamic LDS limit is raised once, to the hardware's 160 KB)
  static const hipError_t attr = [] {
    hipError_t r = hipFuncSetAttribute(reinterpret_cast<const void*>(gconv_wgrad_kernel<f16>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (r != hipSuccess) return r;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(gconv_wgrad_kernel<bf16>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }();
  if (attr != hipSuccess) return (int)attr;
  if (dtype == PDA_F16) hipLaunchKernelGGL((gconv_wgrad_kernel<f16>), dim3(nwg), dim3(256), lds_bytes, stream, a);
  else hipLaunchKernelGGL((gconv_wgrad_kernel<bf16>), dim3(nwg), dim3(256), lds_bytes, stream, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

// See gconv_filters_kernel.  wdtype: type of w (PDA_F32 or == dtype).  d may be null.
PDA_EXPORT int pda_grouped_chunk_filters(int dtype, int wdtype, const void* w, void* f, void* d, int K, int Cg, int rot,
                                         hipStream_t stream) {
  if (dtype != PDA_F16 && dtype != PDA_BF16) return -6;
  if (wdtype != PDA_F32 && wdtype != dtype) return -6;
  if (K <= 0 || K % 32 != 0 || Cg <= 0 || 32 % Cg != 0) return -1;
  if (!w || !f) return -11;
  const long long n = (long long)K * 9 * 32;
  const dim3 grid((unsigned)((n + 255) / 256));
  if (dtype == PDA_F16) {
    if (wdtype == PDA_F32) hipLaunchKernelGGL((gconv_filters_kernel<f16, float>), grid, dim3(256), 0, stream, (const float*)w, (f16*)f, (f16*)d, K, Cg, rot);
    else hipLaunchKernelGGL((gconv_filters_kernel<f16, f16>), grid, dim3(256), 0, stream, (const f16*)w, (f16*)f, (f16*)d, K, Cg, rot);
  } else {
    if (wdtype == PDA_F32) hipLaunchKernelGGL((gconv_filters_kernel<bf16, float>), grid, dim3(256), 0, stream, (const float*)w, (bf16*)f, (bf16*)d, K, Cg, rot);
    else hipLaunchKernelGGL((gconv_filters_kernel<bf16, bf16>), grid, dim3(256), 0, stream, (const bf16*)w, (bf16*)f, (bf16*)d, K, Cg, rot);
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

// See gconv_diag_kernel.  wdtype: type of dw (PDA_F32, PDA_F16 or PDA_BF16).
PDA_EXPORT int pda_grouped_chunk_diag(int wdtype, const float* dwc, void* dw, int K, int Cg, hipStream_t stream) {
  if (K <= 0 || K % 32 != 0 || Cg <= 0 || 32 % Cg != 0) return -1;
  if (!dwc || !dw) return -11;
  const long long n = (long long)K * Cg * 9;
  const dim3 grid((unsigned)((n + 255) / 256));
  if (wdtype == PDA_F32) hipLaunchKernelGGL((gconv_diag_kernel<float>), grid, dim3(256), 0, stream, dwc, (float*)dw, K, Cg);
  else if (wdtype == PDA_F16) hipLaunchKernelGGL((gconv_diag_kernel<f16>), grid, dim3(256), 0, stream, dwc, (f16*)dw, K, Cg);
  else if (wdtype == PDA_BF16) hipLaunchKernelGGL((gconv_diag_kernel<bf16>), grid, dim3(256), 0, stream, dwc, (bf16*)dw, K, Cg);
  else return -6;
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
