// 3x3 convolution with 64 input and 64 output channels (the middle conv of ResNet-50's first stage, 56x56 maps) in
// fp16 / bf16 on gfx950, weight-stationary.
//
// With N = 64 and K = 576 the tiled implicit GEMM spends its time around the GEMM: a 128x64 tile runs nine K steps
// between a prologue and a write-out, and every workgroup stages the same 73 KB of weights through LDS again.  Here
// the weights never move: a wave keeps the [32 output channels][576] slice it needs in 144 registers, in MFMA operand
// layout, for the whole (persistent) kernel.  The activations are staged once per tile of up to 64 output pixels
// as the tile's halo, [rows][columns][64 channels] with a 144-byte pixel pitch (conflict-free ds_read_b128), and a
// lane's operand for tap (r, s) and K step k is one 16-byte read at its pixel's halo position shifted by the tap:
// 36 reads and 36 v_mfma_f32_32x32x16 per wave and tile.  The operands are swapped (D = W * X^T) so that a lane
// holds 4 consecutive channels of one pixel per register quad: 8-byte stores.  The next tile's halo is requested
// (buffer loads; an out-of-range offset is the zero padding) before the MFMAs of the current one.
// Forward with the BatchNorm statistics (one partial row per workgroup; accumulated in lane-owned LDS vectors).
// Measured on 56x56 at batch 400: 133 us against 209 us for the 128x64 tile (io floor 64 us).  The phases do not
// overlap further: a tile's operand reads (36 KB per wave) keep the LDS pipe as busy as its MFMAs keep the matrix
// pipe, and 144 weight registers per lane leave room for two waves per SIMD only (PDA_C64_ABLATE times the parts).
// (Reference role: torchvision's Bottleneck.conv2 of layer1, /root/reference/2.distributed.py:113-114,224-231.)
#include "common.h"

namespace {

struct C64Args {
  const void* x;   // [N][Hi][Wi][64]
  const void* w;   // [64 result channels][9 taps][64 reduced channels]
  void* y;         // [N][Ho][Wo][64]
  float* psum;     // [workgroups][64] per-workgroup column sums of the (rounded) output, or null
  float* psq;      // ... and of its squares (MASKED kernels: of output * bnc)
  const void* bnc;        // MASKED kernels (a dgrad with the producer's BatchNorm + ReLU backward folded in): [N][Ho][Wo][64],
  const float* bn_scale;  // the producing layer's conv output; the result is zeroed where bnc * bn_scale + bn_shift <= 0
  const float* bn_shift;
  int flip;        // the filter is used rotated by 180 degrees (tap 8 - t): the input gradient of a stride-1 layer
  long long x_bytes;
  int N, Hi, Wi, Ho, Wo;
  int stride, pad;
  int TW, TR, HW, HR, tiles_w, tiles_h;
  unsigned mTW, mHW;   // ceil(2^20 / d)
  int ntiles;
  int ablate;   // diagnostics (PDA_C64_ABLATE), bits: 1 = no output stores, 2 = no MFMA loop, 4 = no halo requests
};

template <typename T> struct C64Mfma;
template <> struct C64Mfma<f16> {
  typedef f16x8 frag;
  static __device__ __forceinline__ f32x16 m32(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};
template <> struct C64Mfma<bf16> {
  typedef bf16x8 frag;
  static __device__ __forceinline__ f32x16 m32(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};

// s_waitcnt immediate (gfx9 layout): vmcnt in bits 3:0 and 15:14 (63: do not wait), expcnt 6:4 (7: do not wait),
// lgkmcnt 11:8
constexpr int c64_lgkm_wait(int lgkm) { return 15 | (3 << 14) | (7 << 4) | ((lgkm & 15) << 8); }

// ds_read_b128 as a bare instruction: the kernel orders these reads itself with counted lgkmcnt waits (LDS operations
// return in order).  Left to the compiler, each pair of operand reads is issued right in front of its pair of MFMAs
// (whatever the source order or the scheduling-group hints say) and every pair pays the LDS latency.
template <int OFF>
__device__ __forceinline__ u32x4 c64_read128(unsigned addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
__device__ __forceinline__ u32x4 c64_read128_sub(unsigned addr, int sub) {
  switch (sub) {
    case 0: return c64_read128<0>(addr);
    case 1: return c64_read128<32>(addr);
    case 2: return c64_read128<64>(addr);
    default: return c64_read128<96>(addr);
  }
}
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also fences global memory: every wave would wait
// for its output stores of the tile (a trip to memory) before the next tile's halo may be written.
__device__ __forceinline__ void c64_lds_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_waitcnt(c64_lgkm_wait(0));
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void c64_wait_reads(int pending) {
  switch (pending) {
    case 0: __builtin_amdgcn_s_waitcnt(c64_lgkm_wait(0)); break;
    case 1: __builtin_amdgcn_s_waitcnt(c64_lgkm_wait(1)); break;
    case 2: __builtin_amdgcn_s_waitcnt(c64_lgkm_wait(2)); break;
    default: __builtin_amdgcn_s_waitcnt(c64_lgkm_wait(3)); break;
  }
}

// STATS: column sums of the output (and of its squares, or with MASKED of output * bnc) per workgroup.
// MASKED: see C64Args::bnc (implies STATS).
template <typename T, bool STATS, bool MASKED = false>
__global__ void __launch_bounds__(256, 2)
conv3x3_c64_kernel(C64Args a) {
  constexpr int NT = 256, CH = 64;
  constexpr int PPX = CH * 2 + 16;   // halo pixel pitch, bytes
  constexpr int NVH = 4;             // halo vectors per thread (host: HR * HW * 8 <= NVH * NT)
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  typedef typename C64Mfma<T>::frag frag;
  const int t = threadIdx.x, lane = t & 63, l31 = lane & 31, lh = lane >> 5;
  const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
  const int oc = wid & 1, pb = wid >> 1;   // wave -> output channels oc * 32 .. + 31 of pixels pb * 32 .. + 31
  T* __restrict__ Y = reinterpret_cast<T*>(a.y);

  frag wf[36];
  {
    const T* __restrict__ W = reinterpret_cast<const T*>(a.w) + (long long)(oc * 32 + l31) * 9 * CH + lh * 8;
#pragma unroll
    for (int k = 0; k < 36; ++k) {
      const Vec16<T> v = load16(W + (a.flip ? 8 - (k >> 2) : (k >> 2)) * CH + (k & 3) * 16);
      wf[k] = __builtin_bit_cast(frag, v);
    }
  }
  // the lane's output pixel inside a tile
  const int q = pb * 32 + l31;
  int ptr = (int)(((unsigned)q * a.mTW) >> 20);
  const int ptw = q - ptr * a.TW;
  const bool inside = ptr < a.TR;
  const int hb = (inside ? ((ptr * a.stride) * a.HW + ptw * a.stride) * PPX : 0) + lh * 16;
  if (!inside) ptr = 1 << 20;   // (never a pixel of the tensor)
  typedef __attribute__((address_space(3))) void* lds_ptr;
  unsigned tapaddr[9];          // LDS address of the lane's operand per tap
#pragma unroll
  for (int tp = 0; tp < 9; ++tp)
    tapaddr[tp] = (unsigned)(uintptr_t)(lds_ptr)lds + hb + ((tp / 3) * a.HW + tp % 3) * PPX;

  // halo staging role: vector v = t + NT * u -> halo pixel v / 8 (row hr, column hw), 16-byte chunk v % 8
  const int nvx = a.HR * a.HW * 8;
  int shr[NVH], shw[NVH];
#pragma unroll
  for (int u = 0; u < NVH; ++u) {
    const int v = t + NT * u, hp = v >> 3;
    shr[u] = (int)(((unsigned)hp * a.mHW) >> 20);
    shw[u] = hp - shr[u] * a.HW;
    if (v >= nvx) shr[u] = 1 << 20;   // (never a row of the tensor)
  }
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, (int)a.x_bytes, 0x00020000);
  const int tiles_img = a.tiles_w * a.tiles_h;
  u32x4 pv[NVH];
  auto fetch = [&](int tile) {
    const int n = tile / tiles_img, ti = tile - n * tiles_img;
    const int th = ti / a.tiles_w;
    const int hi0 = th * a.TR * a.stride - a.pad, wi0 = (ti - th * a.tiles_w) * a.TW * a.stride - a.pad;
#pragma unroll
    for (int u = 0; u < NVH; ++u) {
      const int hi = hi0 + shr[u], wi = wi0 + shw[u];
      const bool ok = (unsigned)hi < (unsigned)a.Hi && (unsigned)wi < (unsigned)a.Wi;
      // (an offset past the tensor returns zeros: the padding, and the vectors past the halo)
      const int off = ok ? (int)((((unsigned)n * a.Hi + hi) * a.Wi + wi) * (unsigned)(CH * 2)) + ((t + NT * u) & 7) * 16 : -1;
      pv[u] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
    }
  };

  // The statistics are accumulated in LDS, one slot per lane and value (no two lanes share one): 32 more registers
  // per lane would leave the compiler none to keep operand reads in flight under the MFMAs.
  constexpr int RP = 36;   // floats per lane: 32 values + padding (144 bytes: conflict-free 16-byte accesses)
  float* red = reinterpret_cast<float*>(lds + a.HR * a.HW * PPX);   // [4 waves][64 lanes][RP]
  float* mine = red + (wid * 64 + lane) * RP;
  float* tab = red + 4 * 64 * RP;   // MASKED: [2][64] scale, shift
  if (MASKED) {
    if (t < 128) tab[t] = t < 64 ? a.bn_scale[t] : a.bn_shift[t - 64];   // (read after the first barrier of the loop)
  }
  const T* __restrict__ Bc = reinterpret_cast<const T*>(a.bnc);
  if (STATS) {
#pragma unroll
    for (int g = 0; g < 32; g += 4) *reinterpret_cast<f32x4*>(mine + g) = f32x4{0.f, 0.f, 0.f, 0.f};   // (own slots: no barrier)
  }

  const int slot = blockIdx.x, nslots = gridDim.x;
  // (a workgroup past the last tile still takes part in nothing: the grid never exceeds the tile count)
  fetch(min(slot, a.ntiles - 1));
  for (int tile = slot; tile < a.ntiles; tile += nslots) {
#pragma unroll
    for (int u = 0; u < NVH; ++u) {
      const int v = t + NT * u;
      if (v < nvx) *reinterpret_cast<u32x4*>(lds + (v >> 3) * PPX + (v & 7) * 16) = pv[u];
    }
    c64_lds_barrier();
    // this tile's pixel (the lane's column of the result)
    const int n = tile / tiles_img, ti = tile - n * tiles_img;
    const int th = ti / a.tiles_w;
    const int ho = th * a.TR + ptr, wo = (ti - th * a.tiles_w) * a.TW + ptw;
    const bool pix_ok = ho < a.Ho && wo < a.Wo;
    const long long pix = pix_ok ? (((long long)n * a.Ho + ho) * a.Wo + wo) * CH + oc * 32 + 4 * lh : 0;
    u32x2 bcv[MASKED ? 4 : 1];
    if (MASKED) {
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) bcv[qd] = *reinterpret_cast<const u32x2*>(Bc + pix + qd * 8);
    }
    // the next tile's halo travels while this one computes (past the end: the last tile once more, unused)
    if (!(a.ablate & 4)) fetch(min(tile + nslots, a.ntiles - 1));
    __builtin_amdgcn_sched_barrier(0);
    // two accumulator chains (an MFMA that adds to the result of the one before it waits for all of its passes); the
    // operand reads run four K steps ahead of their MFMAs in a ring of registers
    f32x16 acc, acc1;
#pragma unroll
    for (int g = 0; g < 16; ++g) { acc[g] = 0.f; acc1[g] = 0.f; }
    constexpr int AHEAD = 4;
    u32x4 xb[AHEAD];
#pragma unroll
    for (int k = 0; k < AHEAD; ++k) xb[k] = c64_read128_sub(tapaddr[k >> 2], k & 3);
#pragma unroll
    for (int k = 0; k < 36; ++k) {
      if (a.ablate & 2) break;
      c64_wait_reads(35 - k < AHEAD - 1 ? 35 - k : AHEAD - 1);
      __builtin_amdgcn_sched_barrier(0);   // nothing that reads the operand registers may move above the wait
      const frag f = __builtin_bit_cast(frag, xb[k % AHEAD]);
      if (k & 1) acc1 = C64Mfma<T>::m32(wf[k], f, acc1);
      else acc = C64Mfma<T>::m32(wf[k], f, acc);
      if (k + AHEAD < 36) xb[k % AHEAD] = c64_read128_sub(tapaddr[(k + AHEAD) >> 2], (k + AHEAD) & 3);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < 16; ++g) acc[g] += acc1[g];
    // C/D map of the 32x32 MFMA: column (pixel) lane & 31, rows (channels) (g & 3) + 8 * (g >> 2) + 4 * (lane >> 5)
    if (pix_ok && !(a.ablate & 1)) {
      T* __restrict__ yo = Y + pix;
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        VecN<T, 4> o;
        f32x4 sv = {0.f, 0.f, 0.f, 0.f}, qv = {0.f, 0.f, 0.f, 0.f};
        f32x4 msc = {0.f, 0.f, 0.f, 0.f}, msf = {0.f, 0.f, 0.f, 0.f};
        VecN<T, 4> cf4;
        if (MASKED) {
          msc = *reinterpret_cast<const f32x4*>(tab + oc * 32 + 4 * lh + qd * 8);
          msf = *reinterpret_cast<const f32x4*>(tab + 64 + oc * 32 + 4 * lh + qd * 8);
          *reinterpret_cast<u32x2*>(&cf4) = bcv[qd];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float r = acc[qd * 4 + e];
          float cf = 0.f;
          if (MASKED) {
            // ReLU backward of the layer that produced this conv's forward input, mask recomputed from its conv output
            cf = ElemTraits<T>::to_f32(cf4.v[e]);
            if (!(cf * msc[e] + msf[e] > 0.f)) r = 0.f;
          }
          o.v[e] = ElemTraits<T>::from_f32(r);
          if (STATS) {
            const float fv = ElemTraits<T>::to_f32(o.v[e]);
            sv[e] = fv;
            qv[e] = MASKED ? fv * cf : fv * fv;
          }
        }
        *reinterpret_cast<u32x2*>(yo + qd * 8) = *reinterpret_cast<const u32x2*>(&o);
        if (STATS) {
          *reinterpret_cast<f32x4*>(mine + qd * 4) += sv;
          *reinterpret_cast<f32x4*>(mine + 16 + qd * 4) += qv;
        }
      }
    }
    c64_lds_barrier();   // every wave is done reading this tile's halo
  }
  __builtin_amdgcn_s_waitcnt(0);   // (the unused last request)

  if (STATS) {
    // one row per workgroup: fold the 32 pixel lanes and the two pixel waves of each channel
    __syncthreads();
    if (t < 128) {
      const int which = t >> 6, ch = t & 63;
      const int o2 = ch >> 5, c = ch & 31;
      const int clh = (c >> 2) & 1, g = (c & 3) + 4 * (c >> 3);
      float accs = 0.f;
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll 8
        for (int l = 0; l < 32; ++l) accs += red[((p * 2 + o2) * 64 + clh * 32 + l) * RP + which * 16 + g];
      float* dst = which == 0 ? a.psum : a.psq;
      dst[(long long)blockIdx.x * CH + ch] = accs;
    }
  }
#endif
}

bool c64_geometry(C64Args& a, int Ho, int Wo, int stride) {
  const int parts = (Wo + 31) / 32;   // up to 64 output pixels per tile, as square as the row length allows
  a.TW = (Wo + parts - 1) / parts;
  a.TR = 64 / a.TW;
  if (a.TR > Ho) a.TR = Ho;
  for (;;) {
    a.HW = (a.TW - 1) * stride + 3;
    a.HR = (a.TR - 1) * stride + 3;
    if (a.HR * a.HW * 8 <= 4 * 256) break;
    if (a.TR > 1) --a.TR;
    else return false;
  }
  a.tiles_w = (Wo + a.TW - 1) / a.TW;
  a.tiles_h = (Ho + a.TR - 1) / a.TR;
  a.mTW = ((1u << 20) + a.TW - 1) / a.TW;
  a.mHW = ((1u << 20) + a.HW - 1) / a.HW;
  return true;
}

int c64_workgroups(long long tiles) {
  static const int env = [] { const char* e = getenv("PDA_C64_WGS"); return e ? atoi(e) : 0; }();
  const long long want = env > 0 ? env : 512;   // two per compute unit (230 registers per lane)
  return (int)(tiles < want ? tiles : want);
}

}  // namespace

// Workgroups (= rows of psum / psq) of pda_conv3x3_c64 for this geometry; <= 0: not covered.
PDA_EXPORT int pda_conv3x3_c64_rows(int N, int Ho, int Wo, int stride) {
  C64Args a;
  if (N <= 0 || Ho <= 0 || Wo <= 0 || (stride != 1 && stride != 2) || !c64_geometry(a, Ho, Wo, stride)) return -1;
  const long long tiles = (long long)N * a.tiles_w * a.tiles_h;
  if (tiles >= 0x7fffffffLL) return -4;
  return c64_workgroups(tiles);
}

static int c64_launch(int dtype, C64Args& a, int N, int Hi, int Wi, int Ho, int Wo, int stride, int pad, bool masked,
                      hipStream_t stream) {
  if (dtype != PDA_F16 && dtype != PDA_BF16) return -6;
  if (N <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0 || (stride != 1 && stride != 2) || pad < 0 || pad > 2) return -1;
  if ((Hi + 2 * pad - 3) / stride + 1 != Ho || (Wi + 2 * pad - 3) / stride + 1 != Wo) return -1;
  if (!a.x || !a.w || !a.y || ((a.psum == nullptr) != (a.psq == nullptr))) return -11;
  if ((((uintptr_t)a.x | (uintptr_t)a.w | (uintptr_t)a.y | (uintptr_t)a.bnc) & 15) != 0) return -5;
  a.N = N; a.Hi = Hi; a.Wi = Wi; a.Ho = Ho; a.Wo = Wo; a.stride = stride; a.pad = pad;
  a.x_bytes = (long long)N * Hi * Wi * 64 * 2;
  if (a.x_bytes >= 0x7fffffffLL) return -4;
  if (!c64_geometry(a, Ho, Wo, stride)) return -1;
  const long long tiles = (long long)N * a.tiles_w * a.tiles_h;
  if (tiles >= 0x7fffffffLL) return -4;
  a.ntiles = (int)tiles;
  static const int ablate = [] { const char* e = getenv("PDA_C64_ABLATE"); return e ? atoi(e) : 0; }();
  a.ablate = ablate;
  const int nwg = c64_workgroups(tiles);
  int lds_bytes = a.HR * a.HW * (64 * 2 + 16);
  if (a.psum != nullptr) lds_bytes += 4 * 64 * 36 * 4 + 2 * 64 * 4;   // the statistics slots and the mask table behind the halo
#define PDA_C64_LAUNCH(T, ST, MK) hipLaunchKernelGGL((conv3x3_c64_kernel<T, ST, MK>), dim3(nwg), dim3(256), lds_bytes, stream, a)
  if (dtype == PDA_F16) {
    if (masked) PDA_C64_LAUNCH(f16, true, true);
    else if (a.psum) PDA_C64_LAUNCH(f16, true, false);
    else PDA_C64_LAUNCH(f16, false, false);
  } else {
    if (masked) PDA_C64_LAUNCH(bf16, true, true);
    else if (a.psum) PDA_C64_LAUNCH(bf16, true, false);
    else PDA_C64_LAUNCH(bf16, false, false);
  }
#undef PDA_C64_LAUNCH
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

// y[N][Ho][Wo][64] = 3x3 convolution (stride 1 or 2, padding 0..2) of x[N][Hi][Wi][64] with w[64][3][3][64] (output
// channel, taps, input channel); psum / psq (both or neither): fp32 [pda_conv3x3_c64_rows()][64] per-workgroup sums of
// the rounded output and of its squares.  <0: not covered.
PDA_EXPORT int pda_conv3x3_c64(int dtype, const void* x, const void* w, void* y, float* psum, float* psq, int N, int Hi,
                               int Wi, int Ho, int Wo, int stride, int pad, hipStream_t stream) {
  C64Args a;
  a.x = x; a.w = w; a.y = y; a.psum = psum; a.psq = psq; a.bnc = nullptr; a.bn_scale = nullptr; a.bn_shift = nullptr;
  a.flip = 0;
  return c64_launch(dtype, a, N, Hi, Wi, Ho, Wo, stride, pad, false, stream);
}

// Input gradient of a stride-1 3x3 convolution on 64 -> 64 channels with the backward of the producing layer's
// BatchNorm + ReLU folded in: dx[N][Hi][Wi][64] = (dy[N][Ho][Wo][64] conv^T w) * (bnc * bn_scale + bn_shift > 0), with
// wt[64 input channels][3][3][64 output channels] (the CRSK transpose), bnc[N][Hi][Wi][64] the producing layer's conv
// output; psum / psq: fp32 [pda_conv3x3_c64_rows(N, Hi, Wi, 1)][64] per-workgroup sums of dx and of dx * bnc.
PDA_EXPORT int pda_conv3x3_c64_dgrad_masked(int dtype, const void* dy, const void* wt, void* dx, const void* bnc,
                                            const float* bn_scale, const float* bn_shift, float* psum, float* psq, int N,
                                            int Ho, int Wo, int Hi, int Wi, int pad, hipStream_t stream) {
  if (!bnc || !bn_scale || !bn_shift || !psum || !psq) return -11;
  C64Args a;
  a.x = dy; a.w = wt; a.y = dx; a.psum = psum; a.psq = psq; a.bnc = bnc; a.bn_scale = bn_scale; a.bn_shift = bn_shift;
  a.flip = 1;
  // the forward-type kernel on dy: filter rotated by 180 degrees, padding 2 - pad
  return c64_launch(dtype, a, N, Ho, Wo, Hi, Wi, 1, 2 - pad, true, stream);
}
