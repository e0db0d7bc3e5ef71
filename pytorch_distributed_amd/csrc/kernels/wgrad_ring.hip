// Weight gradient of a convolution, dW[co][r,s,ci] = sum over output pixels of dY[m][co] * X[m @ (r,s)][ci], for
// fp16 / bf16 on gfx950 with both operands streamed into LDS by LDS-DMA (buffer_load ... lds) through a ring of
// buffers.  Same GEMM and the same fragment path as wgrad_kernel in conv_igemm.hip (the reduction runs over
// pixels, the strided dimension of both operands, so fragments come from ds_read_b64_tr_b16 out of [pixel][channel]
// images whose 64-byte segments are XOR-swizzled by the pixel row); what differs is everything around the MFMAs:
//   * no operand passes through registers: a wave instruction moves 1 KiB straight into the ring, so the loop
//     holds no ds_write (13 cycles each on a path two workgroups of a CU share) and no staging registers;
//   * addresses are 32-bit buffer offsets.  dY (and X of a pointwise convolution) are plain matrices: a lane's
//     offset only grows by a constant per step and the descriptor's range check zero-fills the rows past the
//     end of the workgroup's pixel range.  The gathered X of a filter with taps keeps (image, row, column) per
//     piece up to date incrementally and swaps in an out-of-range offset for padding (hardware zero fill);
//   * the ring keeps STAGES - 1 pixel slices in flight under a counted vmcnt and ONE barrier per step.
// (Reference trigger: `loss.backward()`, /root/reference/2.distributed.py:231, which lands in cuDNN's
// backward-filter kernels there.)
#include <algorithm>
#include <type_traits>
#include "common.h"

namespace {

struct WRArgs {
  const void* dy;  // [M][Co]
  const void* x;   // gathered input
  float* dw;       // [Co][Ktot] fp32
  long long sn, sh, sw;
  int Hi, Wi, Ci, Ho, Wo, Co;
  int R, S, stride_h, stride_w, pad_h, pad_w;
  int M, Ktot;
  int pix_per_split;
  float* slab;     // optional [nsplit][Co][Ktot]: split-K partials stored instead of atomically added
  int r_lo, r_hi, s_lo, s_hi;
  const void* in_scale;   // XF kernels: the gradient is taken against relu(x * in_scale[c] + in_shift[c]), Ci values
  const void* in_shift;   // of x's type each (x is a conv output before its BatchNorm; no tap ever reads padding)
};

template <typename T> struct Mfma32r;
template <> struct Mfma32r<f16> {
  typedef f16x8 frag;
  static __device__ __forceinline__ f32x16 run(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};
template <> struct Mfma32r<bf16> {
  typedef bf16x8 frag;
  static __device__ __forceinline__ f32x16 run(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};

// One ds_read_b64_tr_b16 as a bare instruction.  The builtin form of this read counts as a possible LDS *write* in
// the compiler's wait-count pass, which then drains every LDS-DMA in flight (s_waitcnt vmcnt(0)) in front of the
// first fragment read of a step - the ring would never hold more than the tile being read.  Behind an asm statement
// the compiler sees no LDS access at all, so the kernel orders these reads itself: counted lgkmcnt waits (LDS
// operations return in order) followed by a scheduling barrier, before anything consumes the registers.
template <int OFF>
__device__ __forceinline__ s16x4 tr_read_raw(unsigned addr) {
  s16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <typename T>
__device__ __forceinline__ typename Mfma32r<T>::frag join_halves(s16x4 lo, s16x4 hi) {
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(typename Mfma32r<T>::frag, v);
}

// relu(v * sc + sf) on the eight values of a fragment, rounded like the normalise-on-load of the forward kernels
// (conv_igemm.hip, affine_relu8): fp16 as packed half FMAs, bf16 through fp32 with one rounding
__device__ __forceinline__ f16x8 affine_relu(f16x8 v, f16x8 sc, f16x8 sf) {
  const f16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
  return __builtin_elementwise_max(__builtin_elementwise_fma(v, sc, sf), zero);
}
__device__ __forceinline__ bf16x8 affine_relu(bf16x8 v, bf16x8 sc, bf16x8 sf) {
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (bf16)fmaxf(fmaf((float)v[e], (float)sc[e], (float)sf[e]), 0.f);
  return o;
}

// s_waitcnt immediate (gfx9 layout): vmcnt in bits 3:0 and 15:14, expcnt 6:4 (7 = do not wait), lgkmcnt 11:8
constexpr int wr_waitcnt(int vm, int lgkm) { return (vm & 15) | ((vm >> 4) << 14) | (7 << 4) | ((lgkm & 15) << 8); }

// LIN: X is a plain [M][Ci] matrix (1x1 filter, stride 1, no padding, dense NHWC).
// XF: X is normalised on its way into the MFMA.  After the transposing read a lane's B fragment holds eight pixels of
// ONE channel (its output column), so scale and shift are two per-lane constants per fragment and the transform is
// eight packed VALU operations next to sixteen-pass MFMAs.  Rows past the end of the pixel range arrive as zeros
// and become relu(shift), but their dY rows are zeros too; padding taps would not be (the host refuses those).
// WM: waves along the Co dimension of the tile (2 x 2, or 1 x 4 for the 64-row tile: every wave then keeps a 64 x 64
// sub-tile, two fragment reads per MFMA pair, instead of a 32-row sliver).
// Chunk swizzle of an LDS image by pixel row: 256 / 512-byte rows XOR their 64-byte segment index with px & 3, 128-byte
// rows (two per 256-byte bank row) with (px / 2) & 1 - either way the four pixel rows a 16-lane group of a transposing
// read touches land in four different bank quarters.
template <int PITCH> __device__ __forceinline__ int px_swizzle(int px) {
  return PITCH % 256 == 0 ? (px & 3) : ((px >> 1) & 1);
}
template <typename T, int TM, int TN, int BKP, int STAGES, bool LIN, bool XF, int WM = 2>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2)))
wgrad_ring_kernel(WRArgs a) {
  static_assert(sizeof(T) == 2 && (TM == 64 || TM == 128 || TM == 256) && (TN == 128 || TN == 256), "16-bit, 128/256/512-byte pixel rows");
  static_assert(WM == 1 || WM == 2, "four waves");
  static_assert(BKP % 16 == 0 && STAGES >= 2 && STAGES <= 4, "ring geometry");
  constexpr int PY = TM * 2, PX = TN * 2;                     // bytes per pixel row of the two LDS images
  constexpr int STAGE_BYTES = BKP * (PY + PX);
  constexpr int NPY = BKP * PY / 4096, NPX = BKP * PX / 4096;  // DMA instructions per wave and step, per operand
  static_assert(NPY >= 1 && NPX >= 1, "every wave moves at least one piece per operand");
  constexpr int PIECES = NPY + NPX;
  constexpr int WN = 4 / WM;
  constexpr int WTM = TM / WM, WTN = TN / WN, MI = WTM / 32, NI = WTN / 32;
  static_assert(MI >= 1 && NI >= 1, "every wave owns at least one 32 x 32 block");
  constexpr int KG = BKP / 16;                                 // 16-pixel MFMA groups per step
  constexpr unsigned OOB = 0x80000000u;
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ __attribute__((aligned(16))) unsigned char lds[STAGES * STAGE_BYTES];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef typename Mfma32r<T>::frag frag;
  const int t = threadIdx.x, lane = t & 63;
  const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wid / WN, wn = wid % WN;
  const int l31 = lane & 31, lh = lane >> 5;

  const int tiles_k = (a.Co + TM - 1) / TM;
  const int tk = blockIdx.x % tiles_k, tc = blockIdx.x / tiles_k;
  const int k0 = tk * TM, c0n = tc * TN;
  const int mbeg = blockIdx.y * a.pix_per_split;
  const int mend = min(a.M, mbeg + a.pix_per_split);
  const T* __restrict__ DY = reinterpret_cast<const T*>(a.dy);
  const T* __restrict__ X = reinterpret_cast<const T*>(a.x);

  // a tile whose columns all belong to padding-only taps has nothing to compute
  bool live = false;
  {
    const int last = min(a.Ktot, c0n + TN) - 1;
    for (int tap = c0n / a.Ci; tap <= last / a.Ci; ++tap) {
      const int tr = tap / a.S, ts = tap - tr * a.S;
      live = live || (tr >= a.r_lo && tr <= a.r_hi && ts >= a.s_lo && ts <= a.s_hi);
    }
  }
  const int nsteps = (live && mend > mbeg) ? (mend - mbeg + BKP - 1) / BKP : 0;

  // ---- DMA geometry: a wave instruction fills 1 KiB of LDS in lane order = 1024 / pitch pixel rows; the chunk
  // swizzle of the image (64-byte segments XOR (pixel & 3)) moves to the source side: the lane that fills slot s of
  // pixel row p fetches logical chunk s ^ ((p & 3) << 2).  p & 3 is the same for all pieces of a lane.
  constexpr int RPPY = 1024 / PY, CPRY = PY / 16, RPPX = 1024 / PX, CPRX = PX / 16;
  const int lrY = lane / CPRY, slY = lane % CPRY, lrX = lane / CPRX, slX = lane % CPRX;
  const int pxY0 = wid * RPPY + lrY, pxX0 = wid * RPPX + lrX;   // pixel row of the lane's first piece (+ 4 * RPP per piece)
  const int cY = slY ^ (px_swizzle<PY>(pxY0) << 2), cX = slX ^ (px_swizzle<PX>(pxX0) << 2);
  const long long rangeY = (long long)(mend - mbeg) * a.Co * 2;
  const auto rsrcY = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(DY) + (long long)mbeg * a.Co, 0,
                                                        (int)(rangeY > 0 ? rangeY : 0), 0x00020000);
  unsigned voY[NPY];
  {
    const bool ok = k0 + cY * 8 < a.Co;
#pragma unroll
    for (int i = 0; i < NPY; ++i)
      voY[i] = ok ? (unsigned)((((long long)(pxY0 + 4 * RPPY * i)) * a.Co + k0 + cY * 8) * 2) : OOB;
  }
  const unsigned stepY = (unsigned)((long long)BKP * a.Co * 2);
  // X: plain matrix (LIN) or gather
  const int kidx = c0n + cX * 8;
  const bool xok_c = kidx < a.Ktot;
  int xr = 0, xs = 0, xc = 0;
  if (xok_c) {
    const int sc = a.S * a.Ci;
    xr = kidx / sc;
    const int rem = kidx - xr * sc;
    xs = rem / a.Ci;
    xc = rem - xs * a.Ci;
  }
  const long long rangeX = LIN ? (long long)(mend - mbeg) * a.Ci * 2 : 0x7fffffffLL;
  const auto rsrcX = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(X) + (LIN ? (long long)mbeg * a.Ci : 0LL), 0,
                                                        (int)(rangeX > 0 ? rangeX : 0), 0x00020000);
  unsigned voX[NPX];
  int pn[NPX], pho[NPX], pwo[NPX];
  const int hw = a.Ho * a.Wo;
  const int step_n = BKP / hw, step_rem = BKP - step_n * hw;
  const int step_h = step_rem / a.Wo, step_w = step_rem - step_h * a.Wo;
  const unsigned stepX = (unsigned)((long long)BKP * a.Ci * 2);
#pragma unroll
  for (int i = 0; i < NPX; ++i) {
    const int px = pxX0 + 4 * RPPX * i;
    if (LIN) {
      voX[i] = xok_c ? (unsigned)(((long long)px * a.Ci + kidx) * 2) : OOB;
      pn[i] = pho[i] = pwo[i] = 0;
    } else {
      const int m = mbeg + px;
      pn[i] = m / hw;
      const int rem = m - pn[i] * hw;
      pho[i] = rem / a.Wo;
      pwo[i] = rem - pho[i] * a.Wo;
      voX[i] = OOB;
    }
  }
  const int wave_off = wid * 1024;
  int mstep = mbeg;   // first pixel of the next tile to request
  auto issue = [&](int ST) {
    unsigned char* sy = lds + ST * STAGE_BYTES + wave_off;
    unsigned char* sx = sy + BKP * PY;
#pragma unroll
    for (int i = 0; i < NPY; ++i) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcY, (lds_ptr)(sy + i * 4096), 16, voY[i], 0, 0, 0);
      voY[i] += stepY;   // (an out-of-range marker stays out of range: the pixel range is below 2 GiB)
    }
#pragma unroll
    for (int i = 0; i < NPX; ++i) {
      if (LIN) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcX, (lds_ptr)(sx + i * 4096), 16, voX[i], 0, 0, 0);
        voX[i] += stepX;
      } else {
        const int m = mstep + pxX0 + 4 * RPPX * i;
        const int hi = pho[i] * a.stride_h - a.pad_h + xr;
        const int wi = pwo[i] * a.stride_w - a.pad_w + xs;
        const bool ok = xok_c && m < mend && (unsigned)hi < (unsigned)a.Hi && (unsigned)wi < (unsigned)a.Wi;
        const unsigned off = (unsigned)(((long long)pn[i] * a.sn + (long long)hi * a.sh + (long long)wi * a.sw + xc) * 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcX, (lds_ptr)(sx + i * 4096), 16, ok ? off : OOB, 0, 0, 0);
        // this piece's pixel moves on by BKP for the next tile: (dn, dh, dw) with one carry each, no division
        pwo[i] += step_w;
        if (pwo[i] >= a.Wo) { pwo[i] -= a.Wo; ++pho[i]; }
        pho[i] += step_h;
        if (pho[i] >= a.Ho) { pho[i] -= a.Ho; ++pn[i]; }
        pn[i] += step_n;
      }
    }
    mstep += BKP;
  };

  // ---- fragment addresses (see wgrad_kernel): 16-lane group lg takes channel sub-block 16 * (lg & 1) and pixel
  // rows 8 * (lg >> 1) + {0..3}, {4..7} of a 16-pixel group; lane 4q + p of a group addresses pixel row q,
  // channels 4p .. 4p + 3
  const int li = lane & 15, lg = lane >> 4;
  const int prow0 = 8 * (lg >> 1) + (li >> 2);
  const int swA = px_swizzle<PY>(prow0) << 6, swB = px_swizzle<PX>(prow0) << 6;
  unsigned offA[MI], offB[NI];
#pragma unroll
  for (int i = 0; i < MI; ++i) offA[i] = prow0 * PY + (((wm * WTM + i * 32 + 16 * (lg & 1) + 4 * (li & 3)) * 2) ^ swA);
#pragma unroll
  for (int j = 0; j < NI; ++j)
    offB[j] = BKP * PY + prow0 * PX + (((wn * WTN + j * 32 + 16 * (lg & 1) + 4 * (li & 3)) * 2) ^ swB);

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[i][j][g] = 0.f;

  frag xsc[XF ? NI : 1], xsf[XF ? NI : 1];
  if constexpr (XF) {
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int col = c0n + wn * WTN + j * 32 + l31;
      const int ci = col < a.Ktot ? col % a.Ci : 0;
      const T sc = reinterpret_cast<const T*>(a.in_scale)[ci], sf = reinterpret_cast<const T*>(a.in_shift)[ci];
#pragma unroll
      for (int e = 0; e < 8; ++e) { xsc[j][e] = sc; xsf[j][e] = sf; }
    }
  }
  // fragment halves (pixel rows +0..3 and +4..7 of a group) as they come out of the reads; joined at the MFMAs
  s16x4 alo[2][MI], ahi[2][MI], blo[2][NI], bhi[2][NI];
  constexpr int READS = 2 * (MI + NI);   // LDS instructions of one 16-pixel group
  static_assert(READS <= 15, "a group's reads must fit the lgkmcnt field");
  const unsigned lds_base = (unsigned)(uintptr_t)(lds_ptr)lds;
  auto fetch = [&](auto buf_, unsigned sb, auto kk_) {
    constexpr int B = decltype(buf_)::value, kk = decltype(kk_)::value;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      alo[B][i] = tr_read_raw<kk * 16 * PY>(sb + offA[i]);
      ahi[B][i] = tr_read_raw<kk * 16 * PY + 4 * PY>(sb + offA[i]);
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      blo[B][j] = tr_read_raw<kk * 16 * PX>(sb + offB[j]);
      bhi[B][j] = tr_read_raw<kk * 16 * PX + 4 * PX>(sb + offB[j]);
    }
  };
  // MFMAs of the group in register set B once all but `pending` of the LDS reads issued so far have returned
  auto mfmas = [&](auto buf_, auto pending_) {
    constexpr int B = decltype(buf_)::value, pending = decltype(pending_)::value;
    __builtin_amdgcn_s_waitcnt(wr_waitcnt(63, pending));
    __builtin_amdgcn_sched_barrier(0);   // nothing that reads the fragment registers may move above the wait
    frag fa[MI], fb[NI];
#pragma unroll
    for (int i = 0; i < MI; ++i) fa[i] = join_halves<T>(alo[B][i], ahi[B][i]);
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      fb[j] = join_halves<T>(blo[B][j], bhi[B][j]);
      if constexpr (XF) fb[j] = affine_relu(fb[j], xsc[j], xsf[j]);
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = Mfma32r<T>::run(fa[i], fb[j], acc[i][j]);
    __builtin_amdgcn_sched_barrier(0);   // and the next group's reads stay behind these MFMAs' operand fetch
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using IR = std::integral_constant<int, READS>;
  auto compute = [&](int ST) {
    const unsigned sb = lds_base + ST * STAGE_BYTES;
    // reads run one group ahead of the MFMAs in a second register set
    fetch(I0{}, sb, I0{});
    if constexpr (KG >= 2) { fetch(I1{}, sb, I1{}); mfmas(I0{}, IR{}); } else { mfmas(I0{}, I0{}); }
    if constexpr (KG >= 3) { fetch(I0{}, sb, std::integral_constant<int, 2>{}); mfmas(I1{}, IR{}); }
    else if constexpr (KG == 2) { mfmas(I1{}, I0{}); }
    if constexpr (KG >= 4) { fetch(I1{}, sb, std::integral_constant<int, 3>{}); mfmas(I0{}, IR{}); mfmas(I1{}, I0{}); }
    else if constexpr (KG == 3) { mfmas(I0{}, I0{}); }
    static_assert(KG <= 4, "at most 64 pixels per step");
  };

  // ---- ring: tile kt lives in buffer kt % STAGES.  Top of a step: wait for this wave's pieces of tile kt (the
  // younger tiles stay in flight), barrier (tile kt is complete AND everybody is done reading tile kt - 1), request
  // tile kt + STAGES - 1 into the buffer tile kt - 1 just left, then the MFMAs of tile kt.
  int issued = 0;
  for (; issued < STAGES - 1 && issued < nsteps; ++issued) issue(issued);
  int kt = 0, st = 0, sti = (STAGES - 1) % STAGES;
  for (; kt < nsteps - (STAGES - 1); ++kt) {     // steady state: always STAGES - 2 younger tiles, always a request
    __builtin_amdgcn_s_waitcnt(wr_waitcnt((STAGES - 2) * PIECES, 0));
    __builtin_amdgcn_s_barrier();
    issue(sti);
    ++issued;
    compute(st);
    st = st + 1 == STAGES ? 0 : st + 1;
    sti = sti + 1 == STAGES ? 0 : sti + 1;
  }
  for (; kt < nsteps; ++kt) {                    // drain: nothing left to request
    const int younger = issued - kt - 1;
    if (younger >= 2) __builtin_amdgcn_s_waitcnt(wr_waitcnt(2 * PIECES, 0));
    else if (younger == 1) __builtin_amdgcn_s_waitcnt(wr_waitcnt(PIECES, 0));
    else __builtin_amdgcn_s_waitcnt(wr_waitcnt(0, 0));
    __builtin_amdgcn_s_barrier();
    compute(st);
    st = st + 1 == STAGES ? 0 : st + 1;
  }

  // ---- write-out: C/D map of the 32x32 MFMA: col = lane & 31, row = (g & 3) + 8 * (g >> 2) + 4 * (lane >> 5)
  if (a.slab != nullptr) {
    float* __restrict__ sl = a.slab + (long long)blockIdx.y * a.Co * a.Ktot;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int col = c0n + wn * WTN + j * 32 + l31;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          const int row = k0 + wm * WTM + i * 32 + (g & 3) + 8 * (g >> 2) + 4 * lh;
          if (row < a.Co && col < a.Ktot) sl[(long long)row * a.Ktot + col] = acc[i][j][g];
        }
      }
    return;
  }
  if (nsteps == 0) return;
  const bool alone = gridDim.y == 1;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int col = c0n + wn * WTN + j * 32 + l31;
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const int row = k0 + wm * WTM + i * 32 + (g & 3) + 8 * (g >> 2) + 4 * lh;
        if (row < a.Co && col < a.Ktot) {
          float* __restrict__ d = a.dw + (long long)row * a.Ktot + col;
          if (alone) *d += acc[i][j][g];
          else atomicAdd(d, acc[i][j][g]);
        }
      }
    }
#endif
}

__global__ void __launch_bounds__(256)
wr_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, long long n4, long long n, int nsplit) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    f32x4 s = *reinterpret_cast<const f32x4*>(dw + 4 * i);
    int k = 0;
    for (; k + 3 < nsplit; k += 4) {
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(slab + (long long)k * n + 4 * i);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(slab + (long long)(k + 1) * n + 4 * i);
      const f32x4 v2 = *reinterpret_cast<const f32x4*>(slab + (long long)(k + 2) * n + 4 * i);
      const f32x4 v3 = *reinterpret_cast<const f32x4*>(slab + (long long)(k + 3) * n + 4 * i);
      s += (v0 + v1) + (v2 + v3);
    }
    for (; k < nsplit; ++k) s += *reinterpret_cast<const f32x4*>(slab + (long long)k * n + 4 * i);
    *reinterpret_cast<f32x4*>(dw + 4 * i) = s;
  }
}

template <typename T, int TM, int TN, int BKP, int STAGES, bool XF, int WM = 2>
int launch_ring(WRArgs a, bool lin, int nsplit, bool det, long long slab_bytes, hipStream_t st) {
  const int tiles = ((a.Co + TM - 1) / TM) * ((a.Ktot + TN - 1) / TN);
  if (nsplit <= 0) {
    // two workgroups per CU in one round (the LDS ring of either tile admits two); every split must still run
    // enough steps for the ring to matter
    nsplit = 512 / tiles;
    const int maxsplit = (a.M + 8 * BKP - 1) / (8 * BKP);
    if (nsplit > maxsplit) nsplit = maxsplit;
    if (nsplit < 1) nsplit = 1;
  }
  if (det && nsplit > 32) nsplit = 32;
  int pps = (a.M + nsplit - 1) / nsplit;
  pps = ((pps + BKP - 1) / BKP) * BKP;
  nsplit = (a.M + pps - 1) / pps;
  a.pix_per_split = pps;
  const long long n = (long long)a.Co * a.Ktot;
  const bool slabs = nsplit > 1 && nsplit <= 32 && a.slab != nullptr && (long long)nsplit * n * 4 <= slab_bytes &&
                     (n % 4) == 0 && (((uintptr_t)a.dw | (uintptr_t)a.slab) & 15) == 0;
  if (!slabs) a.slab = nullptr;
  if (det && !slabs && nsplit > 1) {
    nsplit = 1;
    a.pix_per_split = ((a.M + BKP - 1) / BKP) * BKP;
  }
  // 32-bit offsets from the operand bases: the byte range a workgroup addresses must stay below 2 GiB
  if ((long long)a.pix_per_split * a.Co * 2 >= 0x7fffffffLL) return -10;
  if (lin) {
    if ((long long)a.pix_per_split * a.Ci * 2 >= 0x7fffffffLL) return -10;
    hipLaunchKernelGGL((wgrad_ring_kernel<T, TM, TN, BKP, STAGES, true, XF, WM>), dim3(tiles, nsplit), dim3(256), 0, st, a);
  } else {
    hipLaunchKernelGGL((wgrad_ring_kernel<T, TM, TN, BKP, STAGES, false, XF, WM>), dim3(tiles, nsplit), dim3(256), 0, st, a);
  }
  if (slabs) {
    const long long n4 = n / 4;
    long long blocks = (n4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(wr_reduce_kernel, dim3((int)blocks), dim3(256), 0, st, a.slab, a.dw, n4, n, nsplit);
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

int wgrad_ring_impl(int dtype, const void* dy, const void* x, float* dw, const void* in_scale, const void* in_shift,
                    int N, int Hi, int Wi, int Ci, long long sn, long long sh, long long sw, int Ho, int Wo, int Co,
                    int R, int S, int stride_h, int stride_w, int pad_h, int pad_w, int cfg, int nsplit,
                    float* workspace, long long workspace_bytes, hipStream_t stream) {
  if (dtype != PDA_F16 && dtype != PDA_BF16) return -6;
  if (Ci % 8 != 0 || Co % 8 != 0) return -1;
  if (((uintptr_t)x & 15) || ((uintptr_t)dy & 15)) return -5;
  WRArgs a;
  a.dy = dy; a.x = x; a.dw = dw; a.sn = sn; a.sh = sh; a.sw = sw;
  a.Hi = Hi; a.Wi = Wi; a.Ci = Ci; a.Ho = Ho; a.Wo = Wo; a.Co = Co; a.R = R; a.S = S;
  a.stride_h = stride_h; a.stride_w = stride_w; a.pad_h = pad_h; a.pad_w = pad_w;
  a.M = N * Ho * Wo; a.Ktot = R * S * Ci; a.pix_per_split = 0;
  a.slab = workspace;
  a.in_scale = in_scale; a.in_shift = in_shift;
  if (a.M <= 0) return 0;
  if ((long long)N * Ho * Wo > 0x7fffffffLL) return -4;
  {
    const long long rl = (long long)pad_h - (long long)(Ho - 1) * stride_h, rh = (long long)Hi - 1 + pad_h;
    const long long sl = (long long)pad_w - (long long)(Wo - 1) * stride_w, sh2 = (long long)Wi - 1 + pad_w;
    a.r_lo = (int)(rl < 0 ? 0 : rl); a.r_hi = (int)(rh > R - 1 ? R - 1 : rh);
    a.s_lo = (int)(sl < 0 ? 0 : sl); a.s_hi = (int)(sh2 > S - 1 ? S - 1 : sh2);
  }
  const bool xf = in_scale != nullptr;
  if (xf) {
    // the transform runs on whole fragments, so no tap may ever fall on padding
    if (in_shift == nullptr || pad_h != 0 || pad_w != 0) return -11;
    if ((long long)(Ho - 1) * stride_h + R > Hi || (long long)(Wo - 1) * stride_w + S > Wi) return -11;
  }
  // the gathered operand is addressed with 32-bit byte offsets from the tensor base
  {
    const long long hmax = std::min<long long>(Hi - 1, (long long)(Ho - 1) * stride_h - pad_h + R - 1);
    const long long wmax = std::min<long long>(Wi - 1, (long long)(Wo - 1) * stride_w - pad_w + S - 1);
    if (((long long)(N - 1) * sn + hmax * sh + wmax * sw + Ci) * 2 >= 0x7fffffffLL) return -10;
  }
  const bool lin = R == 1 && S == 1 && stride_h == 1 && stride_w == 1 && pad_h == 0 && pad_w == 0 && sw == Ci &&
                   sh == (long long)Wi * Ci && sn == (long long)Hi * Wi * Ci && Ho == Hi && Wo == Wi;
  const bool det = (cfg & 32) != 0;
  const int tile = cfg & 7;
#define PDA_WR(T, XF)                                                                                         \
  switch (tile) {                                                                                             \
    case 1: return launch_ring<T, 128, 128, 32, 4, XF>(a, lin, nsplit, det, workspace_bytes, stream);         \
    case 2: return launch_ring<T, 64, 256, 32, 3, XF, 1>(a, lin, nsplit, det, workspace_bytes, stream);       \
    case 4: return launch_ring<T, 256, 128, 32, 3, XF>(a, lin, nsplit, det, workspace_bytes, stream);         \
    case 5: return launch_ring<T, 128, 256, 32, 3, XF>(a, lin, nsplit, det, workspace_bytes, stream);         \
    case 6: return launch_ring<T, 128, 128, 64, 3, XF>(a, lin, nsplit, det, workspace_bytes, stream);         \
    default: return -2;                                                                                       \
  }
  if (dtype == PDA_F16) {
    if (xf) { PDA_WR(f16, true) }
    PDA_WR(f16, false)
  }
  if (xf) { PDA_WR(bf16, true) }
  PDA_WR(bf16, false)
#undef PDA_WR
}

}  // namespace

// Same contract as pda_conv_wgrad (dw zero-initialised or holding the gradient to accumulate into; workspace:
// optional scratch for split-K slabs).  cfg low 3 bits: 1 = 128(Co) x 128 tile, 32-pixel steps, four ring buffers;
// 2 = 64 x 256 on 1 x 4 waves, three buffers (layers with 64 output channels);
// 4 = 256 x 128 and 5 = 128 x 256, three buffers; 6 = 128 x 128 with 64-pixel steps, three buffers; bit 5 = run-to-run
// reproducible reduction.  <0: shape not covered (fp32, channel counts that do not fill 16-byte vectors, operands
// beyond the 32-bit offset range) - the caller uses pda_conv_wgrad.
PDA_EXPORT int pda_conv_wgrad_ring(int dtype, const void* dy, const void* x, float* dw, int N, int Hi, int Wi, int Ci,
                                   long long sn, long long sh, long long sw, int Ho, int Wo, int Co, int R, int S,
                                   int stride_h, int stride_w, int pad_h, int pad_w, int cfg, int nsplit,
                                   float* workspace, long long workspace_bytes, hipStream_t stream) {
  return wgrad_ring_impl(dtype, dy, x, dw, nullptr, nullptr, N, Hi, Wi, Ci, sn, sh, sw, Ho, Wo, Co, R, S, stride_h,
                         stride_w, pad_h, pad_w, cfg, nsplit, workspace, workspace_bytes, stream);
}

// As pda_conv_wgrad_xf: x is the producing conv's output before its BatchNorm and the gradient is taken against
// relu(x * in_scale + in_shift).  Only for filters that never touch padding (the 1x1 convs this hand-off feeds).
PDA_EXPORT int pda_conv_wgrad_ring_xf(int dtype, const void* dy, const void* x, float* dw, const void* in_scale,
                                      const void* in_shift, int N, int Hi, int Wi, int Ci, long long sn, long long sh,
                                      long long sw, int Ho, int Wo, int Co, int R, int S, int stride_h, int stride_w,
                                      int pad_h, int pad_w, int cfg, int nsplit, float* workspace,
                                      long long workspace_bytes, hipStream_t stream) {
  if (in_scale == nullptr || in_shift == nullptr) return -11;
  return wgrad_ring_impl(dtype, dy, x, dw, in_scale, in_shift, N, Hi, Wi, Ci, sn, sh, sw, Ho, Wo, Co, R, S, stride_h,
                         stride_w, pad_h, pad_w, cfg, nsplit, workspace, workspace_bytes, stream);
}
