// Implicit-GEMM convolution for gfx950 on MFMA: forward (fprop), input gradient
// (dgrad) and weight gradient (wgrad), NHWC activations, fp16 / bf16 / fp32.
//
// GEMM view (reference trigger: the `model(images)` /
// `loss.backward()` calls, /root/reference/2.distributed.py:213,231, which land
// in cuDNN there):
//   fprop : Y[m][n]  = sum_k A[m][k] * Wk[n][k]      m=(img,ho,wo) n=cout k=(r,s,cin)
//   dgrad : dX[m][n] = sum_k A'[m][k] * Wt[n][k]     m=(img,hi,wi) n=cin  k=(r,s,cout)
//   wgrad : dW[n][k] = sum_m dY[m][n] * A[m][k]      reduction over output pixels
// A is never materialised: every 16-byte chunk of an A row is gathered straight
// from the NHWC tensor (one (r,s) tap and BK contiguous channels per K-step).
//
// Tiling is for 64-lane waves: 256-thread workgroups = 4 waves in a 2x2 grid,
// each wave owning (BM/2)x(BN/2) of the block tile as 32x32 MFMA accumulators
// (v_mfma_f32_32x32x16_{f16,bf16}; v_mfma_f32_32x32x2_f32 for exact fp32).
// Operand tiles are register-staged (loads for step t+1 are issued before the
// MFMAs of step t and written to the other LDS buffer after them: one barrier
// per K-step) and XOR-swizzled at 16-byte granularity so the ds_read_b128
// fragment reads are bank-conflict free.  wgrad reduces over pixels, which is
// the strided dimension of both operands, so its fragments come from
// ds_read_b64_tr_b16 (hardware transposing LDS read) for the 16-bit types.
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include "common.h"

struct ConvArgs {
  const void* x;    // gathered tensor (fprop: input; dgrad: dY)
  const void* w;    // [Co][R*S*Ci], K contiguous
  void* y;          // [M][Co]
  const void* res;  // optional [M][Co] added in the epilogue
  const void* mask; // optional [M][Co]: output elements whose mask value is <= 0 are written as 0 (after res);
                    // with mask_bits: one byte per 16-byte vector of that tensor, bit e = element e is positive
  int mask_bits;
  int res_s2;       // dgrad with the fused epilogue: `res` is the compact [N][ceil(Ho/2)][ceil(Wo/2)][Co] gradient of a
                    // stride-2 1x1 shortcut - it is added at the pixels with even row and column, nothing elsewhere
  int stream_out;   // the output is far larger than the caches: non-temporal stores (host-decided by its size)
  // BatchNorm backward of the layer that produced this conv's input, folded into a dgrad epilogue:
  // bnc = that layer's pre-normalisation tensor [M][Co].  With bn_scale/bn_shift the layer's ReLU mask
  // (bnc*scale+shift > 0) is applied to the output; with psum/psq the per-tile column sums of the final
  // output v and of v*bnc are written there (instead of the fprop statistics sum(y), sum(y^2)).
  // fprop with bn_shift alone (bnc == nullptr): a per-output-channel bias added before the residual / ReLU
  // (inference with BatchNorm folded into the weights: w' = w * scale, bias = shift).
  const void* bnc;
  const float* bn_scale;
  const float* bn_shift;
  float* psum;      // optional [tiles_m][Co] per-tile column sums of the output
  float* psq;       // optional [tiles_m][Co] per-tile column sums of squares
  long long sn, sh, sw;  // element strides of the gathered tensor
  int Hi, Wi, Ci;        // bounds of the gathered tensor, channels per tap
  int Ho, Wo, Co;        // output pixels per image, GEMM N
  int R, S;
  int stride_h, stride_w, pad_h, pad_w;
  int sh_log2, sw_log2;  // dgrad only (power-of-two strides)
  int M;                 // GEMM rows (per parity class in the stride-2 dgrad mode)
  int relu;
  int Ha, Wb;            // stride-2 dgrad: pixels per parity class along h / w
  int stat_slots;        // 0: psum/psq hold one row per M tile; n > 0: n rows, float-atomic accumulated
  int r_lo, r_hi, s_lo, s_hi;  // taps that can touch a real pixel for at least one row (host-computed):
                               // on tiny feature maps (3x3 conv on a 1x1 or 2x2 map) most taps only see padding
  unsigned tap_pattern_h, tap_pattern_w;  // dgrad: bit k*stride set for every k (taps congruent to 0 mod stride)
  // fprop of a consumer that normalises its own input (XF instantiations): the gathered tensor is the producer's
  // conv output before its BatchNorm, and every loaded value becomes relu(v * in_scale[c] + in_shift[c]) on its
  // way into LDS (tables of Ci values in the tensor's own 16-bit type); padding taps and rows past M stay zero
  const void* in_scale;
  const void* in_shift;
  // split-K (pointwise fprop only, csplit > 0; the KS instantiations): workgroup (x, y) reduces channels [y * csplit, (y + 1) * csplit) and
  // adds its partial tile into the fp32 [M][Co] matrix at `y` with float atomics - for GEMMs with so few output
  // tiles (a Linear layer on a small batch) that one workgroup per tile would leave most of the chip idle
  int ksplit, csplit;
  unsigned long long* stamps;              // diagnostics (PDA_IGEMM_STAMPS=1): 8 s_memtime stamps per workgroup
  int ablate;                              // diagnostics (PDA_IGEMM_ABLATE), bits: 1 = no operand loads after the
                                           // prologue, 2 = no fragment reads / MFMAs, 4 = no epilogue (results are garbage either way)
};

// MODE 0: fprop.  MODE 1: dgrad, any power-of-two stride: every tap is visited and rows whose tap
// does not land on an output pixel of the forward conv load zeros.  MODE 2: dgrad for stride 2,
// split by the parity (h&1, w&1) of the input-gradient pixel: a pixel of one parity class only ever
// receives the taps r = (h+pad) mod 2, +2, ... (likewise s), so each class is its own small
// stride-1 GEMM over dY with 1/4 of the taps on average and no zero rows; one launch runs all
// four classes, the classes of one tile position as neighbouring workgroups.
enum { MODE_FPROP = 0, MODE_DGRAD = 1, MODE_DGRAD_S2 = 2 };

// s_waitcnt immediate (gfx9 layout): vmcnt in bits 3:0 and 15:14, expcnt 6:4 (7 = do not wait), lgkmcnt 11:8
constexpr int waitcnt_imm(int vm, int lgkm) { return (vm & 15) | ((vm >> 4) << 14) | (7 << 4) | ((lgkm & 15) << 8); }

template <typename T> struct Mfma32;
template <> struct Mfma32<f16> {
  typedef f16x8 frag;
  static __device__ __forceinline__ f32x16 run(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};
template <> struct Mfma32<bf16> {
  typedef bf16x8 frag;
  static __device__ __forceinline__ f32x16 run(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};

// relu(v * sc + sf) on the eight 16-bit values of one 16-byte chunk (sc, sf: the chunk's eight channels).
// fp16 runs as packed half-precision FMAs (one rounding, like the stand-alone pass that computes in fp32 and
// rounds once; the tables themselves are rounded to fp16); bf16 has no packed FMA and goes through fp32.
template <typename T> __device__ __forceinline__ u32x4 affine_relu8(u32x4 v, u32x4 sc, u32x4 sf);
template <> __device__ __forceinline__ u32x4 affine_relu8<f16>(u32x4 v, u32x4 sc, u32x4 sf) {
  // (as whole 8-element vectors: four v_pk_fma_f16 + four v_pk_max_f16)
  const f16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
  const f16x8 r = __builtin_elementwise_fma(__builtin_bit_cast(f16x8, v), __builtin_bit_cast(f16x8, sc),
                                            __builtin_bit_cast(f16x8, sf));
  return __builtin_bit_cast(u32x4, __builtin_elementwise_max(r, zero));
}
template <> __device__ __forceinline__ u32x4 affine_relu8<bf16>(u32x4 v, u32x4 sc, u32x4 sf) {
  u32x4 o;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const float lo = fmaxf(fmaf(__uint_as_float(v[d] << 16), __uint_as_float(sc[d] << 16), __uint_as_float(sf[d] << 16)), 0.f);
    const float hi = fmaxf(fmaf(__uint_as_float(v[d] & 0xffff0000u), __uint_as_float(sc[d] & 0xffff0000u),
                                __uint_as_float(sf[d] & 0xffff0000u)), 0.f);
    const unsigned short l16 = __builtin_bit_cast(unsigned short, (bf16)lo);
    const unsigned short h16 = __builtin_bit_cast(unsigned short, (bf16)hi);
    o[d] = (unsigned)l16 | ((unsigned)h16 << 16);
  }
  return o;
}
template <> __device__ __forceinline__ u32x4 affine_relu8<float>(u32x4 v, u32x4, u32x4) { return v; }

// rows per 256-byte LDS bank row and the chunk swizzle that spreads a 16-lane
// ds_read_b128 group over all 64 banks.
template <int ROWB> __device__ __forceinline__ int swz(int row) {
  constexpr int CPR = ROWB / 16;
  constexpr int RPB = 256 / ROWB;
  return (row / RPB) & (CPR - 1);
}

// Two waves per SIMD: the 128x128 tile sits right at the 256-register limit, and without the hint
// the stride-2 dgrad variant tips over it and runs at half the occupancy.
// Diagnostic builds (-DPDA_IGEMM_DIAG) honour ConvArgs::ablate; production builds compile the checks away.
#ifdef PDA_IGEMM_DIAG
#define PDA_ABL(bit_) ((a.ablate & (bit_)) != 0)
#else
#define PDA_ABL(bit_) false
#endif
#ifndef PDA_IGEMM_ATTR
#define PDA_IGEMM_ATTR __attribute__((amdgpu_waves_per_eu(2)))
#endif
// EX: the epilogue extras (res, mask, bnc) are compiled in; the plain instantiation keeps the lean
// write-out (the extras cost registers and scalar state even when their pointers are null).
// WM: waves along M (WM x 2 waves per workgroup).  STAGES = 0: register-staged operand tiles, two LDS
// buffers (the loads of step t+1 are issued before the MFMAs of step t and written to LDS after them).
// STAGES >= 3 (16-bit types): a ring of STAGES LDS buffers filled by LDS-DMA (buffer_load ... lds): the
// operands never pass through registers, the tile of step t+2 is requested while step t computes, and the
// only per-step synchronisation is one counted vmcnt + one barrier (see the main loop).
// XF (fprop, 16-bit types, register-staged tiles): the input is normalised on load, see ConvArgs::in_scale.
// SL (stride-1 "same" convolutions, e.g. 3x3 pad 1; LDS-DMA ring for the weights): the A operand of ALL taps of
// a channel slice comes from one LDS-resident slab of input pixels.  The tile's BM consecutive output pixels
// and their halo are addressed in zero-padded image coordinates, q = n*Hp*Wp + (h+pad)*Wp + (w+pad), so the pixel
// under tap (r, s) of output row m sits at slab slot q(m) - q_lo + (r-pad)*Wp + (s-pad): one slab load per slice
// (borders arrive as hardware zero-fill) serves R*S K-steps, and the L2 -> LDS traffic of the A operand drops by
// about that factor - the implicit-GEMM tiles above re-fetch every input pixel once per tap and run into the
// L2 -> CU bandwidth long before the matrix pipe is busy.
template <typename T, int BM, int BN, int ROWB, int MODE, bool EX, int WM = 2, int STAGES = 0, bool XF = false,
          bool SL = false, bool KS = false>
__global__ void __launch_bounds__(128 * WM) PDA_IGEMM_ATTR
igemm_kernel(ConvArgs a) {
  static_assert(!XF || (MODE == MODE_FPROP && (STAGES == 0 || SL) && sizeof(T) == 2 && !EX),
                "input normalisation: plain forward kernels with register-staged operands or the slab");
  static_assert(!SL || (STAGES == 4 && MODE != MODE_DGRAD_S2 && ROWB == 64), "slab: stride 1, 32-deep slices, 4-deep weight ring");
  constexpr int NT = 128 * WM;             // threads per workgroup
  constexpr bool DMA = STAGES > 0;
  constexpr int NST = DMA ? STAGES : 2;    // LDS buffers per operand
  constexpr int NSLOT = 512;               // SL: pixel slots per slab buffer (the last one is kept all-zero)
  static_assert(!DMA || (sizeof(T) == 2 && STAGES >= 3 && STAGES <= 4), "LDS-DMA ring: 16-bit operands, 3-4 buffers");
  constexpr bool DGRAD = MODE != MODE_FPROP;
  constexpr bool PAR = MODE == MODE_DGRAD_S2;
  constexpr int EPC = 16 / sizeof(T);      // elements per 16-byte chunk
  constexpr int BK = ROWB / sizeof(T);     // K elements per step
  constexpr int CPR = ROWB / 16;           // chunks per tile row
  constexpr int RPT = NT / CPR;            // tile rows covered by one pass of the workgroup
  constexpr int A_IT = BM / RPT;           // chunks per thread, A tile
  constexpr int B_IT = BN / RPT;
  constexpr int WTM = BM / WM, WTN = BN / 2;
  constexpr int MI = WTM / 32, NI = WTN / 32;
  static_assert(MI >= 1 && NI >= 1 && A_IT >= 1 && B_IT >= 1, "tile too small");

  // operand tiles (two stages); the 16-bit epilogue re-uses the space to stage the output tile
  constexpr int TILE_BYTES = SL ? (2 * NSLOT + NST * BN) * ROWB : NST * (BM + BN) * ROWB;
  constexpr int EPI_BYTES = sizeof(T) == 2 ? BM * (BN * (int)sizeof(T) + 16) : 0;
  constexpr int GRED_BYTES = (EX && sizeof(T) == 2) ? 2 * EPC * (NT + 8) * 4 : 0;  // BatchNorm-backward sums
  constexpr int LDS_BYTES0 = TILE_BYTES > EPI_BYTES ? TILE_BYTES : EPI_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES0 > GRED_BYTES ? LDS_BYTES0 : GRED_BYTES];
  unsigned char* ldsA = lds;
  unsigned char* ldsB = lds + (SL ? 2 * NSLOT : NST * BM) * ROWB;

  const int t = threadIdx.x;
  const int lane = t & 63, wid = t >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int l31 = lane & 31, lh = lane >> 5;
  // diagnostics: shader-clock stamps of this workgroup's phases (thread 0 only; a.stamps is null in normal runs)
#define PDA_STAMP(i_)                                                                                \
  do {                                                                                               \
    if (a.stamps != nullptr && t == 0 && blockIdx.x < 8192)                                          \
      a.stamps[blockIdx.x * 8 + (i_)] = __builtin_amdgcn_s_memtime();                                \
  } while (0)
  PDA_STAMP(0);

  const int tiles_n = (a.Co + BN - 1) / BN;
  const int tiles_m = (a.M + BM - 1) / BM;
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  int ph = 0, pw = 0, pcls = 0;
  if (PAR) {
    // the four classes of one tile position run as neighbouring workgroups (same XCD after the
    // remap): they gather the same dY region and together write whole runs of output pixels
    // The class rotates with the tile index: workgroups go to the shader engines of an XCD
    // round-robin, so a fixed "class = id mod 4" would hand every workgroup of the heaviest class
    // (4 taps of a 3x3, or the only live class of a 1x1) to the same engine.
    const int grp = bid >> 2;
    const int cls = (bid + grp) & 3;
    ph = cls >> 1;
    pw = cls & 1;
    pcls = cls;
    bid = grp;
  }
  const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  const T* __restrict__ X = reinterpret_cast<const T*>(a.x);
  const T* __restrict__ Wt = reinterpret_cast<const T*>(a.w);
  const int Ktot = a.R * a.S * a.Ci;

  const int kc = t & (CPR - 1);
  const int trow = t / CPR;

  // GEMM row -> output pixel.  PAR: rows enumerate the pixels of one parity class.
  const int row_hw = PAR ? a.Ha * a.Wb : a.Ho * a.Wo;
  const int row_w = PAR ? a.Wb : a.Wo;
  auto decode = [&](int m, int& n, int& ho, int& wo) -> bool {
    n = m / row_hw;
    const int rem = m - n * row_hw;
    ho = rem / row_w;
    wo = rem - ho * row_w;
    if (PAR) {
      ho = 2 * ho + ph;
      wo = 2 * wo + pw;
      return m < a.M && ho < a.Ho && wo < a.Wo;
    }
    return m < a.M;
  };

  // ---- operand addressing: buffer loads, 32-bit per-lane offset + wave-uniform scalar offset ----
  // Nothing about a thread's A rows changes along K except the tap (r, s) and the channel offset, and both
  // are wave-uniform: per row the thread keeps ONE byte offset (its pixel, relative to the first image of
  // the tile) and two bit masks (which r / which s land on a real pixel); the tap and channel terms ride in
  // the scalar offset of the load.  A tap that only sees padding (or a row past M) swaps the row offset for
  // one past the descriptor's range and the hardware returns zeros: no branch around the load, no 64-bit
  // address arithmetic in the loop, and a new select only when the tap changes.
  constexpr unsigned OOB = 0x80000000u;      // >= num_records of the descriptors below
  constexpr int ES = (int)sizeof(T);
  const int n_first = m0 / row_hw;           // first image this tile touches (wave-uniform)
  // scaled strides: a dgrad tap r moves the gathered dY pixel by -r/stride rows (only taps with an exact
  // quotient are live, see the masks), which is linear in r because sh, sw are multiples of the stride
  const long long shs = DGRAD ? (a.sh >> a.sh_log2) : a.sh;
  const long long sws = DGRAD ? (a.sw >> a.sw_log2) : a.sw;
  // the descriptor base sits before the tile's first image by the largest negative tap displacement, so
  // every per-lane offset and every scalar offset is non-negative
  const long long bias_el = DGRAD ? (long long)(a.R - 1) * shs + (long long)(a.S - 1) * sws
                                  : (long long)a.pad_h * a.sh + (long long)a.pad_w * a.sw;
  const auto rsrcA = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<T*>(X) + ((long long)n_first * a.sn - bias_el), 0, 0x7fffffff, 0x00020000);
  const auto rsrcB = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<T*>(Wt) + (long long)n0 * Ktot, 0, 0x7fffffff, 0x00020000);

  // LDS-DMA writes lane-linear (wave base + 16 * lane): the chunk swizzle of the LDS image moves to the
  // SOURCE side, i.e. the lane that fills chunk slot kc of a row fetches logical chunk kc ^ swz(row)
  // (the same for every pass of a thread: a pass moves on by a multiple of the swizzle period)
  static_assert(!DMA || (RPT % (CPR * (256 / ROWB)) == 0), "swizzle must repeat with the pass stride");
  const int kcs = DMA ? (kc ^ swz<ROWB>(trow)) : kc;
  unsigned apix[A_IT], ahm[A_IT], awm[A_IT];
  auto range_mask = [](int lo, int hi) -> unsigned {  // bits lo..hi (inclusive), empty when hi < lo
    return hi >= lo ? ((2u << hi) - (1u << lo)) : 0u;
  };
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int m = m0 + trow + i * RPT;
    int n, ho, wo;
    if (decode(m, n, ho, wo)) {
      const long long img = (long long)(n - n_first) * a.sn + kcs * EPC;
      if (DGRAD) {
        // live taps: 0 <= h + pad - r <= (Hi - 1) * stride and (h + pad - r) divisible by the stride
        const int ah = ho + a.pad_h, aw = wo + a.pad_w;
        apix[i] = (unsigned)((img + (long long)ah * shs + (long long)aw * sws) * ES);
        ahm[i] = range_mask(max(0, ah - (a.Hi - 1) * a.stride_h), min(a.R - 1, ah)) &
                 (a.tap_pattern_h << (ah & (a.stride_h - 1)));
        awm[i] = range_mask(max(0, aw - (a.Wi - 1) * a.stride_w), min(a.S - 1, aw)) &
                 (a.tap_pattern_w << (aw & (a.stride_w - 1)));
      } else {
        const int h0 = ho * a.stride_h, w0 = wo * a.stride_w;  // + r - pad is the input row
        apix[i] = (unsigned)((img + (long long)h0 * a.sh + (long long)w0 * a.sw) * ES);
        ahm[i] = range_mask(max(0, a.pad_h - h0), min(a.R - 1, a.Hi - 1 + a.pad_h - h0));
        awm[i] = range_mask(max(0, a.pad_w - w0), min(a.S - 1, a.Wi - 1 + a.pad_w - w0));
      }
    } else {
      apix[i] = 0u;
      ahm[i] = 0u;  // no tap is live
      awm[i] = 0u;
    }
  }
  unsigned boff[B_IT];
#pragma unroll
  for (int i = 0; i < B_IT; ++i) {
    const int nl = trow + i * RPT;
    boff[i] = (n0 + nl < a.Co) ? (unsigned)(((long long)nl * Ktot + kcs * EPC) * ES) : OOB;
  }

  // taps visited: all of them, or (PAR) those with the parity of this class' pixels
  // (both restricted to the window of taps that are not pure padding)
  const int r0 = PAR ? a.r_lo + ((a.r_lo ^ (ph + a.pad_h)) & 1) : a.r_lo;
  const int s0 = PAR ? a.s_lo + ((a.s_lo ^ (pw + a.pad_w)) & 1) : a.s_lo;
  constexpr int TSTEP = PAR ? 2 : 1;
  const int nr = r0 <= a.r_hi ? (a.r_hi - r0) / TSTEP + 1 : 0;
  const int ns = s0 <= a.s_hi ? (a.s_hi - s0) / TSTEP + 1 : 0;

  u32x4 ra[A_IT], rb[B_IT];
  unsigned avo[A_IT];  // per-lane offsets of the current tap (apix, or OOB where the tap is padding)
  int r = r0, s = s0, c0 = 0;
  int soA = 0, soB = 0;  // scalar byte offsets of the current K step (tap + channel slice), both operands
  // The per-step bookkeeping below is a handful of scalar instructions; everything per-lane happens in
  // set_tap(), once per tap.  (A wave issues one instruction every ~5 cycles whatever its kind, and what
  // does not sit in the shadow of an MFMA adds to the step: a hundred "cheap" instructions per step
  // cost as much as the step's sixteen MFMAs.)
  auto set_tap = [&]() {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) avo[i] = (((ahm[i] >> r) & (awm[i] >> s)) & 1u) ? apix[i] : OOB;
    // (readfirstlane: the values ARE wave-uniform, but once register pressure parks one of them in a VGPR
    // the compiler wraps every load that uses it as a scalar offset in a serialising "waterfall" loop)
    soA = __builtin_amdgcn_readfirstlane((int)((DGRAD ? (long long)(a.R - 1 - r) * shs + (long long)(a.S - 1 - s) * sws
                                                      : (long long)r * a.sh + (long long)s * a.sw) * ES));
    soB = __builtin_amdgcn_readfirstlane((r * a.S + s) * a.Ci * ES);
    c0 = 0;
  };
  set_tap();

  // XF: scale / shift of this thread's eight channels of the current K step ride along with the tile loads
  // (one 16-byte load each: the tables are tiny and stay in cache); `live` remembers which of the thread's
  // rows were real pixels when the step was loaded, because the tap may have moved on by the time the
  // registers are written to LDS
  u32x4 xsc = {0u, 0u, 0u, 0u}, xsf = {0u, 0u, 0u, 0u};
  unsigned xlive = 0u;
  const bool xsel = XF && (a.R * a.S > 1 || a.pad_h != 0 || a.pad_w != 0);
  const auto rsrcS = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(XF ? a.in_scale : nullptr), 0,
                                                        XF ? a.Ci * ES : 0, 0x00020000);
  const auto rsrcF = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(XF ? a.in_shift : nullptr), 0,
                                                        XF ? a.Ci * ES : 0, 0x00020000);
  auto gload = [&]() {
    // (default cache policy: marking the loads of a read-once 1x1 input non-temporal measured 0.6 % slower on the
    // ResNet-50 step)
#pragma unroll
    for (int i = 0; i < A_IT; ++i)
      ra[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrcA, avo[i], soA, 0));
#pragma unroll
    for (int i = 0; i < B_IT; ++i)
      rb[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrcB, boff[i], soB, 0));
    if constexpr (XF) {
      const int soC = __builtin_amdgcn_readfirstlane(c0 * ES);
      xsc = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrcS, kcs * 16, soC, 0));
      xsf = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrcF, kcs * 16, soC, 0));
      if (xsel) {
        xlive = 0u;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) xlive |= (avo[i] != OOB ? 1u : 0u) << i;
      }
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int row = trow + i * RPT;
      if constexpr (XF) {
        // pointwise convs have no padding taps: their only dead rows are those past M in the last tile, which
        // the statistics below skip explicitly, so the select is only paid where padding exists
        const u32x4 zero = {0u, 0u, 0u, 0u};
        const u32x4 v = affine_relu8<T>(ra[i], xsc, xsf);
        ra[i] = (xsel && !((xlive >> i) & 1u)) ? zero : v;
      }
      *reinterpret_cast<u32x4*>(ldsA + buf * BM * ROWB + row * ROWB + ((kc ^ swz<ROWB>(row)) << 4)) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      const int row = trow + i * RPT;
      *reinterpret_cast<u32x4*>(ldsB + buf * BN * ROWB + row * ROWB + ((kc ^ swz<ROWB>(row)) << 4)) = rb[i];
    }
  };
  // next K step: the next channel slice of this tap, or the first slice of the next tap
  auto advance = [&]() {
    c0 += BK;
    soA += BK * ES;
    soB += BK * ES;
    if (c0 >= a.Ci) {
      s += TSTEP;
      if (s > a.s_hi) { s = s0; r += TSTEP; }
      set_tap();
    }
  };

  // Output vectors of the 16-bit write-out: thread t owns vectors t, t+NT, ... of the BM x (BN/EPC) tile.
  constexpr int VPR = BN / EPC;      // 16-byte vectors per tile row
  constexpr int KI = BM * VPR / NT;  // vectors written by one thread
  static_assert(NT % VPR == 0, "a thread keeps one vector column for all of its rows");
  auto vec_offset = [&](int mbase, int k, long long& o) -> bool {
    const int v = t + NT * k;
    const int row = v / VPR, cv = v % VPR;
    const int m = mbase + row, n = n0 + cv * EPC;
    bool rok = m < a.M;
    long long orow = m;
    if (PAR) {
      int pn, pho, pwo;
      rok = decode(m, pn, pho, pwo);
      orow = ((long long)pn * a.Ho + pho) * a.Wo + pwo;
    }
    const bool ok = rok && n < a.Co;
    o = ok ? orow * a.Co + n : 0;
    return ok;
  };
  // The BatchNorm operand of the write-out is requested up front: its loads complete under the main loop
  // (32 registers) instead of stalling the epilogue of a workgroup that has nothing else left to do.
  constexpr bool VEC16 = sizeof(T) == 2;
  constexpr int KB = (EX && VEC16) ? KI : 1;
  Vec16<T> bv[KB];
  if constexpr (EX && VEC16) {
    if (a.bnc != nullptr && a.Co % EPC == 0) {
      const T* __restrict__ Bc0 = reinterpret_cast<const T*>(a.bnc);
#pragma unroll
      for (int k = 0; k < KI; ++k) {
        long long o;
        vec_offset(m0, k, o);  // a thread without a valid k-th vector reads element 0
        bv[k] = load16_stream(Bc0 + o);
      }
    }
  }

  // ... and so are the producer's scale/shift for the 8 columns of this thread (ReLU mask recomputation)
  float msc[EPC], msf[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) { msc[e] = 0.f; msf[e] = 0.f; }
  if constexpr (EX && VEC16) {
    const int nc = n0 + (t % VPR) * EPC;
    if (a.bnc != nullptr && a.bn_scale != nullptr && nc < a.Co && a.Co % EPC == 0) {
#pragma unroll
      for (int e = 0; e < EPC; e += 4) {
        const float4 s4 = *reinterpret_cast<const float4*>(a.bn_scale + nc + e);
        const float4 f4 = *reinterpret_cast<const float4*>(a.bn_shift + nc + e);
        msc[e] = s4.x; msc[e + 1] = s4.y; msc[e + 2] = s4.z; msc[e + 3] = s4.w;
        msf[e] = f4.x; msf[e + 1] = f4.y; msf[e + 2] = f4.z; msf[e + 3] = f4.w;
      }
    }
  }

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[i][j][g] = 0.f;

  int nk = nr * ns * (a.Ci / BK);  // 0 for a parity class no tap reaches (e.g. 1x1 stride 2)
  // (KS is its own instantiation: as a run-time branch in the common kernels these few lines moved the register
  // allocation of the main loop and cost the ResNet-50 step 4 %)
  if constexpr (KS) {
    const int cb = blockIdx.y * a.csplit;
    c0 = cb;
    soA += cb * ES;
    soB += cb * ES;
    nk = min(a.csplit, a.Ci - cb) / BK;
  }

  // MFMAs of one K step on the tiles at A_ / B_
  // (16-bit types: the 16-deep MFMA groups kk0 .. kk1-1 of the step; the DMA ring runs a step in two halves)
  auto compute = [&](const unsigned char* A_, const unsigned char* B_, int kk0, int kk1) {
    if constexpr (sizeof(T) == 2) {
      typedef typename Mfma32<T>::frag frag;
#pragma unroll
      for (int kk = kk0; kk < kk1; ++kk) {
        frag fa[MI], fb[NI];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const int row = wm * WTM + i * 32 + l31;
          fa[i] = *reinterpret_cast<const frag*>(A_ + row * ROWB + (((kk * 2 + lh) ^ swz<ROWB>(row)) << 4));
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const int row = wn * WTN + j * 32 + l31;
          fb[j] = *reinterpret_cast<const frag*>(B_ + row * ROWB + (((kk * 2 + lh) ^ swz<ROWB>(row)) << 4));
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j) acc[i][j] = Mfma32<T>::run(fa[i], fb[j], acc[i][j]);
      }
    } else {
      // fp32: one 16-byte chunk = 4 k values; lane half h owns k = 2h, 2h+1 of
      // the chunk (any k permutation is fine as long as A and B agree).
#pragma unroll
      for (int c = 0; c < CPR; ++c) {
        f32x2 fa[MI], fb[NI];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const int row = wm * WTM + i * 32 + l31;
          fa[i] = *reinterpret_cast<const f32x2*>(A_ + row * ROWB + ((c ^ swz<ROWB>(row)) << 4) + lh * 8);
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const int row = wn * WTN + j * 32 + l31;
          fb[j] = *reinterpret_cast<const f32x2*>(B_ + row * ROWB + ((c ^ swz<ROWB>(row)) << 4) + lh * 8);
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
          }
      }
    }
  };

  if constexpr (!DMA) {
    if (nk > 0) {
      gload();
      lstore(0);
    }
    __syncthreads();
    PDA_STAMP(2);  // first tile staged
    for (int kt = 0; kt < nk; ++kt) {
      const bool more = kt + 1 < nk;
      if (more) {
        advance();
        if (!PDA_ABL(1)) gload();
      }
      if (!PDA_ABL(2)) compute(ldsA + (kt & 1) * BM * ROWB, ldsB + (kt & 1) * BN * ROWB, 0, BK / 16);
      if (more) lstore((kt + 1) & 1);
      __syncthreads();
    }
  } else {
    // (device pass only: hipcc's host pass silently drops a kernel template whose body uses the LDS-DMA
    // builtin and leaves its launch stub undefined; the host needs the stub, not the body)
#if defined(__HIP_DEVICE_COMPILE__)
    // One wave instruction moves 64 chunks = 1 KiB: the wave's (64 / CPR) rows of one pass, contiguous in LDS.
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef typename Mfma32<T>::frag frag;
    if constexpr (SL) {
      // ---- slab main loop (see the SL note above the kernel) ----
      constexpr int S_IT = NSLOT * CPR / NT;   // slab DMA instructions per thread and channel slice
      constexpr int ZSLOT = NSLOT - 1;         // never filled with data: rows past M read their A operand here
      constexpr int PB = B_IT;                 // vector-memory instructions per thread: one weight tile ...
      constexpr int PS = S_IT + (XF ? 2 : 0);  // ... and one slab (XF: plus the slice's scale / shift vectors)
      static_assert(S_IT >= 1 && (NSLOT * ROWB) % (NT * 16) == 0, "whole passes of the workgroup fill a slab");
      const int wave_off = __builtin_amdgcn_readfirstlane(wid) * 1024;
      const int Wp = a.Wi + 2 * a.pad_w, HWp = (a.Hi + 2 * a.pad_h) * Wp;
      const int halo = a.pad_h * Wp + a.pad_w;
      const int nimg = a.M / row_hw;
      int q_lo;
      {
        int n, ho, wo;
        decode(m0, n, ho, wo);
        q_lo = __builtin_amdgcn_readfirstlane(n * HWp + (ho + a.pad_h) * Wp + (wo + a.pad_w) - halo);
      }
      const auto rsrcSl = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(X) + (long long)n_first * a.sn, 0, 0x7fffffff,
                                                             0x00020000);
      // slab slots this lane fills (the same ones for every slice): source pixel, or out of range for borders,
      // pixels past the last image and the zero slot
      unsigned soff[S_IT];
      const int kcl = lane & (CPR - 1);
      int lchunk = 0;   // logical chunk this lane fetches: the same for all of its pieces (a piece starts at a
                        // multiple of 16 slots, a whole number of swizzle periods)
#pragma unroll
      for (int it = 0; it < S_IT; ++it) {
        const int slot = (it * (NT / 64) + wid) * (1024 / ROWB) + lane / CPR;
        lchunk = kcl ^ swz<ROWB>(slot);
        const int q = q_lo + slot;
        const int n = q / HWp;
        const int rem = q - n * HWp;
        const int hp = rem / Wp;
        const int h = hp - a.pad_h, w = rem - hp * Wp - a.pad_w;
        const bool ok = slot != ZSLOT && n < nimg && (unsigned)h < (unsigned)a.Hi && (unsigned)w < (unsigned)a.Wi;
        soff[it] = ok ? (unsigned)(((long long)(n - n_first) * a.sn + (long long)h * a.sh + (long long)w * a.sw +
                                    lchunk * EPC) * ES)
                      : OOB;
      }
      // slab slot (before the tap offset) of each output row this lane feeds to the MFMAs; -1: row past M
      int sb[MI];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        int n, ho, wo;
        sb[i] = decode(m0 + wm * WTM + i * 32 + l31, n, ho, wo) ? n * HWp + (ho + a.pad_h) * Wp + (wo + a.pad_w) - q_lo : -1;
      }
      auto a_addr = [&](int i, int toff, int buf) -> unsigned {  // byte address of the row's K-group 0 fragment
        const int sl = sb[i] < 0 ? ZSLOT : sb[i] + toff;
        return (unsigned)(buf * NSLOT * ROWB + sl * ROWB + ((lh ^ swz<ROWB>(sl)) << 4));
      };
      unsigned offB[NI];
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int row = wn * WTN + j * 32 + l31;
        offB[j] = row * ROWB + ((lh ^ swz<ROWB>(row)) << 4);
      }
      const int ntap = a.R * a.S;
      const int nslice = a.Ci / BK;
      const int nks = ntap * nslice;
      // weight tiles are requested in (slice, tap) order
      int i_tp = 0, i_c0 = 0, issued = 0;
      auto issue_b = [&](int ST) {
        const int so = __builtin_amdgcn_readfirstlane((i_tp * a.Ci + i_c0) * ES);
        unsigned char* db = ldsB + ST * BN * ROWB + wave_off;
#pragma unroll
        for (int i = 0; i < B_IT; ++i)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, (lds_ptr)(db + i * RPT * ROWB), 16, boff[i], so, 0, 0);
        if (++i_tp == ntap) { i_tp = 0; i_c0 += BK; }
        ++issued;
      };
      u32x4 xsc = {0u, 0u, 0u, 0u}, xsf = {0u, 0u, 0u, 0u};
      const auto rsrcS = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(XF ? a.in_scale : nullptr), 0,
                                                            XF ? a.Ci * ES : 0, 0x00020000);
      const auto rsrcF = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(XF ? a.in_shift : nullptr), 0,
                                                            XF ? a.Ci * ES : 0, 0x00020000);
      auto issue_slab = [&](int sc) {
        const int so = __builtin_amdgcn_readfirstlane(sc * BK * ES);
        unsigned char* da = lds + (sc & 1) * NSLOT * ROWB + wave_off;
#pragma unroll
        for (int it = 0; it < S_IT; ++it)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcSl, (lds_ptr)(da + it * (NT / 64) * 1024), 16, soff[it], so, 0, 0);
        if constexpr (XF) {
          xsc = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrcS, lchunk * 16, so, 0));
          xsf = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrcF, lchunk * 16, so, 0));
        }
      };
      // XF: every lane normalises, in place, exactly the chunks its own DMA instructions brought in (so its own
      // vmcnt wait is all the synchronisation the pass needs; the step's barrier then publishes the result);
      // zero-filled chunks stay zero
      auto xf_slab = [&](int sc) {
        if constexpr (XF) {
          unsigned char* da = lds + (sc & 1) * NSLOT * ROWB + wave_off + lane * 16;
#pragma unroll
          for (int it = 0; it < S_IT; ++it) {
            if (soff[it] != OOB) {
              u32x4* pch = reinterpret_cast<u32x4*>(da + it * (NT / 64) * 1024);
              *pch = affine_relu8<T>(*pch, xsc, xsf);
            }
          }
        }
      };
      frag fa[2][MI], fb[2][NI];
      unsigned caddr[MI], naddr[MI];
      auto fetch = [&](auto buf_, const unsigned* aaddr, unsigned axor, int ST, unsigned bxor) {
        constexpr int B = decltype(buf_)::value;
        const unsigned char* sbp = ldsB + ST * BN * ROWB;
#pragma unroll
        for (int i = 0; i < MI; ++i) fa[B][i] = *reinterpret_cast<const frag*>(lds + (aaddr[i] ^ axor));
#pragma unroll
        for (int j = 0; j < NI; ++j) fb[B][j] = *reinterpret_cast<const frag*>(sbp + (offB[j] ^ bxor));
      };
      auto mfmas = [&](auto buf_) {
        constexpr int B = decltype(buf_)::value;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j) acc[i][j] = Mfma32<T>::run(fa[B][i], fb[B][j], acc[i][j]);
      };
      // tap offsets: fprop reads the pixel at +(r - pad), dgrad (the transposed conv) at +(pad - r)
      auto tap_off = [&](int tr, int ts) -> int {
        return DGRAD ? (a.pad_h - tr) * Wp + (a.pad_w - ts) : (tr - a.pad_h) * Wp + (ts - a.pad_w);
      };
      // prologue: slab 0 and the first weight tiles; everything lands before the first step
      issue_slab(0);
      for (int i = 0; i < STAGES && i < nks; ++i) issue_b(i);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      xf_slab(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      PDA_STAMP(2);
      int tr = 0, ts = 0, cs = 0;   // tap and slice of the running step
#pragma unroll
      for (int i = 0; i < MI; ++i) caddr[i] = a_addr(i, tap_off(0, 0), 0);
      if (nks > 0) fetch(std::integral_constant<int, 0>{}, caddr, 0u, 0, 0u);
      int st = 0;
      for (int kt = 0; kt < nks; ++kt) {
        const int tp = tr * a.S + ts;
        // the step after this one
        int ntr = tr, nts = ts + 1, ncs = cs;
        if (nts == a.S) { nts = 0; ++ntr; }
        if (ntr == a.R) { ntr = 0; ++ncs; }
        const int nst = st + 1 == STAGES ? 0 : st + 1;
        const bool more = kt + 1 < nks;
        const bool slab_next = tp == 0 && cs + 1 < nslice;   // this step requests the next slice's slab
        // K group 0: read group 1's fragments, multiply group 0's
        fetch(std::integral_constant<int, 1>{}, caddr, 32u, st, 32u);
        if (!PDA_ABL(2)) mfmas(std::integral_constant<int, 0>{});
        if (more) {
#pragma unroll
          for (int i = 0; i < MI; ++i) naddr[i] = a_addr(i, tap_off(ntr, nts), ncs & 1);
          if (XF && tp == ntap - 2 && cs + 1 < nslice) xf_slab(cs + 1);   // landed: three younger tiles have been awaited since
          // weight tile kt+1 (and with it everything older, the next slab included) has landed once at most the
          // younger requests remain: two weight tiles, plus the slab while it is younger than tile kt+1
          const int younger = issued - kt - 2;
          if (younger >= 2) {
            if (tp >= 1 && tp <= 3 && cs + 1 < nslice) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * PB + PS) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * PB) : "memory");
          } else {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
          }
          __builtin_amdgcn_s_barrier();
          if (issued < nks && !PDA_ABL(1)) issue_b(st);
          if (slab_next && !PDA_ABL(1)) issue_slab(cs + 1);
          // K group 1: read the next step's group 0, multiply group 1
          fetch(std::integral_constant<int, 0>{}, naddr, 0u, nst, 0u);
        }
        if (!PDA_ABL(2)) mfmas(std::integral_constant<int, 1>{});
#pragma unroll
        for (int i = 0; i < MI; ++i) caddr[i] = naddr[i];
        tr = ntr; ts = nts; cs = ncs; st = nst;
      }
    } else {
    constexpr int PIECES = A_IT + B_IT;  // DMA instructions per thread and tile
    constexpr int KK = BK / 16, KH = KK / 2;
    static_assert(KK == 2 || KK == 4, "a step is two halves of one or two 16-deep MFMA groups");
    const int wave_off = __builtin_amdgcn_readfirstlane(wid) * 1024;
    // fragment addresses of this lane inside one ring buffer, every (row group, kk) once for the whole kernel:
    // the chunk swizzle is an XOR, so it cannot ride in the instruction's immediate offset - the ring stage can
    unsigned offA[MI][KK], offB[NI][KK];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int row = wm * WTM + i * 32 + l31;
        offA[i][kk] = row * ROWB + (((kk * 2 + lh) ^ swz<ROWB>(row)) << 4);
      }
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int row = wn * WTN + j * 32 + l31;
        offB[j][kk] = NST * BM * ROWB + row * ROWB + (((kk * 2 + lh) ^ swz<ROWB>(row)) << 4);
      }
    }
    // Ring discipline: tile kt lives in buffer kt % STAGES; see step() below for the order of one step.
    // DMA data becomes visible to other waves' ds_reads only through the issuing waves' counted vmcnt wait
    // followed by a barrier; the barrier is the raw instruction (a __syncthreads() would drain the DMAs that
    // are meant to stay in flight).
    // (The ring stage stays a run-time scalar: one arm per stage with the stage as a constant makes the compiler
    // rotate the accumulators through a second register set - 128 more registers on the 256x256 tile, spills.)
    int issued = 0;
    auto issue = [&](int ST) {
      unsigned char* da = ldsA + ST * BM * ROWB + wave_off;
      unsigned char* db = ldsB + ST * BN * ROWB + wave_off;
#pragma unroll
      for (int i = 0; i < A_IT; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (lds_ptr)(da + i * RPT * ROWB), 16, avo[i], soA, 0, 0);
#pragma unroll
      for (int i = 0; i < B_IT; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcB, (lds_ptr)(db + i * RPT * ROWB), 16, boff[i], soB, 0, 0);
    };
    // Fragment reads run one 16-deep group AHEAD of the MFMAs, in a second register set: the reads of group
    // g+1 are issued before the MFMAs of group g (a wave issues in order and an MFMA waits for the pipe, so
    // reads placed after a group's MFMAs only start once the last of them has been issued, and their LDS
    // latency then sits between the groups with nothing to hide it).  The last group of a step prefetches
    // group 0 of the next tile, which the mid-step barrier has already made visible.
    frag fa[2][MI], fb[2][NI];
    auto fetch = [&](auto buf_, int ST, auto kk_) {
      constexpr int B = decltype(buf_)::value, kk = decltype(kk_)::value;
      const unsigned char* sa = lds + ST * BM * ROWB;
      const unsigned char* sb = lds + ST * BN * ROWB;
#pragma unroll
      for (int i = 0; i < MI; ++i) fa[B][i] = *reinterpret_cast<const frag*>(sa + offA[i][kk]);
#pragma unroll
      for (int j = 0; j < NI; ++j) fb[B][j] = *reinterpret_cast<const frag*>(sb + offB[j][kk]);
    };
    auto group = [&](auto kk_, int ST, int kt) {  // prefetch group kk+1 (or the next tile's group 0), MFMAs of group kk
      constexpr int kk = decltype(kk_)::value, B = kk & 1;
      if constexpr (kk + 1 < KK) {
        fetch(std::integral_constant<int, B ^ 1>{}, ST, std::integral_constant<int, kk + 1>{});
      } else {
        if (kt + 1 < nk) fetch(std::integral_constant<int, B ^ 1>{}, ST + 1 == STAGES ? 0 : ST + 1, std::integral_constant<int, 0>{});
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = Mfma32<T>::run(fa[B][i], fb[B][j], acc[i][j]);
    };
    auto group_next = [&](auto kk_, int ST) {  // group() of a step that is known not to be the last
      constexpr int kk = decltype(kk_)::value, B = kk & 1;
      if constexpr (kk + 1 < KK) fetch(std::integral_constant<int, B ^ 1>{}, ST, std::integral_constant<int, kk + 1>{});
      else fetch(std::integral_constant<int, B ^ 1>{}, ST + 1 == STAGES ? 0 : ST + 1, std::integral_constant<int, 0>{});
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = Mfma32<T>::run(fa[B][i], fb[B][j], acc[i][j]);
      // (pinned on both sides: MFMAs touch no memory, so nothing else keeps the step's wait + barrier from being
      // hoisted in among them, right behind the fragment reads it then has to wait for)
      __builtin_amdgcn_sched_barrier(0);
    };
    static_assert(KK % 2 == 0, "the fragment double buffer alternates per group");
    // One step = KK groups; the step's only barrier sits before the LAST group.  By then every read of tile
    // kt has been issued (the last group's fragments were prefetched during the group before) and, after the
    // lgkmcnt(0), has returned, so the barrier both publishes tile kt+1 (each wave waited for its own pieces)
    // and frees tile kt's buffer, into which tile kt+STAGES is requested at once: STAGES-1 tiles are in flight
    // while a step computes, with no more LDS than the ring itself.
    // The waits are the builtin, not inline asm: the compiler's own wait-count pass reads a builtin s_waitcnt and
    // knows that every fragment read has returned after it.  Behind an asm statement it assumed the reads of the
    // step's first half were still pending and made the MFMAs after the barrier wait for the fragment reads of
    // the NEXT tile that had just been issued (an `lgkmcnt(1)` four instructions after four ds_reads: one full LDS
    // latency exposed per step).
    auto wait_keep = [&](int younger) {  // wave-uniform: tiles younger than the awaited one that may stay in flight
#ifdef PDA_RING_LEGACY   // A/B build of the previous loop (python -m pytorch_distributed_amd._build --variant ringold PDA_RING_LEGACY)
      if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * PIECES) : "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PIECES) : "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      return;
#endif
      if (younger >= 2) __builtin_amdgcn_s_waitcnt(waitcnt_imm(2 * PIECES, 0));
      else if (younger == 1) __builtin_amdgcn_s_waitcnt(waitcnt_imm(PIECES, 0));
      else __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, 0));
    };
    auto step = [&](int ST, int kt) {
      if (!PDA_ABL(2)) {
        group(std::integral_constant<int, 0>{}, ST, kt);
        if constexpr (KK == 4) {
          group(std::integral_constant<int, 1>{}, ST, kt);
          group(std::integral_constant<int, 2>{}, ST, kt);
        }
      }
      if (kt + 1 < nk) {
        wait_keep(issued - kt - 2);  // tile kt+1
        __builtin_amdgcn_s_barrier();
        if (issued < nk) {
          advance();
          if (!PDA_ABL(1)) issue(ST);
          ++issued;
        }
      }
      if (!PDA_ABL(2)) group(std::integral_constant<int, KK - 1>{}, ST, kt);
    };
    // prologue: the first STAGES tiles
    for (; issued < STAGES && issued < nk; ++issued) {
      if (issued > 0) advance();
      issue(issued);
    }
    wait_keep(issued - 1);  // tile 0
    __builtin_amdgcn_s_barrier();
    PDA_STAMP(2);  // first tile staged
    if (nk > 0) fetch(std::integral_constant<int, 0>{}, 0, std::integral_constant<int, 0>{});
    int st = 0;
    int kt = 0;
    // steady state: a tile is requested in every step and STAGES - 2 younger ones stay in flight - the same
    // straight-line code every time (no branch between the fragment reads and the MFMAs that the compiler
    // would have to merge wait counts over)
#ifndef PDA_RING_LEGACY
    for (; kt < nk - STAGES; ++kt) {
      if (!PDA_ABL(2)) {
        group_next(std::integral_constant<int, 0>{}, st);
        if constexpr (KK == 4) {
          group_next(std::integral_constant<int, 1>{}, st);
          group_next(std::integral_constant<int, 2>{}, st);
        }
      }
      __builtin_amdgcn_s_waitcnt(waitcnt_imm((STAGES - 2) * PIECES, 0));
      __builtin_amdgcn_s_barrier();
      advance();
      if (!PDA_ABL(1)) issue(st);
      ++issued;
      if (!PDA_ABL(2)) group_next(std::integral_constant<int, KK - 1>{}, st);
      st = st + 1 == STAGES ? 0 : st + 1;
    }
#endif
    for (; kt < nk; ++kt) {
      step(st, kt);
      st = st + 1 == STAGES ? 0 : st + 1;
    }
    }  // !SL
#endif
    __syncthreads();  // the epilogue restages the output tile through the ring's memory
  }

  PDA_STAMP(3);  // main loop done
  if (PDA_ABL(4)) return;  // diagnostics: no write-out
  if constexpr (KS) {
    float* __restrict__ Yf = reinterpret_cast<float*>(a.y);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int col = n0 + wn * WTN + j * 32 + l31;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          const int row = m0 + wm * WTM + i * 32 + (g & 3) + 8 * (g >> 2) + 4 * lh;
          if (row < a.M && col < a.Co) atomicAdd(Yf + (long long)row * a.Co + col, acc[i][j][g]);
        }
      }
    return;
  }

  // ---- epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (g&3)+8*(g>>2)+4*(lane>>5)
  T* __restrict__ Y = reinterpret_cast<T*>(a.y);
  const T* __restrict__ Rz = EX ? reinterpret_cast<const T*>(a.res) : nullptr;
  const T* __restrict__ Mk = EX ? reinterpret_cast<const T*>(a.mask) : nullptr;
  const T* __restrict__ Bc = EX ? reinterpret_cast<const T*>(a.bnc) : nullptr;
  const bool stats = a.psum != nullptr;
  const bool gstats = stats && Bc != nullptr;  // sums of the final values and of value*bnc
  const bool remask = Bc != nullptr && a.bn_scale != nullptr;
  const int prow = PAR ? pcls * tiles_m + tm : tm;  // row of the per-tile statistics
  float csum[NI], csq[NI];
#pragma unroll
  for (int j = 0; j < NI; ++j) { csum[j] = 0.f; csq[j] = 0.f; }

  // 16-bit types with a vector-friendly Co: stage the tile through LDS so the global
  // stores (and the residual loads) are whole 16-byte vectors, 16 lanes per 256-byte row
  // segment, instead of 2-byte accesses scattered over 64-byte pieces.
  // Row -> pixel arithmetic of the epilogue must not be scheduled ahead of the main loop (it would
  // sit in registers for the whole loop and cost a wave of occupancy): hide m0 behind an opaque copy.
  int m0e = m0;
  if (PAR) asm volatile("" : "+v"(m0e));
  // (the instantiations with the epilogue extras only exist in the vector form for 16-bit types: the
  // host routes a Co that is not a multiple of 8 elsewhere)
  const bool vec_epi = (sizeof(T) == 2) && (EX || a.Co % EPC == 0);
  if (vec_epi) {
    constexpr int CP = BN * (int)sizeof(T) + 16;  // padded row pitch, keeps 16-byte alignment
    // The residual and the mask are requested here, before the accumulators are staged through LDS:
    // all KI vectors of a stream in one batch with no branch between them (a thread without a valid
    // k-th vector reads element 0 instead), so their latencies overlap each other and the staging pass
    // instead of adding up per row.
    constexpr int KE = EX ? KI : 1;
    long long eo[KE];  // < 0: this thread has no k-th vector
    if constexpr (EX) {
#pragma unroll
      for (int k = 0; k < KI; ++k)
        if (!vec_offset(m0e, k, eo[k])) eo[k] = -1;
    }
    Vec16<T> rv[KE], mv[KE];
    unsigned rskip = 0u;   // bit k: the k-th vector gets no residual (odd row or column of a stride-2 shortcut)
    if (Rz != nullptr) {
      if (EX && a.res_s2) {
        const int H2 = (a.Ho + 1) >> 1, W2 = (a.Wo + 1) >> 1;
#pragma unroll
        for (int k = 0; k < KE; ++k) {
          const int v = t + NT * k;
          const int m = eo[k] < 0 ? 0 : m0e + v / VPR;
          const int hq = m / a.Wo, pw = m - hq * a.Wo;
          const int pn = hq / a.Ho, ph = hq - pn * a.Ho;
          const bool even = eo[k] >= 0 && ((ph | pw) & 1) == 0;
          if (!even) rskip |= 1u << k;
          const long long ro = (((long long)pn * H2 + (ph >> 1)) * W2 + (pw >> 1)) * a.Co + n0 + (v % VPR) * EPC;
          rv[k] = load16_stream(Rz + (even ? ro : 0));
        }
      } else {
#pragma unroll
        for (int k = 0; k < KE; ++k) rv[k] = load16_stream(Rz + (eo[k] < 0 ? 0 : eo[k]));
      }
    }
    if (Mk != nullptr) {
      if (a.mask_bits) {
        // the bits travel in the first element slot of the vector's registers
        const unsigned char* __restrict__ Mb = reinterpret_cast<const unsigned char*>(a.mask);
#pragma unroll
        for (int k = 0; k < KE; ++k) {
          const u32x4 b = {(unsigned)Mb[(eo[k] < 0 ? 0 : eo[k]) / EPC], 0u, 0u, 0u};
          mv[k] = __builtin_bit_cast(Vec16<T>, b);
        }
      } else {
#pragma unroll
        for (int k = 0; k < KE; ++k) mv[k] = load16_stream(Mk + (eo[k] < 0 ? 0 : eo[k]));
      }
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int col = wn * WTN + j * 32 + l31;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          const int row = wm * WTM + i * 32 + (g & 3) + 8 * (g >> 2) + 4 * lh;
          const T tv = ElemTraits<T>::from_f32(acc[i][j][g]);
          *reinterpret_cast<T*>(lds + row * CP + col * (int)sizeof(T)) = tv;
          if (stats && !gstats) {
            // rows past M come from all-zero A rows, so they add nothing (XF: a normalised zero is not zero)
            const float fv = (!XF || m0e + row < a.M) ? ElemTraits<T>::to_f32(tv) : 0.f;
            csum[j] += fv;
            csq[j] += fv * fv;
          }
        }
      }
    __syncthreads();
    float g1[EPC], g2[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) { g1[e] = 0.f; g2[e] = 0.f; }
    // folded-BatchNorm bias of this thread's 8 columns (fprop only; the column is the same for all its vectors)
    const bool has_bias = !DGRAD && a.bn_shift != nullptr && a.bnc == nullptr;
    float bs[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) bs[e] = 0.f;
    if (has_bias) {
      const int nc = n0 + (t % VPR) * EPC;
      if (nc < a.Co) {
#pragma unroll
        for (int e = 0; e < EPC; e += 4) {
          const float4 b4 = *reinterpret_cast<const float4*>(a.bn_shift + nc + e);
          bs[e] = b4.x; bs[e + 1] = b4.y; bs[e + 2] = b4.z; bs[e + 3] = b4.w;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < KI; ++k) {
      const int v = t + NT * k;
      const int row = v / VPR, cv = v % VPR;
      long long o;
      bool ok;
      if constexpr (EX) {
        o = eo[k];
        ok = o >= 0;
      } else {
        ok = vec_offset(m0e, k, o);
      }
      if (ok) {
        Vec16<T> val = *reinterpret_cast<const Vec16<T>*>(lds + row * CP + cv * 16);
        if (has_bias) {
#pragma unroll
          for (int e = 0; e < EPC; ++e) val.v[e] = ElemTraits<T>::from_f32(ElemTraits<T>::to_f32(val.v[e]) + bs[e]);
        }
        if (Rz != nullptr && !((rskip >> k) & 1u)) {
#pragma unroll
          for (int e = 0; e < EPC; ++e)
            val.v[e] = ElemTraits<T>::from_f32(ElemTraits<T>::to_f32(val.v[e]) + ElemTraits<T>::to_f32(rv[EX ? k : 0].v[e]));
        }
        if (a.relu) {
#pragma unroll
          for (int e = 0; e < EPC; ++e)
            val.v[e] = ElemTraits<T>::from_f32(fmaxf(ElemTraits<T>::to_f32(val.v[e]), 0.f));
        }
        if (Mk != nullptr) {
          if (a.mask_bits) {
            const unsigned bits = __builtin_bit_cast(u32x4, mv[EX ? k : 0])[0];
#pragma unroll
            for (int e = 0; e < EPC; ++e)
              if (!((bits >> e) & 1u)) val.v[e] = ElemTraits<T>::from_f32(0.f);
          } else {
#pragma unroll
            for (int e = 0; e < EPC; ++e)
              if (!(ElemTraits<T>::to_f32(mv[EX ? k : 0].v[e]) > 0.f)) val.v[e] = ElemTraits<T>::from_f32(0.f);
          }
        }
        if (Bc != nullptr) {
#pragma unroll
          for (int e = 0; e < EPC; ++e) {
            const float cf = ElemTraits<T>::to_f32(bv[EX ? k : 0].v[e]);
            if (remask && !(cf * msc[e] + msf[e] > 0.f)) val.v[e] = ElemTraits<T>::from_f32(0.f);
            const float vf = ElemTraits<T>::to_f32(val.v[e]);
            g1[e] += vf;
            g2[e] += vf * cf;
          }
        }
        store16_maybe_stream(Y + o, val, a.stream_out);
      }
    }
    if (stats) __syncthreads();  // the statistics scratch below reuses the staging area
    if (gstats) {
      // threads t, t+VPR, t+2*VPR, ... hold the same 8 columns: fold them through LDS
      // image [2][EPC][NT + 8]: element e of thread t at e*(NT+8) + t, so the writes (consecutive t) and
      // the reads below (lane -> e = t%EPC, vector t/EPC: banks e*8 + t/EPC) are both conflict-free
      float* gred = reinterpret_cast<float*>(lds);
      constexpr int GS = NT + 8;
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        gred[e * GS + t] = g1[e];
        gred[(EPC + e) * GS + t] = g2[e];
      }
      __syncthreads();
      if (t < BN && n0 + t < a.Co) {
        float v1 = 0.f, v2 = 0.f;
        const int e = t % EPC, cvx = t / EPC;
#pragma unroll 4
        for (int w = 0; w < NT / VPR; ++w) {
          v1 += gred[e * GS + w * VPR + cvx];
          v2 += gred[(EPC + e) * GS + w * VPR + cvx];
        }
        if (a.stat_slots > 0) {  // few pre-zeroed rows shared by all tiles (see the statistics write-out below)
          const long long o = (long long)(prow % a.stat_slots) * a.Co + n0 + t;
          atomicAdd(a.psum + o, v1);
          atomicAdd(a.psq + o, v2);
        } else {
          a.psum[(long long)prow * a.Co + n0 + t] = v1;
          a.psq[(long long)prow * a.Co + n0 + t] = v2;
        }
      }
      PDA_STAMP(4);
      return;
    }
  } else if constexpr (!(EX && sizeof(T) == 2)) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int n = n0 + wn * WTN + j * 32 + l31;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          const int m = m0e + wm * WTM + i * 32 + (g & 3) + 8 * (g >> 2) + 4 * lh;
          float v = acc[i][j][g];
          bool rok = m < a.M;
          long long orow = m;
          if (PAR) {
            int pn, pho, pwo;
            rok = decode(m, pn, pho, pwo);
            orow = ((long long)pn * a.Ho + pho) * a.Wo + pwo;
          }
          const bool ok = rok && (n < a.Co);
          const long long o = orow * a.Co + n;
          if (!DGRAD && a.bn_shift != nullptr && a.bnc == nullptr && ok) v += a.bn_shift[n];
          if (Rz != nullptr && ok) v += ElemTraits<T>::to_f32(Rz[o]);
          if (a.relu) v = fmaxf(v, 0.f);
          if (Mk != nullptr && ok) {
            const bool pos = a.mask_bits ? ((reinterpret_cast<const unsigned char*>(a.mask)[o / EPC] >> (o % EPC)) & 1) != 0
                                         : ElemTraits<T>::to_f32(Mk[o]) > 0.f;
            if (!pos) v = 0.f;
          }
          float cf = 0.f;
          if (Bc != nullptr && ok) {
            cf = ElemTraits<T>::to_f32(Bc[o]);
            if (remask && !(cf * a.bn_scale[n] + a.bn_shift[n] > 0.f)) v = 0.f;
          }
          const T tv = ElemTraits<T>::from_f32(v);
          if (ok) Y[o] = tv;
          if (stats) {
            // statistics of the value as stored, so mean/var describe the tensor
            // that the normalisation pass will read back
            const float fv = ok ? ElemTraits<T>::to_f32(tv) : 0.f;
            csum[j] += fv;
            csq[j] += gstats ? fv * cf : fv * fv;
          }
        }
      }
  }

  if (stats) {
    float* sred = reinterpret_cast<float*>(lds);  // [WM][BN][2]; tiles are dead after the last barrier
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      float s1 = csum[j] + __shfl_xor(csum[j], 32, 64);
      float s2 = csq[j] + __shfl_xor(csq[j], 32, 64);
      if (lh == 0) {
        const int col = wn * WTN + j * 32 + l31;
        sred[(wm * BN + col) * 2 + 0] = s1;
        sred[(wm * BN + col) * 2 + 1] = s2;
      }
    }
    __syncthreads();
    if (t < BN && n0 + t < a.Co) {
      float v1 = 0.f, v2 = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) {
        v1 += sred[(w * BN + t) * 2];
        v2 += sred[(w * BN + t) * 2 + 1];
      }
      if (a.stat_slots > 0) {
        // few, pre-zeroed rows shared by many tiles: the finalize kernel that follows reads
        // stat_slots rows instead of one per tile (thousands for the early ResNet stages)
        const long long o = (long long)(prow % a.stat_slots) * a.Co + n0 + t;
        atomicAdd(a.psum + o, v1);
        atomicAdd(a.psq + o, v2);
      } else {
        a.psum[(long long)prow * a.Co + n0 + t] = v1;
        a.psq[(long long)prow * a.Co + n0 + t] = v2;
      }
    }
  }
  PDA_STAMP(4);  // write-out done
#undef PDA_STAMP
}

// ---------------------------------------------------------------- wgrad ----
struct WgradArgs {
  const void* dy;  // [M][Co]
  const void* x;   // gathered input
  float* dw;       // [Co][Ktot] fp32, accumulated with float atomics
  long long sn, sh, sw;
  int Hi, Wi, Ci;
  int Ho, Wo, Co;
  int R, S, stride_h, stride_w, pad_h, pad_w;
  int M, Ktot;
  int pix_per_split;  // multiple of the pixel step
  float* slab;        // optional [nsplit][Co][Ktot] scratch: split-K partials stored, not atomically added
  long long slab_bytes;
  int r_lo, r_hi, s_lo, s_hi;  // taps outside this window only ever multiply padding: their dW stays zero
  const void* in_scale;        // XF instantiations: x is normalised on load, relu(x * in_scale[c] + in_shift[c])
  const void* in_shift;        // (Ci values of x's 16-bit type; padding stays zero), see ConvArgs::in_scale
  int det;                     // host side only: run-to-run reproducible split-K reduction (no float atomics
                               // racing on one element): slabs + ordered reduce, or a single split
};

template <typename T> struct TrRead;
template <> struct TrRead<f16> {
  static __device__ __forceinline__ f16x8 read(const unsigned char* p0, const unsigned char* p1) {
    typedef s16x4 __attribute__((address_space(3))) * lp;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(p0));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(p1));
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(f16x8, v);
  }
};
template <> struct TrRead<bf16> {
  static __device__ __forceinline__ bf16x8 read(const unsigned char* p0, const unsigned char* p1) {
    typedef s16x4 __attribute__((address_space(3))) * lp;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(p0));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(p1));
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  }
};

// 64-byte-segment swizzle of a [pixel][channel] image so the four pixel rows of
// one transposed read fall on four different quarters of the 256-byte bank row.
template <int PITCH> __device__ __forceinline__ int trswz(int px) {
  if constexpr (PITCH % 256 == 0) return (px & 3) << 6;
  else if constexpr (PITCH == 128) return ((px >> 1) & 1) << 6;
  else return 0;
}

// 16 bytes of a row at element alignment (RAG instantiations: channel counts that are no multiple of the 16-byte
// vector, 58 / 116 / 232 in ShuffleNet).  Global memory takes the unaligned access in one instruction; `left`
// < EPC marks the chunk that would run past the end of the tensor, fetched element by element with zero fill.
template <typename T>
__device__ __forceinline__ u32x4 rag_load(const T* __restrict__ p, int left) {
  constexpr int EPC = 16 / sizeof(T);
  u32x4 r = {0u, 0u, 0u, 0u};
  if (left >= EPC) {
    __builtin_memcpy(&r, p, 16);
    return r;
  }
  if constexpr (sizeof(T) == 4) {
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = i < left ? __builtin_bit_cast(unsigned, p[i]) : 0u;
  } else {
    unsigned short e[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) e[i] = i < left ? __builtin_bit_cast(unsigned short, p[i]) : (unsigned short)0;
#pragma unroll
    for (int d = 0; d < 4; ++d) r[d] = (unsigned)e[2 * d] | ((unsigned)e[2 * d + 1] << 16);
  }
  return r;
}

// RAG: 1x1 filters over channel counts that do not fill 16-byte vectors.  A chunk that straddles the end of a
// pixel's channels carries the next pixel's first channels in its upper elements: those multiply into output
// rows / columns >= Co / Ci, which are never stored - every dW element depends on its own channel pair only.
template <typename T, int TM, int TN, int BKP, bool XF = false, bool RAG = false>
__global__ void __launch_bounds__(256)
wgrad_kernel(WgradArgs a) {
  constexpr int EPC = 16 / sizeof(T);
  constexpr int PY = TM * sizeof(T), PX = TN * sizeof(T);  // LDS row pitches (bytes)
  constexpr int CPY = PY / 16, CPX = PX / 16;              // chunks per pixel row
  constexpr int RPY = 256 / CPY, RPX = 256 / CPX;          // pixel rows per pass
  constexpr int Y_IT = (BKP + RPY - 1) / RPY, X_IT = (BKP + RPX - 1) / RPX;
  constexpr int WTM = TM / 2, WTN = TN / 2;
  constexpr int MI = WTM / 32, NI = WTN / 32;
  static_assert(MI >= 1 && NI >= 1, "tile too small");
  static_assert(CPY <= 256 && CPX <= 256, "tile too wide");

  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * BKP * (PY + PX)];
  unsigned char* ldsY = lds;
  unsigned char* ldsX = lds + 2 * BKP * PY;

  const int t = threadIdx.x;
  const int lane = t & 63, wid = t >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int l31 = lane & 31, lh = lane >> 5;

  const int tiles_k = (a.Co + TM - 1) / TM;
  const int tk = blockIdx.x % tiles_k, tc = blockIdx.x / tiles_k;
  const int k0 = tk * TM, c0n = tc * TN;
  const int mbeg = blockIdx.y * a.pix_per_split;
  const int mend = min(a.M, mbeg + a.pix_per_split);

  const T* __restrict__ DY = reinterpret_cast<const T*>(a.dy);
  const T* __restrict__ X = reinterpret_cast<const T*>(a.x);

  // dY loader: chunk column fixed per thread
  const int ycc = t % CPY, ypx = t / CPY;
  const bool yok_c = (k0 + ycc * EPC) < a.Co;
  // X loader: chunk column -> (r, s, c) fixed per thread
  const int xcc = t % CPX, xpx = t / CPX;
  const int kidx = c0n + xcc * EPC;
  const bool xok_c = kidx < a.Ktot;
  int xr = 0, xs = 0, xc = 0;
  if (xok_c) {
    const int sc = a.S * a.Ci;
    xr = kidx / sc;
    const int rem = kidx - xr * sc;
    xs = rem / a.Ci;
    xc = rem - xs * a.Ci;
  }
  const int hw = a.Ho * a.Wo;
  // XF: the thread's chunk column is one tap and eight channels for the whole kernel, so their coefficients are
  // loaded once
  u32x4 xsc = {0u, 0u, 0u, 0u}, xsf = {0u, 0u, 0u, 0u};
  if constexpr (XF) {
    if (xok_c) {
      xsc = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(a.in_scale) + xc);
      xsf = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(a.in_shift) + xc);
    }
  }

  // (image, ho, wo) of every pixel row this thread gathers, kept up to date incrementally: a step
  // moves BKP pixels on, i.e. (dn, dh, dw) with one carry each - no integer division in the loop
  // (a division costs ~40 VALU instructions; eight of them per step outweighed the step's MFMAs).
  const int step_n = BKP / hw, step_rem = BKP - step_n * hw;
  const int step_h = step_rem / a.Wo, step_w = step_rem - step_h * a.Wo;
  int pn[X_IT], pho[X_IT], pwo[X_IT];
#pragma unroll
  for (int i = 0; i < X_IT; ++i) {
    const int m = mbeg + xpx + i * RPX;
    pn[i] = m / hw;
    const int rem = m - pn[i] * hw;
    pho[i] = rem / a.Wo;
    pwo[i] = rem - pho[i] * a.Wo;
  }

  u32x4 ry[Y_IT], rx[X_IT];
  unsigned xlive = 0u;  // XF: which of rx[] hold real pixels (the others stay exact zeros)
  auto gload = [&](int mstep) {
    if constexpr (XF) xlive = 0u;
#pragma unroll
    for (int i = 0; i < Y_IT; ++i) {
      const int px = ypx + i * RPY;
      const int m = mstep + px;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (px < BKP && m < mend && yok_c) {
        const T* __restrict__ p = DY + (long long)m * a.Co + k0 + ycc * EPC;
        if constexpr (RAG) v = rag_load<T>(p, m == a.M - 1 ? a.Co - (k0 + ycc * EPC) : EPC);
        else v = *reinterpret_cast<const u32x4*>(p);
      }
      ry[i] = v;
    }
#pragma unroll
    for (int i = 0; i < X_IT; ++i) {
      const int px = xpx + i * RPX;
      const int m = mstep + px;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (px < BKP && m < mend && xok_c) {
        const int hi = pho[i] * a.stride_h - a.pad_h + xr;
        const int wi = pwo[i] * a.stride_w - a.pad_w + xs;
        if ((unsigned)hi < (unsigned)a.Hi && (unsigned)wi < (unsigned)a.Wi) {
          const T* __restrict__ p = X + (long long)pn[i] * a.sn + (long long)hi * a.sh + (long long)wi * a.sw + xc;
          if constexpr (RAG) v = rag_load<T>(p, m == a.M - 1 ? a.Ci - xc : EPC);
          else v = *reinterpret_cast<const u32x4*>(p);
          if constexpr (XF) xlive |= 1u << i;
        }
      }
      rx[i] = v;
      // advance this row by BKP pixels for the next call (calls come in step order)
      pwo[i] += step_w;
      if (pwo[i] >= a.Wo) { pwo[i] -= a.Wo; ++pho[i]; }
      pho[i] += step_h;
      if (pho[i] >= a.Ho) { pho[i] -= a.Ho; ++pn[i]; }
      pn[i] += step_n;
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < Y_IT; ++i) {
      const int px = ypx + i * RPY;
      if (px < BKP) {
        const int bo = (sizeof(T) == 2) ? ((ycc * 16) ^ trswz<PY>(px)) : ycc * 16;
        *reinterpret_cast<u32x4*>(ldsY + buf * BKP * PY + px * PY + bo) = ry[i];
      }
    }
#pragma unroll
    for (int i = 0; i < X_IT; ++i) {
      const int px = xpx + i * RPX;
      if (px < BKP) {
        const int bo = (sizeof(T) == 2) ? ((xcc * 16) ^ trswz<PX>(px)) : xcc * 16;
        // (transformed here, after the MFMAs of the running step, when the load has long arrived)
        if constexpr (XF) { if ((xlive >> i) & 1u) rx[i] = affine_relu8<T>(rx[i], xsc, xsf); }
        *reinterpret_cast<u32x4*>(ldsX + buf * BKP * PX + px * PX + bo) = rx[i];
      }
    }
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int g = 0; g < 16; ++g) acc[i][j][g] = 0.f;

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
  if (nsteps > 0) {
    gload(mbeg);
    lstore(0);
  }
  __syncthreads();

  const int li = lane & 15, lg = lane >> 4;
  for (int st = 0; st < nsteps; ++st) {
    const bool more = st + 1 < nsteps;
    if (more) gload(mbeg + (st + 1) * BKP);
    const unsigned char* Yb = ldsY + (st & 1) * BKP * PY;
    const unsigned char* Xb = ldsX + (st & 1) * BKP * PX;
    if constexpr (sizeof(T) == 2) {
      typedef typename Mfma32<T>::frag frag;
#pragma unroll
      for (int kk = 0; kk < BKP / 16; ++kk) {
        // 16-lane group lg: channel sub-block 16*(lg&1), pixel rows 8*(lg>>1)+{0..3},{4..7};
        // lane 4q+p of a group addresses pixel row q, channels 4p..4p+3 and
        // receives channel (lane&15) of the four rows.
        const int prow = kk * 16 + 8 * (lg >> 1) + (li >> 2);
        frag fa[MI], fb[NI];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const int col = wm * WTM + i * 32 + 16 * (lg & 1) + 4 * (li & 3);
          const unsigned char* p0 = Yb + prow * PY + ((col * 2) ^ trswz<PY>(prow));
          const unsigned char* p1 = Yb + (prow + 4) * PY + ((col * 2) ^ trswz<PY>(prow + 4));
          fa[i] = TrRead<T>::read(p0, p1);
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const int col = wn * WTN + j * 32 + 16 * (lg & 1) + 4 * (li & 3);
          const unsigned char* p0 = Xb + prow * PX + ((col * 2) ^ trswz<PX>(prow));
          const unsigned char* p1 = Xb + (prow + 4) * PX + ((col * 2) ^ trswz<PX>(prow + 4));
          fb[j] = TrRead<T>::read(p0, p1);
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j) acc[i][j] = Mfma32<T>::run(fa[i], fb[j], acc[i][j]);
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < BKP / 2; ++kk) {
        const int prow = kk * 2 + lh;
        float fa[MI], fb[NI];
#pragma unroll
        for (int i = 0; i < MI; ++i)
          fa[i] = *reinterpret_cast<const float*>(Yb + prow * PY + (wm * WTM + i * 32 + l31) * 4);
#pragma unroll
        for (int j = 0; j < NI; ++j)
          fb[j] = *reinterpret_cast<const float*>(Xb + prow * PX + (wn * WTN + j * 32 + l31) * 4);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
    }
    if (more) lstore((st + 1) & 1);
    __syncthreads();
  }

  if (a.slab != nullptr) {
    // split-K partial of this workgroup as plain stores into its own slab (zeros if the split is
    // empty); wgrad_reduce_kernel sums the slabs into dW afterwards.  Float atomics would move the
    // same bytes at about a fifth of the store rate.
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
  // one split: this workgroup is the only writer of its dW tile, so a plain read-add-write does (several times the
  // rate of the float atomics - the weight gradient of a Linear layer on a small batch is all write-out)
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
}

// dw[i] += sum over the nsplit slabs; n is a multiple of 4.
__global__ void __launch_bounds__(256)
wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, long long n4, long long n, int nsplit) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4;
       i += (long long)gridDim.x * blockDim.x) {
    f32x4 s = *reinterpret_cast<const f32x4*>(dw + 4 * i);
    int k = 0;
    for (; k + 3 < nsplit; k += 4) {  // four independent 16-byte loads in flight
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

// ------------------------------------------------------- weight layouts ----
// out[c][r][s][k] = in[k][r][s][c]  (KRSC -> CRSK, the dgrad B operand), with cast.
template <typename TI, typename TO>
__global__ void krsc_to_crsk_kernel(const TI* __restrict__ in, TO* __restrict__ out, int K, int RS, int C) {
  const long long total = (long long)K * RS * C;
  for (long long o = blockIdx.x * (long long)blockDim.x + threadIdx.x; o < total;
       o += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(o % K);
    const long long t2 = o / K;
    const int rs = (int)(t2 % RS);
    const int c = (int)(t2 / RS);
    out[o] = ElemTraits<TO>::from_f32(ElemTraits<TI>::to_f32(in[((long long)k * RS + rs) * C + c]));
  }
}

// The same layout change for a whole list of weights in one launch (every conv of a model, once per
// optimizer step).  meta: int64 [ntensors][5] = {src, dst, K, RS, C}; blk: int32 [nblocks][4] =
// {tensor, rs, k tile, c tile}; one block moves a 64(k) x 64(c) tile through LDS so both the reads
// (along c) and the writes (along k) are contiguous.
template <typename TI, typename TO>
__global__ void __launch_bounds__(256)
krsc_to_crsk_batched_kernel(const long long* __restrict__ meta, const int* __restrict__ blk) {
  __shared__ TO tile[64][65];
  const int4 b = reinterpret_cast<const int4*>(blk)[blockIdx.x];
  const long long* m = meta + (long long)b.x * 5;
  const TI* __restrict__ in = reinterpret_cast<const TI*>(m[0]);
  TO* __restrict__ out = reinterpret_cast<TO*>(m[1]);
  const int K = (int)m[2], RS = (int)m[3], C = (int)m[4];
  const int rs = b.y, k0 = b.z * 64, c0 = b.w * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const int k = k0 + ty + 4 * i, c = c0 + tx;
    if (k < K && c < C)
      tile[ty + 4 * i][tx] = ElemTraits<TO>::from_f32(ElemTraits<TI>::to_f32(in[((long long)k * RS + rs) * C + c]));
  }
  __syncthreads();
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const int c = c0 + ty + 4 * i, k = k0 + tx;
    if (k < K && c < C) out[((long long)c * RS + rs) * K + k] = tile[tx][ty + 4 * i];
  }
}

// ------------------------------------------------------------- host side ----
// Diagnostics: with PDA_IGEMM_STAMPS=1 every igemm workgroup (block id < 8192) leaves 8 shader-clock stamps here.
static unsigned long long* pda_conv_stamp_buffer() {
  static unsigned long long* buf = [] {
    const char* e = getenv("PDA_IGEMM_STAMPS");
    unsigned long long* p = nullptr;
    if (e && atoi(e) != 0 && hipMalloc(&p, 8192 * 8 * sizeof(unsigned long long)) != hipSuccess) p = nullptr;
    return p;
  }();
  return buf;
}
PDA_EXPORT int pda_conv_stamps_read(unsigned long long* host_out) {  // [8192][8]; -1 when stamping is off
  unsigned long long* b = pda_conv_stamp_buffer();
  if (b == nullptr) return -1;
  return (int)hipMemcpy(host_out, b, 8192 * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
}

static int ilog2_exact(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return ((1 << l) == v) ? l : -1;
}

template <typename T, int BM, int BN, int ROWB, int WM = 2, int STAGES = 0, bool SL = false>
static int launch_igemm(const ConvArgs& a, int dgrad, hipStream_t st) {
  const int tiles = ((a.M + BM - 1) / BM) * ((a.Co + BN - 1) / BN);
  constexpr int NT = 128 * WM;
  const bool ex = a.res != nullptr || a.mask != nullptr || a.bnc != nullptr;
  if (ex && sizeof(T) == 2 && a.Co % (16 / (int)sizeof(T)) != 0) return -8;
  if (a.csplit > 0 && (a.in_scale != nullptr || dgrad != MODE_FPROP)) return -13;
  if (a.in_scale != nullptr) {
    // input normalised on load: plain forward launches of the register-staged 16-bit tiles only
    if constexpr (sizeof(T) == 2 && (STAGES == 0 || SL)) {
      if (dgrad != MODE_FPROP || ex || a.in_shift == nullptr) return -11;
      hipLaunchKernelGGL((igemm_kernel<T, BM, BN, ROWB, MODE_FPROP, false, WM, STAGES, true, SL>), dim3(tiles), dim3(NT), 0, st, a);
      return pda_check_launch();
    } else {
      return -11;
    }
  }
#define PDA_IGEMM_LAUNCH(MODE_, EX_, GRID_)                                                                        \
  hipLaunchKernelGGL((igemm_kernel<T, BM, BN, ROWB, MODE_, EX_, WM, STAGES, false, SL>), dim3(GRID_), dim3(NT), 0, st, a)
  if (dgrad == MODE_DGRAD_S2) {
    if constexpr (SL) {
      return -12;
    } else {
      if (ex) PDA_IGEMM_LAUNCH(MODE_DGRAD_S2, true, 4 * tiles);
      else PDA_IGEMM_LAUNCH(MODE_DGRAD_S2, false, 4 * tiles);
    }
  } else if (dgrad == MODE_DGRAD) {
    if (ex) PDA_IGEMM_LAUNCH(MODE_DGRAD, true, tiles);
    else PDA_IGEMM_LAUNCH(MODE_DGRAD, false, tiles);
  } else if (a.csplit > 0) {
    if constexpr (STAGES == 0 && !SL) {
      if (ex) return -13;
      hipLaunchKernelGGL((igemm_kernel<T, BM, BN, ROWB, MODE_FPROP, false, WM, STAGES, false, SL, true>), dim3(tiles, a.ksplit),
                         dim3(NT), 0, st, a);
    } else {
      return -13;
    }
  } else {
    if (ex) PDA_IGEMM_LAUNCH(MODE_FPROP, true, tiles);
    else PDA_IGEMM_LAUNCH(MODE_FPROP, false, tiles);
  }
#undef PDA_IGEMM_LAUNCH
  return pda_check_launch();
}

// Fewest 256x128 tiles for which the big tile is chosen automatically (0 = never, the default: the
// per-layer gain below did not show in the ResNet-50 step time, 10.56-10.62k vs 10.59k images/s).
static const long long kBigTiles = [] {
  const char* e = getenv("PDA_IGEMM_BIG_TILES");
  return e ? atoll(e) : 0LL;
}();

// cfg: low 3 bits 0 = auto, 1 = 128x128, 2 = 128x64, 3 = 64x64, 4 = 256x128 (each of the 2x2 waves owns
// 128x64 outputs); bit 3 forces 64-byte K rows.  16-bit types only, LDS-DMA operand ring: 5 = 256x128 with
// 4x2 waves and three 64-deep stages, 6 = 128x128 with 2x2 waves and four 32-deep stages, 7 = 256x256 with
// 4x2 waves (64x128 outputs each) and four 32-deep stages, 13 (5 + 8) = 256x128 with 2x2 waves and three 32-deep stages.
// big_ok: 16-bit plain kernel.  With the 128x128 tile every wave re-reads a 64-row A and a 64-row B slice
// per 16 MFMAs and waits on those reads; a 128x64 wave tile moves 25 % fewer LDS bytes per MFMA and measured
// 5-10 % faster in isolation on the deep layers at 28x28 and 14x14 - as long as there are enough tiles to fill the chip
// twice, and with 32-deep K steps (the 64-deep version needs 96 KB of LDS: one workgroup per CU).
static int pick_tile(int M, int Co, int cfg, bool big_ok = false) {
  if (cfg & 64) return 8;  // slab tile (256x128), see dispatch_igemm
  int tile = cfg & 7;
  if (tile == 0) {
    const long long t128 = (long long)((M + 127) / 128) * ((Co + 127) / 128);
    const long long t256 = (long long)((M + 255) / 256) * ((Co + 127) / 128);
    if (big_ok && kBigTiles > 0 && Co % 128 == 0 && t256 >= kBigTiles) tile = 4;
    else if (Co <= 64) tile = 2;
    else if (t128 < 192 && M >= 64) tile = (Co % 128 == 0 && t128 * 2 >= 192) ? 2 : 3;
    else tile = 1;
  }
  return tile;
}

static const long long kShallowK = [] {
  const char* e = getenv("PDA_IGEMM_SHALLOW_K");
  return e ? atoll(e) : 576LL;
}();

template <typename T>
static int dispatch_igemm(const ConvArgs& a, int dgrad, int cfg, hipStream_t st) {
  constexpr int ES = sizeof(T);
  const bool k128 = (a.Ci % (128 / ES)) == 0;
  const bool k64 = (a.Ci % (64 / ES)) == 0;
  if (!k64) return -1;
  // shallow GEMMs (K = R*S*Ci up to 576: the 1x1 convs off 64..256 channels and the 3x3 on 64) run a
  // handful of K steps at most; 32-deep steps halve the LDS tile (more workgroups per CU) and give even a
  // K = 64 layer two pipelined steps instead of one load -> compute -> store sequence (measured per layer
  // on the ResNet-50 shapes, fp16: 5-25 % faster there, slower from K = 1024 up)
  // (not with the epilogue extras: that instantiation sits at two waves per SIMD either way, and measured
  // slightly slower with the short steps)
  const bool extras = a.res != nullptr || a.mask != nullptr || a.bnc != nullptr;
  const bool shallow = ES == 2 && (cfg & 7) == 0 && !extras && (long long)a.R * a.S * a.Ci <= kShallowK;
  const int tile = pick_tile(a.M, a.Co, cfg, ES == 2 && !extras);
  const bool use128 = k128 && (cfg & 8) == 0 && !shallow && !((cfg & 7) == 0 && tile == 4);
  const bool ring4 = tile == 5 && (cfg & 8) != 0;  // 256x128 ring with 2x2 waves
  if ((tile == 4 || tile == 7 || ring4) && extras) return -9;  // the epilogue extras are not built for 16 vectors per thread
  if (tile == 8) {
    // cfg bit 6: 256x128 slab tile (8 waves): odd square-ish filters with "same" padding at stride 1, 32-channel
    // slices; the slots a tile of 256 consecutive output pixels can reach (its rows and images in zero-padded
    // coordinates plus the halo on both sides) must fit one slab buffer, whose last slot stays zero
    if constexpr (ES == 2) {
      if (dgrad == MODE_DGRAD_S2 || a.stride_h != 1 || a.stride_w != 1 || a.Hi != a.Ho || a.Wi != a.Wo) return -12;
      if (a.R != 2 * a.pad_h + 1 || a.S != 2 * a.pad_w + 1 || a.R * a.S < 5) return -12;
      const long long Wp = a.Wi + 2 * a.pad_w, halo = (long long)a.pad_h * Wp + a.pad_w;
      const long long rows_crossed = (255 + a.Wo - 1) / a.Wo, imgs_crossed = (255 + (long long)a.Ho * a.Wo - 1) / ((long long)a.Ho * a.Wo);
      const long long span = 255 + rows_crossed * 2 * a.pad_w + imgs_crossed * 2 * a.pad_h * Wp + 2 * halo + 1;
      if (span > 511) return -12;
      return launch_igemm<T, 256, 128, 64, 4, 4, true>(a, dgrad, st);
    }
    return -12;
  }
  if (tile >= 5) {
    if (a.in_scale != nullptr) return -11;  // the LDS-DMA tiles move operands past the registers
    if constexpr (ES == 2) {
      // 5 + 8: the same 256x128 outputs on 2x2 waves of 128x64 (three fragment reads per four MFMAs instead of
      // four) with 32-deep steps in three buffers - the geometry of the ring weight-gradient kernel
      if (ring4) return launch_igemm<T, 256, 128, 64, 2, 3>(a, dgrad, st);
      if (tile == 5 && k128) return launch_igemm<T, 256, 128, 128, 4, 3>(a, dgrad, st);
      if (tile == 6) return launch_igemm<T, 128, 128, 64, 2, 4>(a, dgrad, st);
      if (tile == 7) return launch_igemm<T, 256, 256, 64, 4, 4>(a, dgrad, st);
    }
    return -2;
  }
  switch (tile) {
    case 1: return use128 ? launch_igemm<T, 128, 128, 128>(a, dgrad, st) : launch_igemm<T, 128, 128, 64>(a, dgrad, st);
    case 2: return use128 ? launch_igemm<T, 128, 64, 128>(a, dgrad, st) : launch_igemm<T, 128, 64, 64>(a, dgrad, st);
    case 3: return use128 ? launch_igemm<T, 64, 64, 128>(a, dgrad, st) : launch_igemm<T, 64, 64, 64>(a, dgrad, st);
    case 4: return use128 ? launch_igemm<T, 256, 128, 128>(a, dgrad, st) : launch_igemm<T, 256, 128, 64>(a, dgrad, st);
    default: return -2;
  }
}

// Block-tile rows used for a given cfg/shape (python sizes the stats partials with it).
PDA_EXPORT int pda_conv_igemm_bm(int M, int Co, int cfg, int dtype, int extras) {
  const int tile = pick_tile(M, Co, cfg, dtype != PDA_F32 && !extras);
  return tile == 3 ? 64 : (tile == 4 || tile == 5 || tile == 7 || tile == 8) ? 256 : 128;
}

// Rows of per-tile sums a dgrad launch with bnc/psum writes: N x Hx x Wx x Cx is the dX tensor.
PDA_EXPORT int pda_conv_dgrad_stat_rows(int N, int Hx, int Wx, int Cx, int stride_h, int stride_w, int cfg) {
  const bool par = stride_h == 2 && stride_w == 2 && (cfg & 16) == 0;
  const long long M = par ? (long long)N * ((Hx + 1) / 2) * ((Wx + 1) / 2) : (long long)N * Hx * Wx;
  const int tile = pick_tile((int)M, Cx, cfg);
  const int bm = tile == 3 ? 64 : (tile == 4 || tile == 5 || tile == 7 || tile == 8) ? 256 : 128;
  return (int)((M + bm - 1) / bm) * (par ? 4 : 1);
}

// Returns 0 on success, <0 if the shape is not covered by the MFMA kernels
// (caller falls back to the direct kernels), >0 = hipError_t.
static int conv_igemm_impl(int dtype, int dgrad, const void* x, const void* w, void* y, const void* res,
                           const void* mask, const void* bnc, const float* bn_scale, const float* bn_shift,
                           float* psum, float* psq, int N, int Hi, int Wi, int Ci, long long sn,
                           long long sh, long long sw, int Ho, int Wo, int Co, int R, int S, int stride_h,
                           int stride_w, int pad_h, int pad_w, int relu, int cfg, int stat_slots,
                           const void* in_scale, const void* in_shift, hipStream_t stream, int mask_bits = 0,
                           int ksplit = 0) {
  ConvArgs a;
  a.in_scale = in_scale; a.in_shift = in_shift;
  a.ksplit = 1; a.csplit = 0;
  if (ksplit >= 1) {
    if (dgrad || R != 1 || S != 1 || res != nullptr || mask != nullptr || bnc != nullptr || bn_shift != nullptr ||
        psum != nullptr || relu || Ci % 32 != 0)
      return -13;
    // channel ranges in multiples of 128 elements (a whole number of K steps for every tile), none of them empty
    a.csplit = ((Ci + ksplit - 1) / ksplit + 127) / 128 * 128;
    a.ksplit = (Ci + a.csplit - 1) / a.csplit;
  }
  a.mask_bits = mask != nullptr ? (mask_bits & 1) : 0;
  a.res_s2 = (mask_bits >> 1) & 1;
  if (a.res_s2 && (res == nullptr || dgrad == 0 || stride_h != 1 || stride_w != 1 || dtype == PDA_F32 || Co % 8 != 0))
    return -14;   // the compact residual exists in the vector epilogue of the stride-1 dgrad only
  a.stream_out = (long long)N * Ho * Wo * Co * (dtype == PDA_F32 ? 4 : 2) >= pda_stream_min_bytes();
  if ((((uintptr_t)in_scale | (uintptr_t)in_shift) & 15) != 0) return -5;
  a.stat_slots = stat_slots;
  a.x = x; a.w = w; a.y = y; a.res = res; a.mask = mask; a.bnc = bnc; a.bn_scale = bn_scale; a.bn_shift = bn_shift;
  a.psum = psum; a.psq = psq;
  a.sn = sn; a.sh = sh; a.sw = sw;
  a.Hi = Hi; a.Wi = Wi; a.Ci = Ci; a.Ho = Ho; a.Wo = Wo; a.Co = Co;
  a.R = R; a.S = S; a.stride_h = stride_h; a.stride_w = stride_w; a.pad_h = pad_h; a.pad_w = pad_w;
  a.sh_log2 = ilog2_exact(stride_h); a.sw_log2 = ilog2_exact(stride_w);
  a.M = N * Ho * Wo; a.relu = relu;
  a.Ha = a.Wb = 0;
  if (a.M <= 0 || Co <= 0) return 0;
  if (bnc != nullptr && !dgrad) return -7;
  if (dgrad && (a.sh_log2 < 0 || a.sw_log2 < 0)) return -3;
  if ((long long)N * Ho * Wo > 0x7fffffffLL) return -4;
  {
    // taps that reach a real pixel for some output row; fprop: 0 <= ho*stride - pad + r < Hi,
    // dgrad (Hi = dY height, Ho = dX height): 0 <= h + pad - r <= (Hi-1)*stride
    long long rl, rh, sl, sh2;
    if (dgrad) {
      rl = (long long)pad_h - (long long)(Hi - 1) * stride_h; rh = (long long)Ho - 1 + pad_h;
      sl = (long long)pad_w - (long long)(Wi - 1) * stride_w; sh2 = (long long)Wo - 1 + pad_w;
    } else {
      rl = (long long)pad_h - (long long)(Ho - 1) * stride_h; rh = (long long)Hi - 1 + pad_h;
      sl = (long long)pad_w - (long long)(Wo - 1) * stride_w; sh2 = (long long)Wi - 1 + pad_w;
    }
    a.r_lo = (int)(rl < 0 ? 0 : rl); a.r_hi = (int)(rh > R - 1 ? R - 1 : rh);
    a.s_lo = (int)(sl < 0 ? 0 : sl); a.s_hi = (int)(sh2 > S - 1 ? S - 1 : sh2);
  }
  if (dgrad) dgrad = MODE_DGRAD;
  if (dgrad && stride_h == 2 && stride_w == 2 && (cfg & 16) == 0) {  // cfg bit 4 forces the generic path
    dgrad = MODE_DGRAD_S2;
    a.Ha = (Ho + 1) / 2;
    a.Wb = (Wo + 1) / 2;
    a.M = N * a.Ha * a.Wb;
  }
  {
    // the kernels address both operands with 32-bit byte offsets from a per-tile base (see igemm_kernel):
    // everything one tile can reach must lie within 2 GiB of that base
    if (R > 32 || S > 32) return -10;
    if (dgrad && (sh % stride_h != 0 || sw % stride_w != 0)) return -10;
    const long long es = dtype == PDA_F32 ? 4 : 2;
    const long long row_hw = dgrad == MODE_DGRAD_S2 ? (long long)a.Ha * a.Wb : (long long)Ho * Wo;
    const long long imgs = 255 / row_hw + 2;  // images a 256-row tile can touch
    // rows / columns of the gathered tensor a tile can reach past its first image (Hi, Wi may be "unbounded"
    // markers for pre-padded inputs, so bound them by what the output extent can touch)
    const long long hs = dgrad ? (long long)Hi + R : std::min((long long)Hi, (long long)Ho * stride_h + R) + R + pad_h;
    const long long ws = dgrad ? (long long)Wi + S : std::min((long long)Wi, (long long)Wo * stride_w + S) + S + pad_w;
    const long long reach = (imgs * sn + hs * sh + ws * sw + Ci) * es;
    if (reach >= 0x7fffffffLL || 256LL * R * S * Ci * es >= 0x7fffffffLL) return -10;
    auto pattern = [](int st) { unsigned p = 0; for (int k = 0; k < 32; k += st) p |= 1u << k; return p; };
    a.tap_pattern_h = pattern(stride_h);
    a.tap_pattern_w = pattern(stride_w);
    static const int kAblate = [] { const char* e = getenv("PDA_IGEMM_ABLATE"); return e ? atoi(e) : 0; }();
    a.ablate = kAblate;
    a.stamps = pda_conv_stamp_buffer();
  }
  if (((uintptr_t)x & 15) || ((uintptr_t)w & 15)) return -5;
  if ((((uintptr_t)bn_scale | (uintptr_t)bn_shift) & 15) != 0) return -5;
  if ((((uintptr_t)res | (uintptr_t)bnc) & 15) != 0 || (!a.mask_bits && ((uintptr_t)mask & 15) != 0)) return -5;
  switch (dtype) {
    case PDA_F16: return dispatch_igemm<f16>(a, dgrad, cfg, stream);
    case PDA_BF16: return dispatch_igemm<bf16>(a, dgrad, cfg, stream);
    case PDA_F32: return dispatch_igemm<float>(a, dgrad, cfg, stream);
    default: return -6;
  }
}

PDA_EXPORT int pda_conv_igemm(int dtype, int dgrad, const void* x, const void* w, void* y, const void* res,
                              const void* mask, const void* bnc, const float* bn_scale, const float* bn_shift,
                              float* psum, float* psq, int N, int Hi, int Wi, int Ci, long long sn,
                              long long sh, long long sw, int Ho, int Wo, int Co, int R, int S, int stride_h,
                              int stride_w, int pad_h, int pad_w, int relu, int cfg, int stat_slots,
                              hipStream_t stream) {
  return conv_igemm_impl(dtype, dgrad, x, w, y, res, mask, bnc, bn_scale, bn_shift, psum, psq, N, Hi, Wi, Ci, sn, sh,
                         sw, Ho, Wo, Co, R, S, stride_h, stride_w, pad_h, pad_w, relu, cfg, stat_slots, nullptr,
                         nullptr, stream);
}

// pda_conv_igemm with `mask` given as sign bits (one byte per 16-byte vector of the [M][Co] tensor it stands for,
// as written by pda_bn_apply_mask) instead of the tensor itself.
PDA_EXPORT int pda_conv_igemm_maskbits(int dtype, int dgrad, const void* x, const void* w, void* y, const void* res,
                                       const void* mask, const void* bnc, const float* bn_scale, const float* bn_shift,
                                       float* psum, float* psq, int N, int Hi, int Wi, int Ci, long long sn,
                                       long long sh, long long sw, int Ho, int Wo, int Co, int R, int S, int stride_h,
                                       int stride_w, int pad_h, int pad_w, int relu, int cfg, int stat_slots,
                                       hipStream_t stream) {
  return conv_igemm_impl(dtype, dgrad, x, w, y, res, mask, bnc, bn_scale, bn_shift, psum, psq, N, Hi, Wi, Ci, sn, sh,
                         sw, Ho, Wo, Co, R, S, stride_h, stride_w, pad_h, pad_w, relu, cfg, stat_slots, nullptr,
                         nullptr, stream, 1);
}

// pda_conv_igemm with flags: bit 0 = `mask` is given as sign bits (pda_conv_igemm_maskbits), bit 1 = `res` is the
// compact gradient [N][ceil(Ho/2)][ceil(Wo/2)][Co] of a stride-2 1x1 shortcut (stride-1 dgrad only): added at the
// pixels with even row and column.
PDA_EXPORT int pda_conv_igemm_flags(int dtype, int dgrad, const void* x, const void* w, void* y, const void* res,
                                    const void* mask, const void* bnc, const float* bn_scale, const float* bn_shift,
                                    float* psum, float* psq, int N, int Hi, int Wi, int Ci, long long sn,
                                    long long sh, long long sw, int Ho, int Wo, int Co, int R, int S, int stride_h,
                                    int stride_w, int pad_h, int pad_w, int relu, int cfg, int stat_slots, int flags,
                                    hipStream_t stream) {
  return conv_igemm_impl(dtype, dgrad, x, w, y, res, mask, bnc, bn_scale, bn_shift, psum, psq, N, Hi, Wi, Ci, sn, sh,
                         sw, Ho, Wo, Co, R, S, stride_h, stride_w, pad_h, pad_w, relu, cfg, stat_slots, nullptr,
                         nullptr, stream, flags);
}

// acc[M][N] (fp32, zero or holding a value to add to) += x[M][K] * w[N][K]^T with the reduction split `ksplit` ways
// over workgroups (float atomics): a GEMM with few output tiles and a deep K, such as the forward of a Linear
// layer on a small batch.  cfg as for pda_conv_igemm (register-staged tiles only).
PDA_EXPORT int pda_gemm_splitk(int dtype, const void* x, const void* w, float* acc, int M, int K, int N, int cfg,
                               int ksplit, hipStream_t stream) {
  if (ksplit < 1) return -13;
  return conv_igemm_impl(dtype, 0, x, w, acc, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, M, 1, 1, K,
                         K, K, K, 1, 1, N, 1, 1, 1, 1, 0, 0, 0, cfg, 0, nullptr, nullptr, stream, 0, ksplit);
}

// Forward conv whose input is normalised on load: x holds the producing layer's conv output before its
// BatchNorm; in_scale / in_shift are that BatchNorm's per-channel coefficients as Ci values of x's type, and the
// kernel multiplies relu(x * in_scale + in_shift) into the GEMM (exact zeros for padding).  16-bit types and the
// register-staged tiles (cfg 0-3, 9-12) only: <0 otherwise.
PDA_EXPORT int pda_conv_igemm_xf(int dtype, const void* x, const void* w, void* y, const void* in_scale,
                                 const void* in_shift, float* psum, float* psq, int N, int Hi, int Wi, int Ci,
                                 long long sn, long long sh, long long sw, int Ho, int Wo, int Co, int R, int S,
                                 int stride_h, int stride_w, int pad_h, int pad_w, int cfg, int stat_slots,
                                 hipStream_t stream) {
  if (in_scale == nullptr || in_shift == nullptr || dtype == PDA_F32) return -11;
  return conv_igemm_impl(dtype, 0, x, w, y, nullptr, nullptr, nullptr, nullptr, nullptr, psum, psq, N, Hi, Wi, Ci, sn,
                         sh, sw, Ho, Wo, Co, R, S, stride_h, stride_w, pad_h, pad_w, 0, cfg, stat_slots, in_scale,
                         in_shift, stream);
}

template <typename T, int TM, int TN, int BKP, bool RAG = false>
static int launch_wgrad(WgradArgs a, int nsplit_hint, hipStream_t st) {
  const int tiles = ((a.Co + TM - 1) / TM) * ((a.Ktot + TN - 1) / TN);
  int nsplit = nsplit_hint;
  if (nsplit <= 0) {
    // ~2 workgroups per CU: enough to hide the staging latency, few enough that the float-atomic
    // traffic of the split-K reduction (one tile of adds per workgroup) stays small
    // (the 128x128 / 64-pixel configuration holds 2 workgroups per CU: stay within one round of 512)
    if (TM >= 128 && TN >= 128) nsplit = 512 / tiles;
    else nsplit = (704 + tiles / 2) / tiles;
    const int maxsplit = (a.M + 4 * BKP - 1) / (4 * BKP);
    if (nsplit > maxsplit) nsplit = maxsplit;
    if (nsplit < 1) nsplit = 1;
  }
  if (a.det && nsplit > 32) nsplit = 32;  // the slab reduction takes at most 32 splits
  int pps = (a.M + nsplit - 1) / nsplit;
  pps = ((pps + BKP - 1) / BKP) * BKP;
  nsplit = (a.M + pps - 1) / pps;
  a.pix_per_split = pps;
  // split-K reduction: slabs + one streaming reduce when the caller lent a large enough workspace
  // (and dW is 16-byte aligned), float atomics straight into dW otherwise
  const long long n = (long long)a.Co * a.Ktot;
  // (many splits of a small dW - the 64-channel layers at 56x56 - reduce faster through atomics:
  // the reduce kernel would walk hundreds of tiny slabs with a handful of workgroups)
  const bool slabs = nsplit > 1 && nsplit <= 32 && a.slab != nullptr && (long long)nsplit * n * 4 <= a.slab_bytes &&
                     (n % 4) == 0 && (((uintptr_t)a.dw | (uintptr_t)a.slab) & 15) == 0;
  if (!slabs) a.slab = nullptr;
  if (a.det && !slabs && nsplit > 1) {  // no workspace: one split per tile, each dW element gets one add
    nsplit = 1;
    a.pix_per_split = ((a.M + BKP - 1) / BKP) * BKP;
  }
  if constexpr (RAG) {
    hipLaunchKernelGGL((wgrad_kernel<T, TM, TN, BKP, false, true>), dim3(tiles, nsplit), dim3(256), 0, st, a);
  } else if (a.in_scale != nullptr) {
    if constexpr (sizeof(T) == 2) hipLaunchKernelGGL((wgrad_kernel<T, TM, TN, BKP, true>), dim3(tiles, nsplit), dim3(256), 0, st, a);
    else return -11;
  } else {
    hipLaunchKernelGGL((wgrad_kernel<T, TM, TN, BKP>), dim3(tiles, nsplit), dim3(256), 0, st, a);
  }
  if (slabs) {
    const long long n4 = n / 4;
    long long blocks = (n4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((int)blocks), dim3(256), 0, st, a.slab, a.dw, n4, n, nsplit);
  }
  return pda_check_launch();
}

// dw must be zero-initialised (or hold the gradient to accumulate into).
static int conv_wgrad_impl(int dtype, const void* dy, const void* x, float* dw, int N, int Hi, int Wi, int Ci,
                           long long sn, long long sh, long long sw, int Ho, int Wo, int Co, int R, int S,
                           int stride_h, int stride_w, int pad_h, int pad_w, int cfg, int nsplit,
                           float* workspace, long long workspace_bytes, const void* in_scale, const void* in_shift,
                           hipStream_t stream) {
  WgradArgs a;
  a.in_scale = in_scale; a.in_shift = in_shift;
  if ((((uintptr_t)in_scale | (uintptr_t)in_shift) & 15) != 0) return -5;
  a.slab = workspace; a.slab_bytes = workspace_bytes;
  a.dy = dy; a.x = x; a.dw = dw; a.sn = sn; a.sh = sh; a.sw = sw;
  a.Hi = Hi; a.Wi = Wi; a.Ci = Ci; a.Ho = Ho; a.Wo = Wo; a.Co = Co; a.R = R; a.S = S;
  a.stride_h = stride_h; a.stride_w = stride_w; a.pad_h = pad_h; a.pad_w = pad_w;
  a.M = N * Ho * Wo; a.Ktot = R * S * Ci; a.pix_per_split = 0;
  a.det = (cfg & 32) != 0;  // cfg bit 5
  if (a.M <= 0) return 0;
  {
    const long long rl = (long long)pad_h - (long long)(Ho - 1) * stride_h, rh = (long long)Hi - 1 + pad_h;
    const long long sl = (long long)pad_w - (long long)(Wo - 1) * stride_w, sh2 = (long long)Wi - 1 + pad_w;
    a.r_lo = (int)(rl < 0 ? 0 : rl); a.r_hi = (int)(rh > R - 1 ? R - 1 : rh);
    a.s_lo = (int)(sl < 0 ? 0 : sl); a.s_hi = (int)(sh2 > S - 1 ? S - 1 : sh2);
  }
  const int es = dtype == PDA_F32 ? 4 : 2;
  const int epc = 16 / es;
  if (Ci % epc != 0 || Co % epc != 0) return -1;
  if (((uintptr_t)x & 15) || ((uintptr_t)dy & 15)) return -5;
  // cfg: low 3 bits 0 = auto, 1 = 128x128, 2 = 64x64, 3 = 64(Co) x 128, 4 = 256 x 128, 5 = 128 x 256 (4 and 5
  // only with 32-pixel steps); bit 3 = 32 pixels per step
  // instead of 64 (16-bit types).
  int tile = cfg & 7;
  if (tile == 0) {
    // 16-bit: the 128x128 tile also wins when one side is only 64 wide (the 64-channel layers at 56x56,
    // 16-20 % faster than 64x64 there: half as many workgroups re-read the pixel stream); 64x64 only for
    // the 64 -> 64 1x1.  fp32 keeps 128x128 for exact multiples.
    // (and 64 output channels over a deep K - the 3x3 on 64 channels - take the 64(Co) x 128 tile)
    if (es == 2) tile = Co <= 64 ? (a.Ktot <= 64 ? 2 : a.Ktot > 256 ? 3 : 1) : 1;
    else tile = (Co % 128 == 0 && a.Ktot % 128 == 0) ? 1 : 2;
  }
  bool bk32 = (cfg & 8) != 0;
  if ((cfg & 7) == 0 && nsplit <= 0 && es == 2 && tile == 1 && Co % 256 == 0 && a.Ktot % 128 == 0 &&
      (Co / 256) * (a.Ktot / 128) >= 16) {
    // enough output tiles that 256(Co) x 128 ones still split the pixels at most ~32 ways: each wave then
    // owns 128x64 outputs and moves 25 % fewer LDS bytes per MFMA (measured 5-15 % faster on the 3x3 layers
    // from 256 channels up and the wide 1x1 layers; slower when only a handful of tiles exist)
    tile = 4;
    bk32 = true;
  }
  if ((cfg & 7) == 0 && nsplit <= 0 && tile == 1) {
    // many output tiles and few pixels (layer4-style 3x3): the 32-pixel step needs half the LDS, so
    // three workgroups fit per CU and ~720 of them run in one round; the 64-pixel one would need two
    const int tiles = ((Co + 127) / 128) * ((a.Ktot + 127) / 128);
    if (tiles >= 128 && tiles <= 720) {
      bk32 = true;
      nsplit = (720 + tiles / 2) / tiles;
      const int maxsplit = (a.M + 127) / 128;
      if (nsplit > maxsplit) nsplit = maxsplit;
    }
  }
#define PDA_W16(T)                                                                        \
  if (bk32) {                                                                             \
    if (tile == 1) return launch_wgrad<T, 128, 128, 32>(a, nsplit, stream);               \
    if (tile == 3) return launch_wgrad<T, 64, 128, 32>(a, nsplit, stream);                \
    if (tile == 4) return launch_wgrad<T, 256, 128, 32>(a, nsplit, stream);               \
    if (tile == 5) return launch_wgrad<T, 128, 256, 32>(a, nsplit, stream);               \
    return launch_wgrad<T, 64, 64, 32>(a, nsplit, stream);                                \
  }                                                                                       \
  if (tile == 4 || tile == 5) return -2;                                                  \
  if (tile == 1) return launch_wgrad<T, 128, 128, 64>(a, nsplit, stream);                 \
  if (tile == 3) return launch_wgrad<T, 64, 128, 64>(a, nsplit, stream);                  \
  return launch_wgrad<T, 64, 64, 64>(a, nsplit, stream);
  switch (dtype) {
    case PDA_F16: { PDA_W16(f16) }
    case PDA_BF16: { PDA_W16(bf16) }
    case PDA_F32:
      return tile == 1 ? launch_wgrad<float, 128, 128, 16>(a, nsplit, stream)
                       : launch_wgrad<float, 64, 64, 16>(a, nsplit, stream);
    default: return -6;
  }
#undef PDA_W16
}

PDA_EXPORT int pda_conv_wgrad(int dtype, const void* dy, const void* x, float* dw, int N, int Hi, int Wi, int Ci,
                              long long sn, long long sh, long long sw, int Ho, int Wo, int Co, int R, int S,
                              int stride_h, int stride_w, int pad_h, int pad_w, int cfg, int nsplit,
                              float* workspace, long long workspace_bytes, hipStream_t stream) {
  return conv_wgrad_impl(dtype, dy, x, dw, N, Hi, Wi, Ci, sn, sh, sw, Ho, Wo, Co, R, S, stride_h, stride_w, pad_h,
                         pad_w, cfg, nsplit, workspace, workspace_bytes, nullptr, nullptr, stream);
}

// dW [Co][Ci] (fp32, zero-initialised or holding the gradient to accumulate into) += dY^T [Co][M] x X [M][Ci] for
// ANY channel counts: the weight gradient of a pointwise convolution whose channels do not fill 16-byte vectors.
// dy is dense [M][Co]; x has unit channel stride and a pixel pitch of ldx elements (a channel slice of a wider
// tensor qualifies).  det != 0: run-to-run reproducible (one split per tile).
PDA_EXPORT int pda_conv_wgrad_ragged(int dtype, const void* dy, const void* x, float* dw, long long M, int Ci, int Co,
                                     long long ldx, int det, hipStream_t stream) {
  if (M <= 0) return 0;
  if (M > 0x7fffffffLL || Ci <= 0 || Co <= 0 || ldx < Ci) return -1;
  const int es = dtype == PDA_F32 ? 4 : 2;
  if ((((uintptr_t)x | (uintptr_t)dy) & (uintptr_t)(es - 1)) != 0) return -5;
  WgradArgs a;
  a.in_scale = nullptr; a.in_shift = nullptr; a.slab = nullptr; a.slab_bytes = 0;
  a.dy = dy; a.x = x; a.dw = dw; a.sn = 0; a.sh = 0; a.sw = ldx;
  a.Hi = 1; a.Wi = (int)M; a.Ci = Ci; a.Ho = 1; a.Wo = (int)M; a.Co = Co; a.R = 1; a.S = 1;
  a.stride_h = 1; a.stride_w = 1; a.pad_h = 0; a.pad_w = 0;
  a.M = (int)M; a.Ktot = Ci; a.pix_per_split = 0; a.det = det;
  a.r_lo = 0; a.r_hi = 0; a.s_lo = 0; a.s_hi = 0;
  const bool small = Co <= 64 && Ci <= 64;
  switch (dtype) {
    case PDA_F16:
      return small ? launch_wgrad<f16, 64, 64, 32, true>(a, 0, stream) : launch_wgrad<f16, 128, 128, 32, true>(a, 0, stream);
    case PDA_BF16:
      return small ? launch_wgrad<bf16, 64, 64, 32, true>(a, 0, stream) : launch_wgrad<bf16, 128, 128, 32, true>(a, 0, stream);
    case PDA_F32:
      return small ? launch_wgrad<float, 64, 64, 16, true>(a, 0, stream) : launch_wgrad<float, 128, 128, 16, true>(a, 0, stream);
    default: return -6;
  }
}

// Weight gradient against an input that is normalised on load (see pda_conv_igemm_xf): 16-bit types only.
PDA_EXPORT int pda_conv_wgrad_xf(int dtype, const void* dy, const void* x, float* dw, const void* in_scale,
                                 const void* in_shift, int N, int Hi, int Wi, int Ci, long long sn, long long sh,
                                 long long sw, int Ho, int Wo, int Co, int R, int S, int stride_h, int stride_w,
                                 int pad_h, int pad_w, int cfg, int nsplit, float* workspace,
                                 long long workspace_bytes, hipStream_t stream) {
  if (in_scale == nullptr || in_shift == nullptr || dtype == PDA_F32) return -11;
  return conv_wgrad_impl(dtype, dy, x, dw, N, Hi, Wi, Ci, sn, sh, sw, Ho, Wo, Co, R, S, stride_h, stride_w, pad_h,
                         pad_w, cfg, nsplit, workspace, workspace_bytes, in_scale, in_shift, stream);
}

PDA_EXPORT int pda_krsc_to_crsk_batched(int dt_in, int dt_out, const long long* meta, const int* blk, int nblocks,
                                        hipStream_t stream) {
  if (nblocks <= 0) return 0;
  if ((uintptr_t)blk & 15) return -5;
#define PDA_T(TI, TO) \
  hipLaunchKernelGGL((krsc_to_crsk_batched_kernel<TI, TO>), dim3(nblocks), dim3(256), 0, stream, meta, blk)
  if (dt_in == PDA_F32 && dt_out == PDA_F32) PDA_T(float, float);
  else if (dt_in == PDA_F32 && dt_out == PDA_F16) PDA_T(float, f16);
  else if (dt_in == PDA_F32 && dt_out == PDA_BF16) PDA_T(float, bf16);
  else if (dt_in == PDA_F16 && dt_out == PDA_F16) PDA_T(f16, f16);
  else if (dt_in == PDA_BF16 && dt_out == PDA_BF16) PDA_T(bf16, bf16);
  else return -6;
#undef PDA_T
  return pda_check_launch();
}

PDA_EXPORT int pda_krsc_to_crsk(int dt_in, int dt_out, const void* in, void* out, int K, int RS, int C,
                                hipStream_t stream) {
  const long long total = (long long)K * RS * C;
  if (total == 0) return 0;
  const int blocks = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
#define PDA_T(TI, TO) \
  hipLaunchKernelGGL((krsc_to_crsk_kernel<TI, TO>), dim3(blocks), dim3(256), 0, stream, (const TI*)in, (TO*)out, K, RS, C)
  if (dt_in == PDA_F32 && dt_out == PDA_F32) PDA_T(float, float);
  else if (dt_in == PDA_F32 && dt_out == PDA_F16) PDA_T(float, f16);
  else if (dt_in == PDA_F32 && dt_out == PDA_BF16) PDA_T(float, bf16);
  else if (dt_in == PDA_F16 && dt_out == PDA_F16) PDA_T(f16, f16);
  else if (dt_in == PDA_BF16 && dt_out == PDA_BF16) PDA_T(bf16, bf16);
  else return -6;
#undef PDA_T
  return pda_check_launch();
}
