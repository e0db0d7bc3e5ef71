// Forward of a pointwise (1x1, unpadded) convolution with a shallow reduction, Y[M][N] = X[M][K] * W[N][K]^T for
// K = 64 / 128 / 256 / 512 input channels, fp16 / bf16 on gfx950: the w -> 4w convolution that ends every ResNet
// bottleneck (and the 1x1 shortcut convolutions).  These layers move 5 bytes of output per byte of input and have
// one to eight K steps per output tile, so an output-tile-per-workgroup kernel spends its life in the prologue
// (operand latency) and the write-out of tiles that compute for a fraction of a microsecond.  Here the A operand is
// stationary and the rest is one software pipeline:
//   * a wave owns 64 (32 for K = 512) pixel rows and loads ALL their K values once, straight from memory into the
//     registers in MFMA operand layout (a lane = one pixel row, 16 bytes per 16-deep group).  When the input is a
//     conv output before its BatchNorm ("normalise on load") relu(x * scale + shift) is applied to these registers
//     once - eight channels per fragment, coefficients loaded once per wave.
//   * the workgroup then walks over its output-channel range in 64-column tiles.  The weights (L2 resident, read by
//     every workgroup) stream through a four-deep LDS ring by LDS-DMA in 64-column x 64-deep steps under a counted
//     vmcnt and one barrier per step; there is no A traffic through LDS at all, so LDS reads per MFMA are half
//     those of a tiled kernel.
//   * a finished 64 x 64 tile of a wave goes through a wave-private LDS staging area (the MFMA runs with swapped
//     operands, so a lane holds four consecutive channels of one pixel: 8-byte LDS writes), is read back as 16-byte
//     row vectors, stored with whole 128-byte lines per row and summed into the BatchNorm statistics on the way.
//     The stores of tile t drain while tile t + 1 computes: the ring's waits allow for them by counting every
//     younger request, loads and stores alike (vector memory operations complete in order).
// LDS writes of the epilogue are bare instructions: a compiler-visible LDS store after an LDS-DMA request makes the
// wait-count pass drain the DMA ring first.
// (Reference trigger: `model(images)` on torchvision's Bottleneck, /root/reference/2.distributed.py:213.)
#include <type_traits>
#include <utility>
#include "common.h"

namespace {

struct PWArgs {
  const void* x;
  const void* w;         // [N][K]
  void* y;               // [M][N]
  const void* in_scale;  // XF: K values of x's type
  const void* in_shift;
  float* psum;           // optional statistics: [rows][N] partial column sums and sums of squares
  float* psq;
  int M, K, N;
  int Ho, Wo;
  long long sn, sh, sw;  // element strides of x per image / output row / output column (conv stride folded in)
  long long x_bytes;
  int lin;               // x is a dense [M][K] matrix
  int nch, nx, ny;       // output channels per workgroup, row tiles, channel chunks
  int stat_slots;
  int stream_out;
  // MAT kernels: x is the conv output y3 of the previous residual block's last layer (dense [M][K]) and this kernel
  // finishes that block on the way in: out = relu(y3 * mat_scale + mat_shift + mat_res [* mat_res_scale]) is computed
  // in fp32 on the A rows, rounded, stored to mat_out ([M][K]) with its sign bits (mat_mask: one byte per 16-byte
  // vector) and used as the operand - the block-end pass and this conv's read of its result become one
  const void* mat_res;
  const float* mat_scale;
  const float* mat_shift;
  const float* mat_res_scale;   // (ones for an identity shortcut)
  void* mat_out;
  unsigned char* mat_mask;
};

template <typename T> struct MfmaPW;
template <> struct MfmaPW<f16> {
  typedef f16x8 frag;
  static __device__ __forceinline__ f32x16 run(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};
template <> struct MfmaPW<bf16> {
  typedef bf16x8 frag;
  static __device__ __forceinline__ f32x16 run(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};

// relu(v * sc + sf), rounded like the normalise-on-load of the tiled kernels (conv_igemm.hip, affine_relu8)
__device__ __forceinline__ f16x8 pw_affine_relu(f16x8 v, f16x8 sc, f16x8 sf) {
  const f16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
  return __builtin_elementwise_max(__builtin_elementwise_fma(v, sc, sf), zero);
}
__device__ __forceinline__ bf16x8 pw_affine_relu(bf16x8 v, bf16x8 sc, bf16x8 sf) {
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (bf16)fmaxf(fmaf((float)v[e], (float)sc[e], (float)sf[e]), 0.f);
  return o;
}

// s_waitcnt immediate (gfx9 layout): vmcnt in bits 3:0 and 15:14, expcnt 6:4 (7 = no wait), lgkmcnt 11:8 (15 = no wait)
constexpr int pw_waitcnt_vm(int vm) { return (vm & 15) | ((vm >> 4) << 14) | (7 << 4) | (15 << 8); }
// ... and lgkmcnt(0): the form in front of the ring barrier.  The compiler issues a step's last fragment reads ahead of
// the barrier and consumes them behind it; the request that follows the barrier overwrites the very stage those reads
// address, so they must have RETURNED (not merely been issued) before any wave passes the barrier.  (Without it about
// one wave tile in two thousand multiplied a stale 8-row weight piece when a neighbouring workgroup kept the LDS busy.)
constexpr int pw_waitcnt_vm_lds(int vm) { return (vm & 15) | ((vm >> 4) << 14) | (7 << 4) | (0 << 8); }

template <int OFF>
__device__ __forceinline__ void lds_write_b64(unsigned addr, u32x2 v) {
  asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_write_b128(unsigned addr, u32x4 v) {
  asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}

template <int... Is, typename F>
__device__ __forceinline__ void static_for(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}

template <typename T> __device__ __forceinline__ unsigned pack2(float a, float b) {
  const T x = (T)a, y = (T)b;
  return (unsigned)__builtin_bit_cast(unsigned short, x) | ((unsigned)__builtin_bit_cast(unsigned short, y) << 16);
}

// KS: K / 64.  MI: 32-row blocks per wave (2, or 1 when the K = 512 fragments of two blocks would not fit).
// MODE: 0 = x as it is, 1 (XF) = relu(x * in_scale + in_shift), 2 (MAT) = see PWArgs::mat_res.
template <typename T, int KS, int MI, int MODE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2)))
pw_fprop_kernel(PWArgs a) {
  constexpr bool XF = MODE == 1;
  constexpr int KK = KS * 4;               // 16-deep MFMA groups along K
  constexpr int NI = 2;                    // a tile is 64 output channels
  constexpr int STAGES = 4, STEP_BYTES = 64 * 128, PIECES = 2;   // ring step: 64 weight rows x 64 k
  constexpr int WROWS = MI * 32, BM = 4 * WROWS;
  constexpr int SP = 144;                  // staging pitch: 64 channels + 16 bytes
  constexpr int STG = WROWS * SP;          // per wave; reused for the statistics fold ([2][8][64] floats)
  static_assert(STG >= 2 * 8 * 64 * 4, "the statistics fold fits the staging area");
  constexpr int NCH_MAX = 256;
  constexpr int RING = STAGES * STEP_BYTES;
  constexpr int STORES = MI * 4;           // global stores of one tile per wave
  constexpr unsigned OOB = 0x80000000u;
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ __attribute__((aligned(16))) unsigned char lds[RING + 4 * STG + 4 * NCH_MAX * 2 * 4];
  typedef __attribute__((address_space(3))) void* lds_ptr;
  typedef typename MfmaPW<T>::frag frag;
  const int t = threadIdx.x, lane = t & 63;
  const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int L = xcd_remap(blockIdx.x, a.nx * a.ny);
  const int mx = L / a.ny, my = L - mx * a.ny;
  const int m0 = mx * BM + wid * WROWS;   // first pixel row of this wave
  const int n_begin = my * a.nch;
  const int ncols = min(a.nch, a.N - n_begin);
  const int ntile = (ncols + 63) >> 6;
  const int K = KS * 64;
  const unsigned lds_base = (unsigned)(uintptr_t)(lds_ptr)lds;

  // ---- A: every K value of this wave's rows, into MFMA operand layout (lane = row l31, k = 8 * lh .. + 7 of a group)
  const auto rsrcX = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, (int)a.x_bytes, 0x00020000);
  frag afr[MI][KK];
  bool rok[MI];
  unsigned vos[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int r = m0 + i * 32 + l31;
    rok[i] = r < a.M;
    long long off;
    if (a.lin) {
      off = (long long)r * K;
    } else {
      const int hw = a.Ho * a.Wo;
      const int n = r / hw, rem = r - n * hw;
      const int ho = rem / a.Wo, wo = rem - ho * a.Wo;
      off = n * a.sn + ho * a.sh + wo * a.sw;
    }
    const unsigned vo = rok[i] ? (unsigned)(off * 2 + lh * 16) : OOB;
    vos[i] = vo;
    if constexpr (MODE != 2) {
#pragma unroll
      for (int kk = 0; kk < KK; ++kk)
        afr[i][kk] = __builtin_bit_cast(frag, __builtin_amdgcn_raw_buffer_load_b128(rsrcX, vo + kk * 32, 0, 0));
    }
  }

  // ---- weight ring: a wave instruction fills 1 KiB = 8 rows of a step's [64 rows][64 k] image; 16-byte chunks are
  // XOR-swizzled by (row / 2) & 7 (conflict-free 16-byte fragment reads), applied on the source side
  const auto rsrcW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, (int)((long long)a.N * K * 2), 0x00020000);
  unsigned wv[PIECES];
  {
    const int r8 = lane >> 3, slot = lane & 7;
    const int chunk = slot ^ ((4 * (wid & 1) + (r8 >> 1)) & 7);   // (row / 2) & 7 with row = (wid + 4 p) * 8 + r8
#pragma unroll
    for (int p = 0; p < PIECES; ++p) {
      const int row = (wid + 4 * p) * 8 + r8;
      wv[p] = (unsigned)((((long long)n_begin + row) * K + chunk * 8) * 2);
    }
  }
  int is_tile = 0, is_ks = 0;   // the next step to request
  auto issue = [&](int stage) {
    const unsigned add = (unsigned)is_tile * (unsigned)(64 * K * 2) + (unsigned)is_ks * 128u;
    unsigned char* dst = lds + stage * STEP_BYTES + wid * 1024;
#pragma unroll
    for (int p = 0; p < PIECES; ++p)   // (steps past the last tile read other channels or, past N, zeros: never used)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, (lds_ptr)(dst + p * 4096), 16, wv[p] + add, 0, 0, 0);
    if (++is_ks == KS) { is_ks = 0; ++is_tile; }
  };
  issue(0);
  issue(1);
  issue(2);

  if constexpr (XF) {
    const auto rsrcS = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in_scale), 0, K * 2, 0x00020000);
    const auto rsrcF = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.in_shift), 0, K * 2, 0x00020000);
    const frag zero = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      const frag sc = __builtin_bit_cast(frag, __builtin_amdgcn_raw_buffer_load_b128(rsrcS, (kk * 16 + lh * 8) * 2, 0, 0));
      const frag sf = __builtin_bit_cast(frag, __builtin_amdgcn_raw_buffer_load_b128(rsrcF, (kk * 16 + lh * 8) * 2, 0, 0));
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const frag v = pw_affine_relu(afr[i][kk], sc, sf);
        afr[i][kk] = rok[i] ? v : zero;   // rows past M stay exact zeros (statistics)
      }
    }
  }

  if constexpr (MODE == 2) {
    // The previous block's last pass: BatchNorm + shortcut + ReLU in fp32 (as bn_apply_kernel), 64 channels at a
    // time, with the lanes in row-vector roles (lane -> 16-byte vector lane & 7 of rows q * 8 + lane / 8): whole
    // 128-byte lines per row are read from y3 and the shortcut and written to `out`, one sign byte per vector goes to
    // the mask, and the rounded values pass through this wave's staging area into MFMA operand layout (the lane = row
    // layout of the direct A loads would read and write 16 bytes per row and instruction: measured 20 % slower than
    // the two separate kernels).  Rows past M stay exact zeros (statistics).
    const auto rsrcR = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.mat_res), 0, (int)a.x_bytes, 0x00020000);
    const auto rsrcO = __builtin_amdgcn_make_buffer_rsrc(a.mat_out, 0, (int)a.x_bytes, 0x00020000);
    const auto rsrcM = __builtin_amdgcn_make_buffer_rsrc(a.mat_mask, 0, (int)(a.x_bytes / 16), 0x00020000);
    const int cv = lane & 7, rg = lane >> 3;
    const unsigned stg_w = lds_base + RING + wid * STG;
    const unsigned char* stg_r = lds + RING + wid * STG;
    static_for(std::make_integer_sequence<int, KS>{}, [&](auto ks_) {
      constexpr int ks = decltype(ks_)::value;
      const int c0 = ks * 64 + cv * 8;
      const f32x4 sc0 = *reinterpret_cast<const f32x4*>(a.mat_scale + c0), sc1 = *reinterpret_cast<const f32x4*>(a.mat_scale + c0 + 4);
      const f32x4 sf0 = *reinterpret_cast<const f32x4*>(a.mat_shift + c0), sf1 = *reinterpret_cast<const f32x4*>(a.mat_shift + c0 + 4);
      // (always a table - ones for a plain identity shortcut: a run-time branch inside this unrolled code spills)
      const f32x4 rs0 = *reinterpret_cast<const f32x4*>(a.mat_res_scale + c0), rs1 = *reinterpret_cast<const f32x4*>(a.mat_res_scale + c0 + 4);
      frag yv[WROWS / 8], rv[WROWS / 8];
      unsigned go[WROWS / 8];
#pragma unroll
      for (int q = 0; q < WROWS / 8; ++q) {
        const int m = m0 + q * 8 + rg;
        go[q] = m < a.M ? (unsigned)(((long long)m * K + c0) * 2) : OOB;
        yv[q] = __builtin_bit_cast(frag, __builtin_amdgcn_raw_buffer_load_b128(rsrcX, go[q], 0, 0));
        rv[q] = __builtin_bit_cast(frag, __builtin_amdgcn_raw_buffer_load_b128(rsrcR, go[q], 0, 0));
      }
#pragma unroll
      for (int q = 0; q < WROWS / 8; ++q) {
        frag o;
        unsigned bits = 0u;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float scv = e < 4 ? sc0[e & 3] : sc1[e & 3], sfv = e < 4 ? sf0[e & 3] : sf1[e & 3];
          const float rsv = e < 4 ? rs0[e & 3] : rs1[e & 3];
          const float f = fmaxf((float)yv[q][e] * scv + sfv + (float)rv[q][e] * rsv, 0.f);
          o[e] = (T)f;
          bits |= ((float)o[e] > 0.f ? 1u : 0u) << e;
        }
        if (go[q] == OOB) o = frag{};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rsrcO, go[q], 0, 0);
        __builtin_amdgcn_raw_buffer_store_b8((unsigned char)bits, rsrcM, go[q] == OOB ? OOB : go[q] >> 4, 0, 0);
        lds_write_b128<0>(stg_w + (q * 8 + rg) * SP + cv * 16, __builtin_bit_cast(u32x4, o));
      }
      // back in operand layout (LDS operations of a wave run in order: the reads see the writes)
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
          afr[i][ks * 4 + kk] = *reinterpret_cast<const frag*>(stg_r + (i * 32 + l31) * SP + (kk * 2 + lh) * 16);
      // (one 64-channel slice at a time: the staging area is reused, and hoisting every slice's requests spills)
      __builtin_amdgcn_s_waitcnt(15 | (3 << 14) | (7 << 4) | (0 << 8));   // lgkmcnt(0): the reads above, before the next writes
      __builtin_amdgcn_sched_barrier(0);
    });
  }

  // fragment addresses of the weight tile: row = output channel j * 32 + l31, chunk (kk * 2 + lh) ^ swizzle
  unsigned offW[NI][4];
#pragma unroll
  for (int j = 0; j < NI; ++j)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int row = j * 32 + l31;
      offW[j][kk] = row * 128 + (((kk * 2 + lh) ^ ((row >> 1) & 7)) << 4);
    }

  const unsigned stg = lds_base + RING + wid * STG;                       // this wave's staging area
  const unsigned char* stgp = lds + RING + wid * STG;
  const unsigned stat_w = lds_base + RING + 4 * STG + wid * (NCH_MAX * 8);   // this wave's [nch][2] column sums
  T* __restrict__ Y = reinterpret_cast<T*>(a.y);
  const bool stats = a.psum != nullptr;

  const bool ragged = (mx + 1) * BM > a.M;
  int s = 0;   // ring step
  for (int tile = 0; tile < ntile; ++tile) {
    f32x16 acc[NI][MI];   // swapped operands: acc[j][i][g] = channel j*32 + 8*(g>>2) + 4*lh + (g&3), pixel i*32 + l31
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int g = 0; g < 16; ++g) acc[j][i][g] = 0.f;

    static_for(std::make_integer_sequence<int, KS>{}, [&](auto ks_) {
      constexpr int ks = decltype(ks_)::value;
      // Requests younger than this step's weights: the two ring steps behind it, and the stores of every tile that
      // was written out during the last three steps (tile ends at steps = KS - 1 mod KS), as far as tiles exist.
      constexpr int dsel = (ks + 1) % KS;
      constexpr int e_static = ((1 % KS) == dsel) + ((2 % KS) == dsel) + ((3 % KS) == dsel);
      // (a workgroup on the ragged end of M skips whole store instructions: it waits for everything instead)
      const int e = ragged ? 0 : min(e_static, tile);
      if (e == 0) __builtin_amdgcn_s_waitcnt(pw_waitcnt_vm_lds((STAGES - 2) * PIECES));
      else if (e == 1) __builtin_amdgcn_s_waitcnt(pw_waitcnt_vm_lds((STAGES - 2) * PIECES + STORES));
      else if (e == 2) __builtin_amdgcn_s_waitcnt(pw_waitcnt_vm_lds((STAGES - 2) * PIECES + 2 * STORES));
      else __builtin_amdgcn_s_waitcnt(pw_waitcnt_vm_lds((STAGES - 2) * PIECES + 3 * STORES));
      __builtin_amdgcn_s_barrier();   // step s is complete, and everybody's reads of step s - 1 have returned
      issue((s + STAGES - 1) & (STAGES - 1));
      const unsigned char* sb = lds + (s & (STAGES - 1)) * STEP_BYTES;
      frag wf[2][NI];
#pragma unroll
      for (int j = 0; j < NI; ++j) wf[0][j] = *reinterpret_cast<const frag*>(sb + offW[j][0]);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        if (kk + 1 < 4) {
#pragma unroll
          for (int j = 0; j < NI; ++j) wf[(kk + 1) & 1][j] = *reinterpret_cast<const frag*>(sb + offW[j][kk + 1]);
        }
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int i = 0; i < MI; ++i) acc[j][i] = MfmaPW<T>::run(wf[kk & 1][j], afr[i][ks * 4 + kk], acc[j][i]);
      }
      ++s;
    });

    // ---- write-out of this wave's WROWS x 64 tile
    // 1. to the staging area as [pixel][channel]: a lane holds four consecutive channels of its pixel per (j, g / 4)
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const unsigned rowa = stg + (i * 32 + l31) * SP + lh * 8;
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const u32x2 q0 = {pack2<T>(acc[j][i][0], acc[j][i][1]), pack2<T>(acc[j][i][2], acc[j][i][3])};
        const u32x2 q1 = {pack2<T>(acc[j][i][4], acc[j][i][5]), pack2<T>(acc[j][i][6], acc[j][i][7])};
        const u32x2 q2 = {pack2<T>(acc[j][i][8], acc[j][i][9]), pack2<T>(acc[j][i][10], acc[j][i][11])};
        const u32x2 q3 = {pack2<T>(acc[j][i][12], acc[j][i][13]), pack2<T>(acc[j][i][14], acc[j][i][15])};
        if (j == 0) {
          lds_write_b64<0>(rowa, q0); lds_write_b64<16>(rowa, q1); lds_write_b64<32>(rowa, q2); lds_write_b64<48>(rowa, q3);
        } else {
          lds_write_b64<64>(rowa, q0); lds_write_b64<80>(rowa, q1); lds_write_b64<96>(rowa, q2); lds_write_b64<112>(rowa, q3);
        }
      }
    }
    // 2. back as row vectors (LDS operations of a wave run in order, so the reads see the writes): lane -> channel
    // group lane & 7 of rows q * 8 + lane / 8; whole 128-byte lines per row go to memory
    const int cv = lane & 7, rg = lane >> 3;
    const int n = n_begin + tile * 64 + cv * 8;
    float s1[8], s2[8];
#pragma unroll
    for (int e2 = 0; e2 < 8; ++e2) { s1[e2] = 0.f; s2[e2] = 0.f; }
#pragma unroll
    for (int q = 0; q < STORES; ++q) {
      const int row = q * 8 + rg;
      const Vec16<T> v = *reinterpret_cast<const Vec16<T>*>(stgp + row * SP + cv * 16);
      const int m = m0 + row;
      if (m < a.M && n < a.N) store16_maybe_stream(Y + (long long)m * a.N + n, v, a.stream_out);
      if (stats) {
#pragma unroll
        for (int e2 = 0; e2 < 8; ++e2) {
          const float f = ElemTraits<T>::to_f32(v.v[e2]);   // statistics of the values as stored
          s1[e2] += f;
          s2[e2] += f * f;
        }
      }
    }
    if (stats) {
      // 3. fold the eight row groups: partial sums through the (now free) staging area, lane c sums column c
      const unsigned pa = stg + (rg * 64 + cv * 8) * 4;
      lds_write_b128<0>(pa, u32x4{__float_as_uint(s1[0]), __float_as_uint(s1[1]), __float_as_uint(s1[2]), __float_as_uint(s1[3])});
      lds_write_b128<16>(pa, u32x4{__float_as_uint(s1[4]), __float_as_uint(s1[5]), __float_as_uint(s1[6]), __float_as_uint(s1[7])});
      lds_write_b128<2048>(pa, u32x4{__float_as_uint(s2[0]), __float_as_uint(s2[1]), __float_as_uint(s2[2]), __float_as_uint(s2[3])});
      lds_write_b128<2064>(pa, u32x4{__float_as_uint(s2[4]), __float_as_uint(s2[5]), __float_as_uint(s2[6]), __float_as_uint(s2[7])});
      const float* pf = reinterpret_cast<const float*>(stgp);
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        t1 += pf[g * 64 + lane];
        t2 += pf[512 + g * 64 + lane];
      }
      lds_write_b64<0>(stat_w + (tile * 64 + lane) * 8, u32x2{__float_as_uint(t1), __float_as_uint(t2)});
    }
  }

  __builtin_amdgcn_s_waitcnt(pw_waitcnt_vm(0));   // the ring requests past the last tile must land before the LDS is released
  if (stats) {
    __syncthreads();   // (also waits for the bare LDS writes: it drains lgkmcnt)
    const float* sf = reinterpret_cast<const float*>(lds + RING + 4 * STG);
    for (int c = t; c < ntile * 64; c += 256) {
      if (n_begin + c >= a.N) continue;
      float v1 = 0.f, v2 = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        v1 += sf[(w * NCH_MAX + c) * 2];
        v2 += sf[(w * NCH_MAX + c) * 2 + 1];
      }
      if (a.stat_slots > 0) {
        const long long o = (long long)(mx % a.stat_slots) * a.N + n_begin + c;
        atomicAdd(a.psum + o, v1);
        atomicAdd(a.psq + o, v2);
      } else {
        a.psum[(long long)mx * a.N + n_begin + c] = v1;
        a.psq[(long long)mx * a.N + n_begin + c] = v2;
      }
    }
  }
#endif
}

template <typename T, int KS, int MI>
int launch_pw(PWArgs a, hipStream_t st) {
  constexpr int BM = 4 * MI * 32;
  a.nx = (a.M + BM - 1) / BM;
  int nch = ((a.N + 63) / 64) * 64;
  if (nch > 256) nch = 256;
  // (the widest range the statistics scratch allows measured best on every ResNet-50 layer, down to 154 row tiles x 8
  // chunks: every extra chunk loads - and normalises - the A rows once more)
  static const int forced = [] { const char* e = getenv("PDA_PW_NCH"); return e ? atoi(e) : 0; }();   // experiments
  if (forced >= 64 && forced <= 256 && forced % 64 == 0) nch = forced;
  a.nch = nch;
  a.ny = (a.N + nch - 1) / nch;
  const long long nwg = (long long)a.nx * a.ny;
  if (nwg > 0x7fffffffLL) return -4;
  if (a.mat_res != nullptr) {
    if constexpr (KS == 4 || KS == 8) {
      if (a.ny != 1) return -1;   // one workgroup per row block writes the materialised rows
      hipLaunchKernelGGL((pw_fprop_kernel<T, KS, MI, 2>), dim3((unsigned)nwg), dim3(256), 0, st, a);
    } else {
      return -1;
    }
  } else if (a.in_scale != nullptr)
    hipLaunchKernelGGL((pw_fprop_kernel<T, KS, MI, 1>), dim3((unsigned)nwg), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((pw_fprop_kernel<T, KS, MI, 0>), dim3((unsigned)nwg), dim3(256), 0, st, a);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

}  // namespace

// Pixel rows per statistics row of pda_conv_pw_fprop for this input channel count (0: not covered).
PDA_EXPORT int pda_conv_pw_bm(int Ci) { return Ci == 64 || Ci == 128 || Ci == 256 ? 256 : (Ci == 512 ? 128 : 0); }

// y[N,Ho,Wo,Co] = conv1x1(x[N,Hi,Wi,Ci], w[Co][Ci]) with stride, no padding; optionally relu(x * in_scale + in_shift)
// in place of x (in_scale / in_shift: Ci values of x's type) and the per-row-tile BatchNorm statistics
// (psum / psq: [ceil(M / pda_conv_pw_bm(Ci))][Co], or stat_slots > 0 pre-zeroed rows accumulated with atomics).
// <0: not covered (fp32, other channel counts, tensors beyond the 32-bit offset range) - use pda_conv_igemm.
PDA_EXPORT int pda_conv_pw_fprop(int dtype, const void* x, const void* w, void* y, const void* in_scale,
                                 const void* in_shift, float* psum, float* psq, int N, int Hi, int Wi, int Ci,
                                 long long sn, long long sh, long long sw, int Ho, int Wo, int Co, int stride_h,
                                 int stride_w, int stat_slots, hipStream_t stream) {
  if (dtype != PDA_F16 && dtype != PDA_BF16) return -6;
  if (pda_conv_pw_bm(Ci) == 0 || Co % 8 != 0 || Co <= 0) return -1;
  if ((((uintptr_t)x | (uintptr_t)w | (uintptr_t)y | (uintptr_t)in_scale | (uintptr_t)in_shift) & 15) != 0) return -5;
  if ((in_scale == nullptr) != (in_shift == nullptr)) return -11;
  if ((psum == nullptr) != (psq == nullptr)) return -11;
  if ((long long)(Ho - 1) * stride_h >= Hi || (long long)(Wo - 1) * stride_w >= Wi) return -3;
  const long long M = (long long)N * Ho * Wo;
  if (M <= 0) return 0;
  const long long xb = ((long long)(N - 1) * sn + (long long)(Hi - 1) * sh + (long long)(Wi - 1) * sw + Ci) * 2;
  if (M > 0x7fffffffLL || xb >= 0x7fffffffLL || (long long)Co * Ci * 2 >= 0x7fffffffLL) return -10;
  PWArgs a;
  a.x = x; a.w = w; a.y = y; a.in_scale = in_scale; a.in_shift = in_shift; a.psum = psum; a.psq = psq;
  a.M = (int)M; a.K = Ci; a.N = Co; a.Ho = Ho; a.Wo = Wo;
  a.sn = sn; a.sh = sh * stride_h; a.sw = sw * stride_w;
  a.x_bytes = xb;
  a.lin = stride_h == 1 && stride_w == 1 && sw == Ci && sh == (long long)Wi * Ci && sn == (long long)Hi * Wi * Ci &&
          Ho == Hi && Wo == Wi;
  a.stat_slots = stat_slots;
  a.stream_out = M * Co * 2 >= pda_stream_min_bytes();
  a.nch = a.nx = a.ny = 0;
  a.mat_res = nullptr; a.mat_scale = nullptr; a.mat_shift = nullptr; a.mat_res_scale = nullptr; a.mat_out = nullptr;
  a.mat_mask = nullptr;
#define PDA_PW(T)                                          \
  switch (Ci) {                                            \
    case 64: return launch_pw<T, 1, 2>(a, stream);         \
    case 128: return launch_pw<T, 2, 2>(a, stream);        \
    case 256: return launch_pw<T, 4, 2>(a, stream);        \
    default: return launch_pw<T, 8, 1>(a, stream);         \
  }
  if (dtype == PDA_F16) { PDA_PW(f16) }
  PDA_PW(bf16)
#undef PDA_PW
}

// The previous residual block's last pass and this block's first 1x1 convolution in one kernel (see PWArgs::mat_res):
// out[M][Ci] = relu(y3 * scale + shift + res * res_scale) with its sign bits (mask: M * Ci / 8 bytes), and
// y[M][Co] = out * w[Co][Ci]^T with the optional statistics of pda_conv_pw_fprop.  Ci = 256 or 512, Co <= 256, all
// tensors dense [M][.].  <0: not covered.
PDA_EXPORT int pda_conv_pw_fprop_mat(int dtype, const void* y3, const void* res, const float* scale, const float* shift,
                                     const float* res_scale, void* out, unsigned char* mask, const void* w, void* y,
                                     float* psum, float* psq, long long M, int Ci, int Co, int stat_slots,
                                     hipStream_t stream) {
  if (dtype != PDA_F16 && dtype != PDA_BF16) return -6;
  if ((Ci != 256 && Ci != 512) || Co % 8 != 0 || Co <= 0 || Co > 256) return -1;
  if (!y3 || !res || !scale || !shift || !res_scale || !out || !mask || !w || !y) return -11;
  if ((((uintptr_t)y3 | (uintptr_t)res | (uintptr_t)out | (uintptr_t)mask | (uintptr_t)w | (uintptr_t)y |
        (uintptr_t)scale | (uintptr_t)shift | (uintptr_t)res_scale) & 15) != 0)
    return -5;
  if ((psum == nullptr) != (psq == nullptr)) return -11;
  if (M <= 0) return 0;
  const long long xb = M * Ci * 2;
  if (M > 0x7fffffffLL || xb >= 0x7fffffffLL) return -10;
  PWArgs a;
  a.x = y3; a.w = w; a.y = y; a.in_scale = nullptr; a.in_shift = nullptr; a.psum = psum; a.psq = psq;
  a.M = (int)M; a.K = Ci; a.N = Co; a.Ho = 1; a.Wo = 1;
  a.sn = a.sh = a.sw = Ci;
  a.x_bytes = xb;
  a.lin = 1;
  a.stat_slots = stat_slots;
  a.stream_out = M * Co * 2 >= pda_stream_min_bytes();
  a.nch = a.nx = a.ny = 0;
  a.mat_res = res; a.mat_scale = scale; a.mat_shift = shift; a.mat_res_scale = res_scale; a.mat_out = out; a.mat_mask = mask;
  if (dtype == PDA_F16) return Ci == 256 ? launch_pw<f16, 4, 2>(a, stream) : launch_pw<f16, 8, 1>(a, stream);
  return Ci == 256 ? launch_pw<bf16, 4, 2>(a, stream) : launch_pw<bf16, 8, 1>(a, stream);
}
