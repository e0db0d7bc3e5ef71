// BatchNorm (training + eval) for NHWC activations viewed as [M][C], with the
// ReLU and residual-add of a ResNet block fused into the same passes
// (reference trigger: torchvision's BatchNorm2d/ReLU inside
// `model(images)`, /root/reference/2.distributed.py:213).
//
// Forward (train):  statistics are per-tile partial sums (written either by the
// conv epilogue or by bn_stats_partial_kernel), reduced in fp64 by
// bn_finalize_kernel, which also updates the running statistics and emits the
// fused per-channel scale/shift; bn_apply_kernel then does
//     out = relu(y * scale[c] + shift[c] + residual)
// in one 16-byte-vectorised pass.  Cross-rank SyncBN slots an all-reduce of the
// [2][C] sums between bn_reduce_partials_kernel and bn_finalize_kernel.
// Backward: bn_bwd_reduce_kernel produces partials of dbeta = sum(g) and
// dgamma = sum(g * xhat) with the ReLU mask applied on the fly, and
// bn_bwd_apply_kernel writes dy = gamma*invstd*(g - dbeta/M - xhat*dgamma/M)
// (and the masked gradient for the residual branch).
#include "common.h"

// VB = bytes one lane moves per access in the element-wise and column-reduction kernels below: 16 for every channel
// count that fills whole 16-byte vectors (the instantiations all ResNet-style widths use), 4 for the others with an
// even 16-bit (or any fp32) count - the 58 / 116 channels per branch of ShuffleNetV2.
template <typename T, int VB> using VecB = VecN<T, VB / (int)sizeof(T)>;
template <typename T, int VB>
__device__ __forceinline__ VecB<T, VB> ldv(const T* p) {
  VecB<T, VB> r;
  if constexpr (VB == 16) *reinterpret_cast<u32x4*>(&r) = *reinterpret_cast<const u32x4*>(p);
  else *reinterpret_cast<unsigned*>(&r) = *reinterpret_cast<const unsigned*>(p);
  return r;
}
template <typename T, int VB>
__device__ __forceinline__ VecB<T, VB> ldv_stream(const T* p) {
  VecB<T, VB> r;
  if constexpr (VB == 16) *reinterpret_cast<u32x4*>(&r) = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
  else *reinterpret_cast<unsigned*>(&r) = *reinterpret_cast<const unsigned*>(p);
  return r;
}
template <typename T, int VB>
__device__ __forceinline__ void stv_maybe_stream(T* p, const VecB<T, VB>& v, int stream) {
  if constexpr (VB == 16) {
    if (stream) __builtin_nontemporal_store(*reinterpret_cast<const u32x4*>(&v), reinterpret_cast<u32x4*>(p));
    else *reinterpret_cast<u32x4*>(p) = *reinterpret_cast<const u32x4*>(&v);
  } else {
    *reinterpret_cast<unsigned*>(p) = *reinterpret_cast<const unsigned*>(&v);
  }
}

// Column-reduction geometry shared by the reduce kernels: a 256-thread block
// covers VT = min(C/EPC, 256) channel vectors x RS = 256/VT rows per pass.
template <typename T, int VB = 16> struct ColGeom {
  static constexpr int EPC = VB / sizeof(T);
  int VT, RS, vec, rl;
  __device__ ColGeom(int C, int cblk) {
    const int vpr = C / EPC;
    VT = vpr < 256 ? vpr : 256;
    // round VT down to a power of two so RS divides 256 (C/EPC is a power of two
    // for every ResNet width; other widths still work, just with idle lanes)
    int p = 1;
    while (p * 2 <= VT) p *= 2;
    VT = p;
    RS = 256 / VT;
    vec = cblk * VT + (threadIdx.x % VT);
    rl = threadIdx.x / VT;
  }
};

template <typename T, int VB = 16>
__global__ void __launch_bounds__(256)
bn_stats_partial_kernel(const T* __restrict__ y, long long M, int C, float* __restrict__ psum,
                        float* __restrict__ psq) {
  constexpr int EPC = VB / sizeof(T);
  __shared__ float red[2][256 * EPC];
  ColGeom<T, VB> g(C, blockIdx.y);
  const int vpr = C / EPC;
  float s[EPC], q[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) { s[e] = 0.f; q[e] = 0.f; }
  if (g.vec < vpr) {
    const long long rows_per_blk = (M + gridDim.x - 1) / gridDim.x;
    const long long r0 = blockIdx.x * rows_per_blk;
    const long long r1 = r0 + rows_per_blk < M ? r0 + rows_per_blk : M;
    for (long long r = r0 + g.rl; r < r1; r += g.RS) {
      VecB<T, VB> v = ldv<T, VB>(y + r * C + (long long)g.vec * EPC);
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        const float f = ElemTraits<T>::to_f32(v.v[e]);
        s[e] += f;
        q[e] += f * f;
      }
    }
  }
#pragma unroll
  for (int e = 0; e < EPC; ++e) {
    red[0][threadIdx.x * EPC + e] = s[e];
    red[1][threadIdx.x * EPC + e] = q[e];
  }
  __syncthreads();
  if (g.rl == 0 && g.vec < vpr) {
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      float a = 0.f, b = 0.f;
      for (int k = 0; k < g.RS; ++k) {
        a += red[0][(k * g.VT + (threadIdx.x % g.VT)) * EPC + e];
        b += red[1][(k * g.VT + (threadIdx.x % g.VT)) * EPC + e];
      }
      psum[(long long)blockIdx.x * C + g.vec * EPC + e] = a;
      psq[(long long)blockIdx.x * C + g.vec * EPC + e] = b;
    }
  }
}

// Reduce [nparts][C] partial sums to per-channel totals inside one block of 1024 threads =
// CH channels x NL = 1024/CH partial lanes (fp64 accumulation, 4 independent loads in flight per
// array), then an NL-way LDS tree.  Returns the totals to the CH threads with pl == 0.
// CH = 16 reads coalesced 64-byte row segments; CH = 4 trades that for 4x more workgroups and 4x
// shorter per-lane loops, which is what matters when C is small and nparts is in the thousands
// (one partial row per conv tile: the early ResNet stages at large batch).
// With clear != 0 every partial is zeroed after it has been read (self-cleaning accumulation rows).
template <int CH>
struct PartialReduce {
  static constexpr int NL = 1024 / CH;
  typedef double Scratch[2][NL][CH + 1];
  static __device__ __forceinline__ void run(float* __restrict__ pa, float* __restrict__ pb, int nparts, int C,
                                             int& c, int& pl, double& ra, double& rb, Scratch& red, int clear = 0) {
    const int cl = threadIdx.x % CH;
    c = blockIdx.x * CH + cl;
    pl = threadIdx.x / CH;
    double a = 0.0, b = 0.0;
    if (c < C) {
      int p = pl;
      for (; p + 3 * NL < nparts; p += 4 * NL) {
        const float a0 = pa[(long long)p * C + c], a1 = pa[(long long)(p + NL) * C + c];
        const float a2 = pa[(long long)(p + 2 * NL) * C + c], a3 = pa[(long long)(p + 3 * NL) * C + c];
        const float b0 = pb[(long long)p * C + c], b1 = pb[(long long)(p + NL) * C + c];
        const float b2 = pb[(long long)(p + 2 * NL) * C + c], b3 = pb[(long long)(p + 3 * NL) * C + c];
        a += ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
        b += ((double)b0 + (double)b1) + ((double)b2 + (double)b3);
      }
      for (; p < nparts; p += NL) {
        a += (double)pa[(long long)p * C + c];
        b += (double)pb[(long long)p * C + c];
      }
      if (clear)
        for (p = pl; p < nparts; p += NL) {
          pa[(long long)p * C + c] = 0.f;
          pb[(long long)p * C + c] = 0.f;
        }
    }
    red[0][pl][cl] = a;
    red[1][pl][cl] = b;
    __syncthreads();
    for (int s = NL / 2; s > 0; s >>= 1) {
      if (pl < s) {
        red[0][pl][cl] += red[0][pl + s][cl];
        red[1][pl][cl] += red[1][pl + s][cl];
      }
      __syncthreads();
    }
    ra = red[0][0][cl];
    rb = red[1][0][cl];
  }
};

// [nparts][C] partial rows -> [gridDim.y][C]: workgroup (cg, rs) adds rows rs*chunk ... of 32 channels
// (128-byte coalesced row segments, 8 row lanes, fp64 accumulation).  Run ahead of a finalize-type kernel
// when there are thousands of partial rows (one per conv tile in the early ResNet stages at large batch):
// those kernels give each channel group ONE workgroup, which then walks every row alone.
__global__ void __launch_bounds__(256)
bn_compress_partials_kernel(const float* __restrict__ pa, const float* __restrict__ pb, int nparts, int C,
                            int chunk, float* __restrict__ oa, float* __restrict__ ob) {
  __shared__ double red[2][8][33];
  const int lc = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + lc;
  const int r0 = blockIdx.y * chunk;
  const int r1 = r0 + chunk < nparts ? r0 + chunk : nparts;
  double a = 0.0, b = 0.0;
  if (c < C) {
    int r = r0 + rl;
    for (; r + 24 < r1; r += 32) {
      const float a0 = pa[(long long)r * C + c], a1 = pa[(long long)(r + 8) * C + c];
      const float a2 = pa[(long long)(r + 16) * C + c], a3 = pa[(long long)(r + 24) * C + c];
      const float b0 = pb[(long long)r * C + c], b1 = pb[(long long)(r + 8) * C + c];
      const float b2 = pb[(long long)(r + 16) * C + c], b3 = pb[(long long)(r + 24) * C + c];
      a += ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
      b += ((double)b0 + (double)b1) + ((double)b2 + (double)b3);
    }
    for (; r < r1; r += 8) {
      a += (double)pa[(long long)r * C + c];
      b += (double)pb[(long long)r * C + c];
    }
  }
  red[0][rl][lc] = a;
  red[1][rl][lc] = b;
  __syncthreads();
  if (rl == 0 && c < C) {
#pragma unroll
    for (int k = 1; k < 8; ++k) {
      a += red[0][k][lc];
      b += red[1][k][lc];
    }
    oa[(long long)blockIdx.y * C + c] = (float)a;
    ob[(long long)blockIdx.y * C + c] = (float)b;
  }
}

// channels per block for the finalize-type kernels
static inline int partial_ch(int nparts, int C) { return (nparts > 512 && C < 16 * 96) ? 4 : 16; }

// count = number of reduced elements per channel (over all ranks for SyncBN).
// Updates running stats in place (momentum, unbiased variance) and emits
// mean / invstd / scale / shift.  Grid = ceil(C/CH) blocks of 1024 threads.
template <int CH>
__global__ void __launch_bounds__(1024)
bn_finalize_kernel(float* __restrict__ psum, float* __restrict__ psq, int nparts,
                                   int C, double count, float eps, float momentum,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ running_mean, float* __restrict__ running_var,
                                   float* __restrict__ mean_out, float* __restrict__ invstd_out,
                                   float* __restrict__ scale_out, float* __restrict__ shift_out,
                                   long long* __restrict__ num_batches_tracked, int clear,
                                   int lp_dtype, void* __restrict__ lp_out, float* __restrict__ rt_out) {
  __shared__ typename PartialReduce<CH>::Scratch red;
  int c, pl;
  double s, q;
  PartialReduce<CH>::run(psum, psq, nparts, C, c, pl, s, q, red, clear);
  if (pl != 0 || c >= C) return;
  if (num_batches_tracked && c == 0) num_batches_tracked[0] += 1;
  const double mean = s / count;
  double var = q / count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  mean_out[c] = (float)mean;
  invstd_out[c] = invstd;
  const float g = gamma ? gamma[c] : 1.f;
  const float b = beta ? beta[c] : 0.f;
  const float sc = g * invstd, sf = b - (float)mean * g * invstd;
  scale_out[c] = sc;
  shift_out[c] = sf;
  if (lp_out != nullptr) {
    // the coefficient tables of a consumer that normalises on load (see bn_lp_tables_kernel), without a launch
    // of their own: lp[0..C) / lp[C..2C) in the activation type, rt = those rounded values back in fp32
    if (lp_dtype == PDA_F16) {
      const f16 s16 = (f16)sc, f16v = (f16)sf;
      reinterpret_cast<f16*>(lp_out)[c] = s16;
      reinterpret_cast<f16*>(lp_out)[C + c] = f16v;
      rt_out[c] = (float)s16;
      rt_out[C + c] = (float)f16v;
    } else {
      const bf16 s16 = (bf16)sc, f16v = (bf16)sf;
      reinterpret_cast<bf16*>(lp_out)[c] = s16;
      reinterpret_cast<bf16*>(lp_out)[C + c] = f16v;
      rt_out[c] = (float)s16;
      rt_out[C + c] = (float)f16v;
    }
  }
  if (running_mean) {
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

// sums[0][c] = sum over partials, sums[1][c] = sum of squares (fp32 out, fp64 math)
template <int CH>
__global__ void __launch_bounds__(1024)
bn_reduce_partials_kernel(float* __restrict__ psum, float* __restrict__ psq, int nparts, int C,
                          float* __restrict__ sums, int clear) {
  __shared__ typename PartialReduce<CH>::Scratch red;
  int c, pl;
  double s, q;
  PartialReduce<CH>::run(psum, psq, nparts, C, c, pl, s, q, red, clear);
  if (pl != 0 || c >= C) return;
  sums[c] = (float)s;
  sums[C + c] = (float)q;
}

// Eval mode: scale/shift from running statistics.
__global__ void bn_eval_coeffs_kernel(int C, float eps, const float* __restrict__ gamma,
                                      const float* __restrict__ beta, const float* __restrict__ running_mean,
                                      const float* __restrict__ running_var, float* __restrict__ scale_out,
                                      float* __restrict__ shift_out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float invstd = rsqrtf(running_var[c] + eps);
  const float g = gamma ? gamma[c] : 1.f;
  const float b = beta ? beta[c] : 0.f;
  scale_out[c] = g * invstd;
  shift_out[c] = b - running_mean[c] * g * invstd;
}

// Every thread keeps one 16-byte channel vector for the whole grid-stride loop
// (the stride is a multiple of the vectors per row), so scale/shift live in registers.
// CAP (compile time): the activation is a ReLU clipped at `cap` (ReLU6); the plain instantiations are unchanged.
// RA (compile time): `res` is the output of a convolution whose own BatchNorm (no activation) has not been applied:
// res * res_scale + res_shift is added - the shortcut branch of a residual block, normalised here instead of in a
// pass of its own.
template <typename T, bool CAP = false, int VB = 16, bool RA = false>
__global__ void __launch_bounds__(256)
bn_apply_kernel(const T* __restrict__ y, const T* __restrict__ res, T* __restrict__ out,
                const float* __restrict__ scale, const float* __restrict__ shift, long long nvec, int C,
                int relu, unsigned char* __restrict__ mask_out, int stream_out, float cap = 0.f,
                const float* __restrict__ res_scale = nullptr, const float* __restrict__ res_shift = nullptr) {
  constexpr int EPC = VB / sizeof(T);
  const int vpr = C / EPC;
  const long long stride = (long long)gridDim.x * blockDim.x;  // host guarantees stride % vpr == 0
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  const int c0 = (int)(i % vpr) * EPC;
  float sc[EPC], sf[EPC], rs[RA ? EPC : 1];
#pragma unroll
  for (int e = 0; e < EPC; ++e) { sc[e] = scale[c0 + e]; sf[e] = shift[c0 + e]; }
  if (RA) {
#pragma unroll
    for (int e = 0; e < EPC; ++e) { rs[e] = res_scale[c0 + e]; sf[e] += res_shift[c0 + e]; }
  }
  for (; i < nvec; i += stride) {
    VecB<T, VB> v = ldv_stream<T, VB>(y + i * EPC);
    VecB<T, VB> rv;
    if (res) rv = ldv_stream<T, VB>(res + i * EPC);
    VecB<T, VB> o;
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      float f = ElemTraits<T>::to_f32(v.v[e]) * sc[e] + sf[e];
      if (RA) f += ElemTraits<T>::to_f32(rv.v[e]) * rs[e];
      else if (res) f += ElemTraits<T>::to_f32(rv.v[e]);
      if (relu) f = fmaxf(f, 0.f);
      if (CAP) f = fminf(f, cap);
      o.v[e] = ElemTraits<T>::from_f32(f);
    }
    stv_maybe_stream<T, VB>(out + i * EPC, o, stream_out);
    if (mask_out != nullptr) {
      // one byte per 16-byte vector: bit e = element e is positive.  Whoever applies this layer's ReLU
      // backward (the dgrad epilogue of the consuming conv) reads these bits instead of the activation:
      // 1/16 of the bytes.
      unsigned bits = 0u;
#pragma unroll
      for (int e = 0; e < EPC; ++e) bits |= (ElemTraits<T>::to_f32(o.v[e]) > 0.f ? 1u : 0u) << e;
      mask_out[i] = (unsigned char)bits;
    }
  }
}

// g = dout masked by the ReLU of the forward (mask from `out` when given, else
// recomputed from y*scale+shift which is exact when there was no residual).
// MASK (compile time, so that no load sits behind a run-time branch): 0 = no ReLU, 1 = mask
// recomputed from y, 2 = mask read from `out`.
// CAP: the ReLU was clipped at `cap`; values at or above it pass no gradient either.
template <typename T, int MASK, bool CAP = false, int VB = 16>
__global__ void __launch_bounds__(256)
bn_bwd_reduce_kernel(const T* __restrict__ dout, const T* __restrict__ y, const T* __restrict__ out,
                     const float* __restrict__ mean, const float* __restrict__ invstd,
                     const float* __restrict__ scale, const float* __restrict__ shift, long long M, int C,
                     int relu, float* __restrict__ pdg, float* __restrict__ pdb, int slots, float cap = 0.f) {
  constexpr int EPC = VB / sizeof(T);
  __shared__ float red[2][256 * EPC];
  ColGeom<T, VB> g(C, blockIdx.y);
  const int vpr = C / EPC;
  // The loop accumulates sum(g) and sum(g*y) only; sum(g*xhat) = invstd * (sum(g*y) - mean*sum(g)) is
  // formed once per workgroup at the end, which keeps mean/invstd out of the loop's registers
  // (the kernel is bandwidth-bound and lives on occupancy).
  float dg[EPC], db[EPC], sc[EPC], sf[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) { dg[e] = 0.f; db[e] = 0.f; sc[e] = 0.f; sf[e] = 0.f; }
  (void)relu;
  if (g.vec < vpr) {
    if (MASK == 1) {
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        sc[e] = scale[g.vec * EPC + e];
        sf[e] = shift[g.vec * EPC + e];
      }
    }
    const long long rows_per_blk = (M + gridDim.x - 1) / gridDim.x;
    const long long r0 = blockIdx.x * rows_per_blk;
    const long long r1 = r0 + rows_per_blk < M ? r0 + rows_per_blk : M;
    auto accum = [&](const VecB<T, VB>& vd, const VecB<T, VB>& vy, const VecB<T, VB>& vo) {
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        float gd = ElemTraits<T>::to_f32(vd.v[e]);
        const float yv = ElemTraits<T>::to_f32(vy.v[e]);
        if (MASK == 2) {
          const float ov = ElemTraits<T>::to_f32(vo.v[e]);
          gd = (ov > 0.f && (!CAP || ov < cap)) ? gd : 0.f;
        }
        if (MASK == 1) {
          const float fv = yv * sc[e] + sf[e];
          gd = (fv > 0.f && (!CAP || fv < cap)) ? gd : 0.f;
        }
        db[e] += gd;
        dg[e] += gd * yv;
      }
    };
    constexpr bool need_out = MASK == 2;
    long long r = r0 + g.rl;
    // four rows in flight per thread: the loop is latency-bound otherwise
    for (; r + 3 * g.RS < r1; r += 4 * g.RS) {
      VecB<T, VB> vd[4], vy[4], vo[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long long o = (r + (long long)u * g.RS) * C + (long long)g.vec * EPC;
        vd[u] = ldv<T, VB>(dout + o);
        vy[u] = ldv<T, VB>(y + o);
        if (need_out) vo[u] = ldv<T, VB>(out + o);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) accum(vd[u], vy[u], vo[u]);
    }
    for (; r < r1; r += g.RS) {
      const long long o = r * C + (long long)g.vec * EPC;
      VecB<T, VB> vd = ldv<T, VB>(dout + o);
      VecB<T, VB> vy = ldv<T, VB>(y + o);
      VecB<T, VB> vo;
      if (need_out) vo = ldv<T, VB>(out + o);
      accum(vd, vy, vo);
    }
  }
  // Row lanes that share a channel vector sit VT lanes apart: fold them inside the wave with
  // shuffles first (VT is a power of two), so only <= 4 values per vector cross LDS.
  if (g.VT < 64) {
    for (int o = g.VT; o < 64; o <<= 1) {
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        dg[e] += __shfl_xor(dg[e], o, 64);
        db[e] += __shfl_xor(db[e], o, 64);
      }
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (g.VT >= 64 || lane < g.VT) {
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      red[0][(wave * 64 + lane) * EPC + e] = dg[e];
      red[1][(wave * 64 + lane) * EPC + e] = db[e];
    }
  }
  __syncthreads();
  if (g.rl == 0 && g.vec < vpr) {
    const int v = threadIdx.x;               // == vector index inside this block's VT
    const int nent = g.VT < 64 ? 4 : g.RS;   // partial sums per vector left in LDS
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      float a = 0.f, b = 0.f;
      for (int k = 0; k < nent; ++k) {
        const int w = g.VT < 64 ? k : (v + k * g.VT) >> 6;
        const int l = g.VT < 64 ? v : (v & 63);
        a += red[0][(w * 64 + l) * EPC + e];
        b += red[1][(w * 64 + l) * EPC + e];
      }
      a = invstd[g.vec * EPC + e] * (a - mean[g.vec * EPC + e] * b);
      if (slots > 0) {  // pre-zeroed accumulation rows shared by many workgroups
        atomicAdd(pdg + (long long)(blockIdx.x % slots) * C + g.vec * EPC + e, a);
        atomicAdd(pdb + (long long)(blockIdx.x % slots) * C + g.vec * EPC + e, b);
      } else {
        pdg[(long long)blockIdx.x * C + g.vec * EPC + e] = a;
        pdb[(long long)blockIdx.x * C + g.vec * EPC + e] = b;
      }
    }
  }
}

// dy = A[c]*g + B[c]*y + K[c]  with the three per-channel coefficients prepared by
// bn_bwd_coefs_kernel (algebraically gamma*invstd*(g - dbeta/M - xhat*dgamma/M));
// coef = [5][C] = {A, B, K, scale, shift}, all held in registers (see bn_apply_kernel).
template <typename T, bool CAP = false, int VB = 16>
__global__ void __launch_bounds__(256)
bn_bwd_apply_kernel(const T* __restrict__ dout, const T* __restrict__ y, const T* __restrict__ out,
                    const float* __restrict__ coef, long long nvec, int C, int relu,
                    T* __restrict__ dy, T* __restrict__ dres, int stream_out, float cap = 0.f) {
  constexpr int EPC = VB / sizeof(T);
  const int vpr = C / EPC;
  const long long stride = (long long)gridDim.x * blockDim.x;
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  const int c0 = (int)(i % vpr) * EPC;
  float cA[EPC], cB[EPC], cK[EPC], sc[EPC], sf[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) {
    cA[e] = coef[c0 + e];
    cB[e] = coef[C + c0 + e];
    cK[e] = coef[2 * C + c0 + e];
    sc[e] = coef[3 * C + c0 + e];
    sf[e] = coef[4 * C + c0 + e];
  }
  for (; i < nvec; i += stride) {
    VecB<T, VB> vd = ldv_stream<T, VB>(dout + i * EPC);
    VecB<T, VB> vy = ldv_stream<T, VB>(y + i * EPC);
    VecB<T, VB> vo;
    if (relu && out) vo = ldv<T, VB>(out + i * EPC);
    VecB<T, VB> o, og;
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      float gd = ElemTraits<T>::to_f32(vd.v[e]);
      const float yv = ElemTraits<T>::to_f32(vy.v[e]);
      if (relu) {
        const float fv = out ? ElemTraits<T>::to_f32(vo.v[e]) : yv * sc[e] + sf[e];
        const bool pos = fv > 0.f && (!CAP || fv < cap);
        gd = pos ? gd : 0.f;
      }
      o.v[e] = ElemTraits<T>::from_f32(cA[e] * gd + cB[e] * yv + cK[e]);
      og.v[e] = ElemTraits<T>::from_f32(gd);
    }
    stv_maybe_stream<T, VB>(dy + i * EPC, o, stream_out);
    if (dres) stv_maybe_stream<T, VB>(dres + i * EPC, og, stream_out);
  }
}

// Reduce the dgamma / dbeta partials and (when coef != null) emit the backward
// coefficients in the same launch.  For SyncBN the totals are all-reduced first and
// bn_bwd_coefs_kernel runs separately on the global sums.
__device__ __forceinline__ void bwd_coefs(int c, int C, float dg, float db, const float* mean, const float* invstd,
                                          const float* scale, const float* shift, const float* gamma,
                                          float inv_count, float* coef) {
  const float gm = gamma ? gamma[c] : 1.f;
  const float is = invstd[c];
  const float A = gm * is;
  const float B = -A * is * dg * inv_count;
  coef[c] = A;
  coef[C + c] = B;
  coef[2 * C + c] = -A * db * inv_count - B * mean[c];
  coef[3 * C + c] = scale[c];
  coef[4 * C + c] = shift[c];
}

template <int CH>
__global__ void __launch_bounds__(1024)
bn_bwd_finalize_kernel(float* __restrict__ pdg, float* __restrict__ pdb, int nparts, int C,
                       float* __restrict__ dgamma, float* __restrict__ dbeta, const float* __restrict__ mean,
                       const float* __restrict__ invstd, const float* __restrict__ scale,
                       const float* __restrict__ shift, const float* __restrict__ gamma, float inv_count,
                       float* __restrict__ coef, float* __restrict__ acc_dgamma, float* __restrict__ acc_dbeta,
                       int clear, int raw) {
  __shared__ typename PartialReduce<CH>::Scratch red;
  int c, pl;
  double a, b;
  PartialReduce<CH>::run(pdg, pdb, nparts, C, c, pl, a, b, red, clear);
  if (pl != 0 || c >= C) return;
  // raw partials (written by a dgrad epilogue): pdb = sum(g), pdg = sum(g*y) -> sum(g*xhat)
  if (raw) a = (double)invstd[c] * (a - (double)mean[c] * b);
  dgamma[c] = (float)a;
  dbeta[c] = (float)b;
  // parameter gradients accumulated straight into the optimizer's / reducer's buffers
  if (acc_dgamma) acc_dgamma[c] += (float)a;
  if (acc_dbeta) acc_dbeta[c] += (float)b;
  if (coef) bwd_coefs(c, C, (float)a, (float)b, mean, invstd, scale, shift, gamma, inv_count, coef);
}

__global__ void bn_bwd_coefs_kernel(const float* __restrict__ dg, const float* __restrict__ db, int C,
                                    const float* __restrict__ mean, const float* __restrict__ invstd,
                                    const float* __restrict__ scale, const float* __restrict__ shift,
                                    const float* __restrict__ gamma, float inv_count, float* __restrict__ coef) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) bwd_coefs(c, C, dg[c], db[c], mean, invstd, scale, shift, gamma, inv_count, coef);
}

// ------------------------------------------------------------- host side ----
// bytes per lane access for a channel count (see VB above); 0: not covered (odd 16-bit channel counts)
static inline int bn_vec_bytes(int C, int es) {
  if (C % (16 / es) == 0) return 16;
  return (C * es) % 4 == 0 ? 4 : 0;
}

static inline int col_blocks_y(int C, int es) {
  const int vb = bn_vec_bytes(C, es);
  const int vpr = C / ((vb ? vb : 16) / es);
  int vt = vpr < 256 ? vpr : 256;
  int p = 1;
  while (p * 2 <= vt) p *= 2;
  return (vpr + p - 1) / p;
}

#define PDA_DISPATCH_T(dtype, CALL)                 \
  switch (dtype) {                                  \
    case PDA_F32: { typedef float T; CALL; } break; \
    case PDA_F16: { typedef f16 T; CALL; } break;   \
    case PDA_BF16: { typedef bf16 T; CALL; } break; \
    default: return -6;                             \
  }

// Number of row-partials (= workgroups along M) a column reduce over [M][C] will produce; python
// allocates [n][C].  A block covers RS = 256/VT rows per pass (VT = channel vectors per block), so
// wide, short tensors (late ResNet stages: M small, C = 2048) still get ~1024 workgroups with at
// least four passes each instead of a handful of long serial loops.
// The call sees the element type T and the compile-time access width VB (16 or 4).  (Variadic: the launch macro
// expands to <<<a, b, c, d>>>, whose commas must not split the argument when it is handed on.)
#define PDA_DISPATCH_TVB_CASE(TYPE, ...) { typedef TYPE T; __VA_ARGS__; } break;
#define PDA_DISPATCH_TVB_WIDTH(dtype, ...)               \
  switch (dtype) {                                       \
    case PDA_F32: PDA_DISPATCH_TVB_CASE(float, __VA_ARGS__) \
    case PDA_F16: PDA_DISPATCH_TVB_CASE(f16, __VA_ARGS__)   \
    case PDA_BF16: PDA_DISPATCH_TVB_CASE(bf16, __VA_ARGS__) \
    default: return -6;                                  \
  }
#define PDA_DISPATCH_TVB(dtype, vb, ...)                                            \
  if ((vb) == 16) { constexpr int VB = 16; PDA_DISPATCH_TVB_WIDTH(dtype, __VA_ARGS__) } \
  else { constexpr int VB = 4; PDA_DISPATCH_TVB_WIDTH(dtype, __VA_ARGS__) }

PDA_EXPORT int pda_bn_num_partials(long long M, int C, int dtype) {
  const int es = dtype == PDA_F32 ? 4 : 2;
  const int vb = bn_vec_bytes(C, es);
  const int vpr = C / ((vb ? vb : 16) / es);
  int vt = vpr < 256 ? vpr : 256;
  int p = 1;
  while (p * 2 <= vt) p *= 2;
  const long long rs = 256 / p;
  long long n = (M + 4 * rs - 1) / (4 * rs);
  if (n > 1024) n = 1024;
  if (n < 1) n = 1;
  return (int)n;
}

PDA_EXPORT int pda_bn_stats_partial(int dtype, const void* y, long long M, int C, float* psum, float* psq,
                                    hipStream_t st) {
  const int es = dtype == PDA_F32 ? 4 : 2;
  const int vb = bn_vec_bytes(C, es);
  if (vb == 0 || ((uintptr_t)y & (vb - 1))) return -1;
  const int nb = pda_bn_num_partials(M, C, dtype);
  PDA_DISPATCH_TVB(dtype, vb, hipLaunchKernelGGL((bn_stats_partial_kernel<T, VB>), dim3(nb, col_blocks_y(C, es)),
                                                 dim3(256), 0, st, (const T*)y, M, C, psum, psq));
  return pda_check_launch();
}

PDA_EXPORT int pda_bn_reduce_partials(float* psum, float* psq, int nparts, int C, float* sums, int clear,
                                      hipStream_t st) {
  if (partial_ch(nparts, C) == 4)
    hipLaunchKernelGGL(bn_reduce_partials_kernel<4>, dim3((C + 3) / 4), dim3(1024), 0, st, psum, psq, nparts, C,
                       sums, clear);
  else
    hipLaunchKernelGGL(bn_reduce_partials_kernel<16>, dim3((C + 15) / 16), dim3(1024), 0, st, psum, psq, nparts, C,
                       sums, clear);
  return pda_check_launch();
}

// Rows a compression pass should leave for nparts partial rows (0: not worth a pass).
PDA_EXPORT int pda_bn_compress_rows(int nparts, int C) {
  if (nparts < 1024) return 0;
  int rs = nparts / 32;  // >= 32 rows per workgroup
  const int cap = 2048 / ((C + 31) / 32) > 8 ? 2048 / ((C + 31) / 32) : 8;  // ~2048 workgroups at most
  if (rs > cap) rs = cap;
  if (rs > 128) rs = 128;
  return rs;
}

PDA_EXPORT int pda_bn_compress_partials(const float* pa, const float* pb, int nparts, int C, float* oa, float* ob,
                                        int rs, hipStream_t st) {
  if (rs < 1 || nparts < 1) return -1;
  const int chunk = (nparts + rs - 1) / rs;
  hipLaunchKernelGGL(bn_compress_partials_kernel, dim3((C + 31) / 32, rs), dim3(256), 0, st, pa, pb, nparts, C,
                     chunk, oa, ob);
  return pda_check_launch();
}

static int bn_finalize_impl(float* psum, float* psq, int nparts, int C, double count, float eps,
                            float momentum, const float* gamma, const float* beta, float* running_mean,
                            float* running_var, float* mean_out, float* invstd_out, float* scale_out,
                            float* shift_out, long long* num_batches_tracked, int clear, int lp_dtype, void* lp_out,
                            float* rt_out, hipStream_t st) {
  if (lp_out != nullptr && lp_dtype != PDA_F16 && lp_dtype != PDA_BF16) return -6;
  if (partial_ch(nparts, C) == 4)
    hipLaunchKernelGGL(bn_finalize_kernel<4>, dim3((C + 3) / 4), dim3(1024), 0, st, psum, psq, nparts, C, count,
                       eps, momentum, gamma, beta, running_mean, running_var, mean_out, invstd_out, scale_out,
                       shift_out, num_batches_tracked, clear, lp_dtype, lp_out, rt_out);
  else
    hipLaunchKernelGGL(bn_finalize_kernel<16>, dim3((C + 15) / 16), dim3(1024), 0, st, psum, psq, nparts, C, count,
                       eps, momentum, gamma, beta, running_mean, running_var, mean_out, invstd_out, scale_out,
                       shift_out, num_batches_tracked, clear, lp_dtype, lp_out, rt_out);
  return pda_check_launch();
}

PDA_EXPORT int pda_bn_finalize(float* psum, float* psq, int nparts, int C, double count, float eps,
                               float momentum, const float* gamma, const float* beta, float* running_mean,
                               float* running_var, float* mean_out, float* invstd_out, float* scale_out,
                               float* shift_out, long long* num_batches_tracked, int clear, hipStream_t st) {
  return bn_finalize_impl(psum, psq, nparts, C, count, eps, momentum, gamma, beta, running_mean, running_var, mean_out,
                          invstd_out, scale_out, shift_out, num_batches_tracked, clear, 0, nullptr, nullptr, st);
}

// pda_bn_finalize that also writes the 16-bit coefficient tables (lp_out: [2][C] of lp_dtype, rt_out: [2][C] fp32).
PDA_EXPORT int pda_bn_finalize_lp(float* psum, float* psq, int nparts, int C, double count, float eps,
                                  float momentum, const float* gamma, const float* beta, float* running_mean,
                                  float* running_var, float* mean_out, float* invstd_out, float* scale_out,
                                  float* shift_out, long long* num_batches_tracked, int clear, int lp_dtype,
                                  void* lp_out, float* rt_out, hipStream_t st) {
  return bn_finalize_impl(psum, psq, nparts, C, count, eps, momentum, gamma, beta, running_mean, running_var, mean_out,
                          invstd_out, scale_out, shift_out, num_batches_tracked, clear, lp_dtype, lp_out, rt_out, st);
}

PDA_EXPORT int pda_bn_eval_coeffs(int C, float eps, const float* gamma, const float* beta,
                                  const float* running_mean, const float* running_var, float* scale_out,
                                  float* shift_out, hipStream_t st) {
  hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3((C + 127) / 128), dim3(128), 0, st, C, eps, gamma, beta,
                     running_mean, running_var, scale_out, shift_out);
  return pda_check_launch();
}

// Coefficient tables for a consumer that normalises its input on load (conv_igemm.hip, ConvArgs::in_scale):
// lp[0][c], lp[1][c] = scale, shift rounded to the activation type; rt[0][c], rt[1][c] = those rounded values
// back in fp32, which is what the backward's ReLU-mask recomputation must use so that it decides every element
// the way the forward did.
template <typename T>
__global__ void bn_lp_tables_kernel(int C, const float* __restrict__ scale, const float* __restrict__ shift,
                                    T* __restrict__ lp, float* __restrict__ rt) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const T s = ElemTraits<T>::from_f32(scale[c]), f = ElemTraits<T>::from_f32(shift[c]);
  lp[c] = s;
  lp[C + c] = f;
  rt[c] = ElemTraits<T>::to_f32(s);
  rt[C + c] = ElemTraits<T>::to_f32(f);
}

PDA_EXPORT int pda_bn_lp_tables(int dtype, int C, const float* scale, const float* shift, void* lp, float* rt,
                                hipStream_t st) {
  if (C <= 0) return 0;
  if (dtype == PDA_F16)
    hipLaunchKernelGGL(bn_lp_tables_kernel<f16>, dim3((C + 127) / 128), dim3(128), 0, st, C, scale, shift, (f16*)lp, rt);
  else if (dtype == PDA_BF16)
    hipLaunchKernelGGL(bn_lp_tables_kernel<bf16>, dim3((C + 127) / 128), dim3(128), 0, st, C, scale, shift, (bf16*)lp, rt);
  else
    return -6;
  return pda_check_launch();
}

// Grid for the register-resident-coefficient kernels: blocks*256 must be a multiple of
// the vectors per row so a thread's channel vector never changes along its loop.
static inline int ew_blocks_chan(long long nvec, int vpr) {
  long long b = (nvec + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  long long g = vpr, r = 256;
  while (r) { const long long t = g % r; g = r; r = t; }  // gcd(vpr, 256)
  const long long m = vpr / g;                             // blocks must be a multiple of m
  b = ((b + m - 1) / m) * m;
  return (int)b;
}

static int bn_apply_impl(int dtype, const void* y, const void* res, void* out, const float* scale,
                         const float* shift, long long M, int C, int relu, unsigned char* mask_out, hipStream_t st) {
  const int es = dtype == PDA_F32 ? 4 : 2;
  const int vb = bn_vec_bytes(C, es);
  if (vb == 0 || (M * C * es) % vb != 0 || (mask_out != nullptr && vb != 16)) return -1;
  const long long nvec = M * C / (vb / es);
  if (nvec == 0) return 0;
  const int vpr = C / (vb / es);
  PDA_DISPATCH_TVB(dtype, vb, hipLaunchKernelGGL((bn_apply_kernel<T, false, VB>), dim3(ew_blocks_chan(nvec, vpr)), dim3(256),
                                                 0, st, (const T*)y, (const T*)res, (T*)out, scale, shift, nvec, C, relu,
                                                 mask_out, (int)(nvec * vb >= pda_stream_min_bytes())));
  return pda_check_launch();
}

PDA_EXPORT int pda_bn_apply(int dtype, const void* y, const void* res, void* out, const float* scale,
                            const float* shift, long long M, int C, int relu, hipStream_t st) {
  return bn_apply_impl(dtype, y, res, out, scale, shift, M, C, relu, nullptr, st);
}

// out = relu(y * scale + shift + res * res_scale + res_shift): both branches of a residual block's end normalised in
// one pass (res = the shortcut convolution's raw output); mask_out as in pda_bn_apply_mask, or null.
PDA_EXPORT int pda_bn_apply_res_affine(int dtype, const void* y, const void* res, void* out, const float* scale,
                                       const float* shift, const float* res_scale, const float* res_shift, long long M,
                                       int C, int relu, unsigned char* mask_out, hipStream_t st) {
  const int es = dtype == PDA_F32 ? 4 : 2;
  const int vb = bn_vec_bytes(C, es);
  if (!res || !res_scale || !res_shift) return -11;
  if (vb == 0 || (M * C * es) % vb != 0 || (mask_out != nullptr && vb != 16)) return -1;
  const long long nvec = M * C / (vb / es);
  if (nvec == 0) return 0;
  const int vpr = C / (vb / es);
  PDA_DISPATCH_TVB(dtype, vb, hipLaunchKernelGGL((bn_apply_kernel<T, false, VB, true>), dim3(ew_blocks_chan(nvec, vpr)),
                                                 dim3(256), 0, st, (const T*)y, (const T*)res, (T*)out, scale, shift, nvec,
                                                 C, relu, mask_out, (int)(nvec * vb >= pda_stream_min_bytes()), 0.f,
                                                 res_scale, res_shift));
  return pda_check_launch();
}

// out = min(relu(y * scale + shift + res), cap): BatchNorm + clipped ReLU (ReLU6) in one pass.
PDA_EXPORT int pda_bn_apply_cap(int dtype, const void* y, const void* res, void* out, const float* scale,
                                const float* shift, long long M, int C, float cap, hipStream_t st) {
  const int es = dtype == PDA_F32 ? 4 : 2;
  const int vb = bn_vec_bytes(C, es);
  if (vb == 0 || (M * C * es) % vb != 0) return -1;
  const long long nvec = M * C / (vb / es);
  if (nvec == 0) return 0;
  const int vpr = C / (vb / es);
  PDA_DISPATCH_TVB(dtype, vb, hipLaunchKernelGGL((bn_apply_kernel<T, true, VB>), dim3(ew_blocks_chan(nvec, vpr)), dim3(256),
                                                 0, st, (const T*)y, (const T*)res, (T*)out, scale, shift, nvec, C, 1,
                                                 (unsigned char*)nullptr, (int)(nvec * vb >= pda_stream_min_bytes()), cap));
  return pda_check_launch();
}

// As pda_bn_apply, and mask_out[i] (one byte per 16-byte vector of `out`, M*C*sizeof(T)/16 bytes) gets the
// "element is positive" bits of vector i.
PDA_EXPORT int pda_bn_apply_mask(int dtype, const void* y, const void* res, void* out, const float* scale,
                                 const float* shift, long long M, int C, int relu, unsigned char* mask_out,
                                 hipStream_t st) {
  return bn_apply_impl(dtype, y, res, out, scale, shift, M, C, relu, mask_out, st);
}

PDA_EXPORT int pda_bn_bwd_reduce(int dtype, const void* dout, const void* y, const void* out, const float* mean,
                                 const float* invstd, const float* scale, const float* shift, long long M, int C,
                                 int relu, float* pdg, float* pdb, int slots, hipStream_t st) {
  const int es = dtype == PDA_F32 ? 4 : 2;
  const int vb = bn_vec_bytes(C, es);
  if (vb == 0) return -1;
  const int nb = pda_bn_num_partials(M, C, dtype);
  const int mask = !relu ? 0 : (out ? 2 : 1);
#define PDA_BWD_REDUCE(MASK)                                                                                       \
  PDA_DISPATCH_TVB(dtype, vb,                                                                                      \
                   hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, MASK, false, VB>), dim3(nb, col_blocks_y(C, es)),   \
                                      dim3(256), 0, st, (const T*)dout, (const T*)y, (const T*)out, mean, invstd, \
                                      scale, shift, M, C, relu, pdg, pdb, slots))
  if (mask == 0) { PDA_BWD_REDUCE(0); }
  else if (mask == 1) { PDA_BWD_REDUCE(1); }
  else { PDA_BWD_REDUCE(2); }
#undef PDA_BWD_REDUCE
  return pda_check_launch();
}

// pda_bn_bwd_reduce for a ReLU clipped at `cap`: elements at or above the cap pass no gradient.
PDA_EXPORT int pda_bn_bwd_reduce_cap(int dtype, const void* dout, const void* y, const void* out, const float* mean,
                                     const float* invstd, const float* scale, const float* shift, long long M, int C,
                                     float cap, float* pdg, float* pdb, int slots, hipStream_t st) {
  const int es = dtype == PDA_F32 ? 4 : 2;
  const int vb = bn_vec_bytes(C, es);
  if (vb == 0) return -1;
  const int nb = pda_bn_num_partials(M, C, dtype);
#define PDA_BWD_REDUCE_CAP(MASK)                                                                                  \
  PDA_DISPATCH_TVB(dtype, vb,                                                                                     \
                   hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, MASK, true, VB>), dim3(nb, col_blocks_y(C, es)),   \
                                      dim3(256), 0, st, (const T*)dout, (const T*)y, (const T*)out, mean, invstd, \
                                      scale, shift, M, C, 1, pdg, pdb, slots, cap))
  if (out == nullptr) { PDA_BWD_REDUCE_CAP(1); }
  else { PDA_BWD_REDUCE_CAP(2); }
#undef PDA_BWD_REDUCE_CAP
  return pda_check_launch();
}

// Reduce dgamma/dbeta partials; with coef != null also emits the bwd-apply coefficients ([5][C]).
PDA_EXPORT int pda_bn_bwd_finalize(float* pdg, float* pdb, int nparts, int C, float* dgamma,
                                   float* dbeta, const float* mean, const float* invstd, const float* scale,
                                   const float* shift, const float* gamma, double count, float* coef,
                                   float* acc_dgamma, float* acc_dbeta, int clear, int raw, hipStream_t st) {
  if (partial_ch(nparts, C) == 4)
    hipLaunchKernelGGL(bn_bwd_finalize_kernel<4>, dim3((C + 3) / 4), dim3(1024), 0, st, pdg, pdb, nparts, C, dgamma,
                       dbeta, mean, invstd, scale, shift, gamma, (float)(1.0 / count), coef, acc_dgamma, acc_dbeta,
                       clear, raw);
  else
    hipLaunchKernelGGL(bn_bwd_finalize_kernel<16>, dim3((C + 15) / 16), dim3(1024), 0, st, pdg, pdb, nparts, C,
                       dgamma, dbeta, mean, invstd, scale, shift, gamma, (float)(1.0 / count), coef, acc_dgamma,
                       acc_dbeta, clear, raw);
  return pda_check_launch();
}

PDA_EXPORT int pda_bn_bwd_coefs(const float* dg, const float* db, int C, const float* mean, const float* invstd,
                                const float* scale, const float* shift, const float* gamma, double count,
                                float* coef, hipStream_t st) {
  hipLaunchKernelGGL(bn_bwd_coefs_kernel, dim3((C + 127) / 128), dim3(128), 0, st, dg, db, C, mean, invstd, scale,
                     shift, gamma, (float)(1.0 / count), coef);
  return pda_check_launch();
}

PDA_EXPORT int pda_bn_bwd_apply(int dtype, const void* dout, const void* y, const void* out, const float* coef,
                                long long M, int C, int relu, void* dy, void* dres, hipStream_t st) {
  const int es = dtype == PDA_F32 ? 4 : 2;
  const int vb = bn_vec_bytes(C, es);
  if (vb == 0 || (M * C * es) % vb != 0) return -1;
  const long long nvec = M * C / (vb / es);
  if (nvec == 0) return 0;
  const int vpr = C / (vb / es);
  PDA_DISPATCH_TVB(dtype, vb, hipLaunchKernelGGL((bn_bwd_apply_kernel<T, false, VB>), dim3(ew_blocks_chan(nvec, vpr)),
                                                 dim3(256), 0, st, (const T*)dout, (const T*)y, (const T*)out, coef, nvec,
                                                 C, relu, (T*)dy, (T*)dres, (int)(nvec * vb >= pda_stream_min_bytes())));
  return pda_check_launch();
}

// pda_bn_bwd_apply for a ReLU clipped at `cap`.
PDA_EXPORT int pda_bn_bwd_apply_cap(int dtype, const void* dout, const void* y, const void* out, const float* coef,
                                    long long M, int C, float cap, void* dy, void* dres, hipStream_t st) {
  const int es = dtype == PDA_F32 ? 4 : 2;
  const int vb = bn_vec_bytes(C, es);
  if (vb == 0 || (M * C * es) % vb != 0) return -1;
  const long long nvec = M * C / (vb / es);
  if (nvec == 0) return 0;
  const int vpr = C / (vb / es);
  PDA_DISPATCH_TVB(dtype, vb, hipLaunchKernelGGL((bn_bwd_apply_kernel<T, true, VB>), dim3(ew_blocks_chan(nvec, vpr)),
                                                 dim3(256), 0, st, (const T*)dout, (const T*)y, (const T*)out, coef, nvec,
                                                 C, 1, (T*)dy, (T*)dres, (int)(nvec * vb >= pda_stream_min_bytes()), cap));
  return pda_check_launch();
}
