// C[M][N] = A[M][K] * B[N][K]^T (+ bias[N]) for fp16 / bf16 / fp32 with ANY K and N: the pointwise
// convolutions of the mobile architectures whose channel counts (16, 24, 58, 116, 232 ...) the implicit-GEMM
// kernels in conv_igemm.hip do not step through (those advance K in 32-element slices of 16-byte chunks and
// need 16-byte aligned rows).  Forward: A = activations [pixels][Cin], B = the filter [K][Cin]; input
// gradient: A = dY [pixels][K], B = the transposed filter [Cin][K].  (Reference trigger: `model(images)` /
// `loss.backward()` with `--arch` = any torchvision name, /root/reference/1.dataparallel.py:23-35,97-102.)
//
// 128x128 output tile per 256-thread workgroup, 2x2 waves of 64x64 (v_mfma_f32_32x32x16 for the 16-bit types,
// the exact v_mfma_f32_32x32x2_f32 for fp32), K steps of four 16-byte chunks (32 / 16 elements),
// register-staged through two XOR-swizzled LDS buffers like the register-staged tiles of conv_igemm.hip.
// What differs is the operand fetch: a row is only guaranteed element alignment, so a thread's 8-element
// chunk is fetched as one 16-byte access when the row pitch and the chunk allow it (pitch a multiple of 8
// elements, chunk inside the row) and element by element with zero fill otherwise - the K tail (and rows past
// M / N) therefore multiply exact zeros.
#include "common.h"

namespace {

template <typename T> struct Mma;
template <> struct Mma<f16> {
  typedef f16x8 frag;
  static __device__ __forceinline__ f32x16 run(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};
template <> struct Mma<bf16> {
  typedef bf16x8 frag;
  static __device__ __forceinline__ f32x16 run(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};

struct GemmArgs {
  const void* a;
  const void* b;
  const float* bias;
  void* c;
  int M, N, K;
  long long lda, ldb, ldc;   // row pitches in elements
  int relu;
};

// one 16-byte chunk of a row starting at column k (zero past the row's end `kend`); `vec`: the row pitch keeps
// every in-row chunk 16-byte aligned
template <typename T>
__device__ __forceinline__ u32x4 fetch16(const T* __restrict__ row, int k, int kend, bool rowok, bool vec) {
  constexpr int E = 16 / sizeof(T);
  u32x4 r = {0u, 0u, 0u, 0u};
  if (!rowok || k >= kend) return r;
  if (vec && k + E <= kend) return *reinterpret_cast<const u32x4*>(row + k);
  if (k + E <= kend) {
    // a chunk inside a row whose pitch is not a multiple of 16 bytes (58, 116 channels ...): global memory
    // takes a 16-byte access at element alignment (the HSA queues run in the unaligned access mode), so the
    // chunk is still ONE load - only the chunk that straddles the end of the row goes element by element
    __builtin_memcpy(&r, row + k, 16);
    return r;
  }
  if constexpr (sizeof(T) == 4) {
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = (k + i < kend) ? __builtin_bit_cast(unsigned, row[k + i]) : 0u;
  } else {
    unsigned short e[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) e[i] = (k + i < kend) ? __builtin_bit_cast(unsigned short, row[k + i]) : (unsigned short)0;
#pragma unroll
    for (int d = 0; d < 4; ++d) r[d] = (unsigned)e[2 * d] | ((unsigned)e[2 * d + 1] << 16);
  }
  return r;
}

template <typename T>
__global__ void __launch_bounds__(256) gemm_anyk_kernel(GemmArgs g) {
  constexpr int BM = 128, BN = 128, ROWB = 64, CPR = 4;   // 32-deep steps: 4 chunks of 16 bytes per tile row
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * (BM + BN) * ROWB];
  unsigned char* ldsA = lds;
  unsigned char* ldsB = lds + 2 * BM * ROWB;
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int wm = wid >> 1, wn = wid & 1, l31 = lane & 31, lh = lane >> 5;
  const int tiles_n = (g.N + BN - 1) / BN;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const T* __restrict__ A = static_cast<const T*>(g.a);
  const T* __restrict__ B = static_cast<const T*>(g.b);
  constexpr int EPC = 16 / (int)sizeof(T), BK = 4 * EPC;
  const bool avec = (g.lda % EPC) == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0;
  const bool bvec = (g.ldb % EPC) == 0 && (reinterpret_cast<uintptr_t>(B) & 15) == 0;
  const int kc = t & (CPR - 1), trow = t / CPR;            // 64 rows per pass, two passes per operand
  // the swizzle spreads the 16 lanes of a ds_read_b128 group over all banks: chunk ^= (row / 4) & 3
  auto slot = [](int row, int chunk) -> int { return row * ROWB + ((chunk ^ ((row >> 2) & 3)) << 4); };
  u32x4 ra[2], rb[2];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ma = m0 + trow + 64 * i, nb = n0 + trow + 64 * i;
      ra[i] = fetch16<T>(A + (long long)ma * g.lda, k0 + kc * EPC, g.K, ma < g.M, avec);
      rb[i] = fetch16<T>(B + (long long)nb * g.ldb, k0 + kc * EPC, g.K, nb < g.N, bvec);
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      *reinterpret_cast<u32x4*>(ldsA + buf * BM * ROWB + slot(trow + 64 * i, kc)) = ra[i];
      *reinterpret_cast<u32x4*>(ldsB + buf * BN * ROWB + slot(trow + 64 * i, kc)) = rb[i];
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  const int nk = (g.K + BK - 1) / BK;
  gload(0);
  lstore(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    if (more) gload((kt + 1) * BK);
    const unsigned char* Ab = ldsA + (kt & 1) * BM * ROWB;
    const unsigned char* Bb = ldsB + (kt & 1) * BN * ROWB;
    if constexpr (sizeof(T) == 2) {
      typedef typename Mma<T>::frag frag;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        frag fa[2], fb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const frag*>(Ab + slot(wm * 64 + i * 32 + l31, kk * 2 + lh));
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const frag*>(Bb + slot(wn * 64 + j * 32 + l31, kk * 2 + lh));
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = Mma<T>::run(fa[i], fb[j], acc[i][j]);
      }
    } else {
      // fp32: a chunk holds 4 k values; lane half h takes k = 2h, 2h + 1 of it (A and B agree on the order)
#pragma unroll
      for (int c = 0; c < CPR; ++c) {
        f32x2 fa[2], fb[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const f32x2*>(Ab + slot(wm * 64 + i * 32 + l31, c) + lh * 8);
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[j] = *reinterpret_cast<const f32x2*>(Bb + slot(wn * 64 + j * 32 + l31, c) + lh * 8);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
          }
      }
    }
    if (more) lstore((kt + 1) & 1);
    __syncthreads();
  }
  // C/D map of the 32x32 MFMA: col = lane & 31, row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)
  T* __restrict__ C = static_cast<T*>(g.c);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn * 64 + j * 32 + l31;
      const float bv = (g.bias != nullptr && n < g.N) ? g.bias[n] : 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (m < g.M && n < g.N) {
          float v = acc[i][j][e] + bv;
          if (g.relu) v = fmaxf(v, 0.f);
          C[(long long)m * g.ldc + n] = ElemTraits<T>::from_f32(v);
        }
      }
    }
}

}  // namespace

// dtype: PDA_F16 / PDA_BF16 / PDA_F32.  Returns 0, <0 for unsupported arguments, >0 = hipError_t.
PDA_EXPORT int pda_gemm_anyk(int dtype, const void* a, const void* b, const float* bias, void* c, int M, int N, int K,
                             long long lda, long long ldb, long long ldc, int relu, hipStream_t stream) {
  if (M <= 0 || N <= 0) return 0;
  if (K <= 0 || lda < K || ldb < K || ldc < N) return -1;
  if (dtype != PDA_F16 && dtype != PDA_BF16 && dtype != PDA_F32) return -6;
  const long long tiles = (long long)((M + 127) / 128) * ((N + 127) / 128);
  if (tiles > 0x7fffffffLL) return -4;
  GemmArgs g;
  g.a = a; g.b = b; g.bias = bias; g.c = c; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.relu = relu;
  if (dtype == PDA_F16) hipLaunchKernelGGL((gemm_anyk_kernel<f16>), dim3((unsigned)tiles), dim3(256), 0, stream, g);
  else if (dtype == PDA_F32) hipLaunchKernelGGL((gemm_anyk_kernel<float>), dim3((unsigned)tiles), dim3(256), 0, stream, g);
  else hipLaunchKernelGGL((gemm_anyk_kernel<bf16>), dim3((unsigned)tiles), dim3(256), 0, stream, g);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
