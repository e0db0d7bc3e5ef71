// Adasum on the device (op=hvd.Adasum, /root/reference/5.2.horovod_pytorch_mnist.py:30,184).
//
// Pairwise rule: adasum(a, b) = (1 - a.b / 2|a|^2) a + (1 - a.b / 2|b|^2) b.  The ranks run it as
// vector-halving / distance-doubling: at distance d a rank trades half of its current segment with rank^d,
// so it holds the SAME piece of both operands; the three scalars a.b, |a|^2, |b|^2 are sums over the pieces
// of all 2d ranks that share the operand pair.  Two kernels per level:
//   adasum_dots    : partial (a.b, |a|^2, |b|^2) of this rank's piece, accumulated in fp64;
//   adasum_combine : sums the gathered partials of the rank's block, forms the two coefficients and writes
//                    ca * a + cb * b over the piece - no host round trip for the scalars.
#include "common.h"

template <typename T>
__global__ void __launch_bounds__(256)
adasum_dots_kernel(const T* __restrict__ a, const T* __restrict__ b, long long n, double* __restrict__ out3) {
  __shared__ double red[3][4];
  double d = 0.0, na = 0.0, nb = 0.0;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const double x = (double)ElemTraits<T>::to_f32(a[i]), y = (double)ElemTraits<T>::to_f32(b[i]);
    d += x * y; na += x * x; nb += y * y;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    d += __shfl_xor(d, o, 64); na += __shfl_xor(na, o, 64); nb += __shfl_xor(nb, o, 64);
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) { red[0][w] = d; red[1][w] = na; red[2][w] = nb; }
  __syncthreads();
  if (threadIdx.x < 3) {
    const double s = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
    atomicAdd(out3 + threadIdx.x, s);
  }
}

// parts: [world][3] gathered partials; the operand pair is shared by ranks [blk, blk + nblk).
// dst = ca * a + cb * b (dst may alias a or b).
template <typename T>
__global__ void __launch_bounds__(256)
adasum_combine_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ dst, long long n,
                      const double* __restrict__ parts, int blk, int nblk) {
  double d = 0.0, na = 0.0, nb = 0.0;
  for (int r = blk; r < blk + nblk; ++r) { d += parts[r * 3]; na += parts[r * 3 + 1]; nb += parts[r * 3 + 2]; }
  const float ca = (float)(na > 0.0 ? 1.0 - d / (2.0 * na) : 1.0);
  const float cb = (float)(nb > 0.0 ? 1.0 - d / (2.0 * nb) : 1.0);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    dst[i] = ElemTraits<T>::from_f32(ca * ElemTraits<T>::to_f32(a[i]) + cb * ElemTraits<T>::to_f32(b[i]));
}

static unsigned adasum_blocks(long long n) {
  long long b = (n + 255) / 256;
  return (unsigned)(b < 1 ? 1 : b > 1024 ? 1024 : b);
}

// out3 (3 doubles) must be zero on entry.
PDA_EXPORT int pda_adasum_dots(int dtype, const void* a, const void* b, long long n, double* out3, hipStream_t st) {
  if (n <= 0) return 0;
  switch (dtype) {
    case PDA_F32: hipLaunchKernelGGL((adasum_dots_kernel<float>), dim3(adasum_blocks(n)), dim3(256), 0, st, (const float*)a, (const float*)b, n, out3); break;
    case PDA_F16: hipLaunchKernelGGL((adasum_dots_kernel<f16>), dim3(adasum_blocks(n)), dim3(256), 0, st, (const f16*)a, (const f16*)b, n, out3); break;
    case PDA_BF16: hipLaunchKernelGGL((adasum_dots_kernel<bf16>), dim3(adasum_blocks(n)), dim3(256), 0, st, (const bf16*)a, (const bf16*)b, n, out3); break;
    default: return -6;
  }
  return pda_check_launch();
}

PDA_EXPORT int pda_adasum_combine(int dtype, const void* a, const void* b, void* dst, long long n, const double* parts,
                                  int blk, int nblk, hipStream_t st) {
  if (n <= 0) return 0;
  if (blk < 0 || nblk < 1) return -1;
  switch (dtype) {
    case PDA_F32: hipLaunchKernelGGL((adasum_combine_kernel<float>), dim3(adasum_blocks(n)), dim3(256), 0, st, (const float*)a, (const float*)b, (float*)dst, n, parts, blk, nblk); break;
    case PDA_F16: hipLaunchKernelGGL((adasum_combine_kernel<f16>), dim3(adasum_blocks(n)), dim3(256), 0, st, (const f16*)a, (const f16*)b, (f16*)dst, n, parts, blk, nblk); break;
    case PDA_BF16: hipLaunchKernelGGL((adasum_combine_kernel<bf16>), dim3(adasum_blocks(n)), dim3(256), 0, st, (const bf16*)a, (const bf16*)b, (bf16*)dst, n, parts, blk, nblk); break;
    default: return -6;
  }
  return pda_check_launch();
}
