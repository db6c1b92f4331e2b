// Generic functor-driven tile GEMM on the vector ALUs (fp32 math): the shape-agnostic baseline every
// GEMM-shaped op of the path (Linear fwd/dgrad/wgrad, conv fwd/dgrad/wgrad) can be expressed with,
// and the fallback for shapes the MFMA kernels do not cover (Cin = 1 or 3, Cout = 1, odd K ...).
//
//   C[m][n] = sum_{k in [z*klen, (z+1)*klen)} A(m,k) * B(k,n)        z = blockIdx.z (split-K)
//
// A and B are device functors (they fold LayerNorm, activations, im2col addressing, pixel-shuffle
// addressing and transposes into the load), EP is the epilogue functor (bias, scale, residual,
// activation gradient, split-K slab store ...).  64x64 tile per 256-thread workgroup, 4x4 outputs per
// thread, K staged 16 at a time through LDS.
#pragma once
#include "common.h"

template <class LA, class LB, class EP>
__global__ void __launch_bounds__(256)
gemm_valu_kernel(LA la, LB lb, EP ep, int64_t M, int N, int64_t K, int64_t klen) {
  __shared__ __attribute__((aligned(16))) float As[16][68];
  __shared__ __attribute__((aligned(16))) float Bs[16][68];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int64_t m0 = (int64_t)blockIdx.x * 64;
  const int n0 = blockIdx.y * 64;
  const int64_t kb = (int64_t)blockIdx.z * klen;
  const int64_t ke = (kb + klen < K) ? kb + klen : K;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

  for (int64_t k0 = kb; k0 < ke; k0 += 16) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int idx = tid + 256 * r;
      int kk, mm;
      if (LA::kFast) { kk = idx & 15; mm = idx >> 4; } else { mm = idx & 63; kk = idx >> 6; }
      {
        const int64_t m = m0 + mm, k = k0 + kk;
        As[kk][mm] = (m < M && k < ke) ? la(m, k) : 0.f;
      }
      if (LB::kFast) { kk = idx & 15; mm = idx >> 4; } else { mm = idx & 63; kk = idx >> 6; }
      {
        const int n = n0 + mm;
        const int64_t k = k0 + kk;
        Bs[kk][mm] = (n < N && k < ke) ? lb(k, n) : 0.f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const float4 a = *reinterpret_cast<const float4*>(&As[kk][ty * 4]);
      const float4 b = *reinterpret_cast<const float4*>(&Bs[kk][tx * 4]);
      const float av[4] = {a.x, a.y, a.z, a.w};
      const float bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t m = m0 + ty * 4 + i;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + tx * 4 + j;
      if (n < N) ep(m, n, acc[i][j], (int)blockIdx.z);
    }
  }
}

template <class LA, class LB, class EP>
static inline int gemm_valu_launch(const LA& la, const LB& lb, const EP& ep, int64_t M, int N, int64_t K, int splits,
                                   hipStream_t st, const char* what) {
  if (M <= 0 || N <= 0) return 0;
  int64_t klen = (K + splits - 1) / splits;
  klen = ((klen + 15) / 16) * 16;
  if (klen <= 0) klen = 16;
  const int64_t z = (K + klen - 1) / klen;
  dim3 grid((unsigned)((M + 63) / 64), (unsigned)((N + 63) / 64), (unsigned)(z > 0 ? z : 1));
  hipLaunchKernelGGL((gemm_valu_kernel<LA, LB, EP>), grid, dim3(256), 0, st, la, lb, ep, M, N, K, klen);
  return rdst_launch_status(what);
}

// how many split-K slices gemm_valu_launch will actually use for (K, splits)
static inline int gemm_valu_splits(int64_t K, int splits) {
  int64_t klen = (K + splits - 1) / splits;
  klen = ((klen + 15) / 16) * 16;
  if (klen <= 0) klen = 16;
  const int64_t z = (K + klen - 1) / klen;
  return (int)(z > 0 ? z : 1);
}

// out[i] = sum_s slab[s*n + i]  (fixed order => bitwise reproducible)
__global__ void __launch_bounds__(256) slab_reduce_kernel(const float* __restrict__ slab, float* __restrict__ out,
                                                          int S, int64_t n);
int slab_reduce(const float* slab, float* out, int S, int64_t n, hipStream_t st);
int slab_reduce2(const float* slab, float* out_a, float* out_b, int S, int K, hipStream_t st);

// column sums of a functor matrix F(m, n), m < M, n < N <= 1024: out[n] = sum_m F(m,n).
// Two passes through a [blocks][N] slab for a reproducible result.
template <class F>
__global__ void __launch_bounds__(256) colsum_kernel(F f, int64_t M, int N, float* __restrict__ slab) {
  // thread t handles column (t % NC) for rows striding by 256/NC ... keep it simple: each thread owns
  // columns n = tid, tid+256, ... and walks this block's row range.
  const int64_t rows_per = (M + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per;
  const int64_t r1 = (r0 + rows_per < M) ? r0 + rows_per : M;
  for (int n = threadIdx.x; n < N; n += 256) {
    float s = 0.f;
    for (int64_t m = r0; m < r1; ++m) s += f(m, n);
    slab[(int64_t)blockIdx.x * N + n] = s;
  }
}

template <class F>
static inline int colsum_launch(const F& f, int64_t M, int N, float* slab, int blocks, float* out, hipStream_t st,
                                const char* what) {
  hipLaunchKernelGGL((colsum_kernel<F>), dim3(blocks), dim3(256), 0, st, f, M, N, slab);
  if (int rc = rdst_launch_status(what)) return rc;
  return slab_reduce(slab, out, blocks, N, st);
}
