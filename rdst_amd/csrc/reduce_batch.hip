// See reduce_batch.h.
#include "reduce_batch.h"
#include <vector>

namespace rbatch {
namespace {

constexpr int MAXJ = 8;
struct SumBatch { SumJob j[MAXJ]; int first[MAXJ + 1]; int n; };   // first[k] = first block of job k
struct FinBatch { FinJob j[MAXJ]; int first[MAXJ + 1]; int n; };

thread_local bool g_active = false;
thread_local std::vector<SumJob> g_sums;
thread_local std::vector<FinJob> g_fins;

// sum of the slabs in fixed order: 32 outputs per block, 8 slab groups x 8 loads in flight, two fixed-order levels
__global__ void __launch_bounds__(256) batched_sum_kernel(const SumBatch bt) {
  __shared__ float part[8][33];
  int k = 0;
#pragma unroll
  for (int q = 1; q < MAXJ; ++q) k += (q < bt.n && (int)blockIdx.x >= bt.first[q]) ? 1 : 0;
  const SumJob& J = bt.j[k];
  const int o = threadIdx.x & 31, sg = threadIdx.x >> 5;
  const int i = ((int)blockIdx.x - bt.first[k]) * 32 + o;
  float a = 0.f;
  if (i < J.tot) {
    // slab element of output i (MAP_DTABLE: the slab rows are [heads][T] like the output index i = h * T + t)
    for (int w0 = sg; w0 < J.nwg; w0 += 8 * 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = (w0 + 8 * u < J.nwg) ? J.slab[(int64_t)(w0 + 8 * u) * J.stride + i] : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) a += v[u];
    }
  }
  part[sg][o] = a;
  __syncthreads();
  if (sg != 0 || i >= J.tot) return;
  a = 0.f;
#pragma unroll
  for (int q = 0; q < 8; ++q) a += part[q][o];
  if (J.map == MAP_COPY) {
    J.out[i] = a;
  } else if (J.map == MAP_LINEAR) {
    const int K = J.a, Kx = J.b, n = i / Kx, kk = i - n * Kx;
    if (kk < K) { if (J.out) J.out[(int64_t)n * K + kk] = a * J.s; }
    else if (kk == K && J.out2) J.out2[n] = a * J.s;
  } else if (J.map == MAP_MLP) {
    const int C = J.a, hid = J.b, n1 = hid * (C + 1);
    if (i < n1) J.out[i] = a;
    else {
      const int i2 = i - n1, jj = i2 / C, c = i2 - jj * C;
      if (jj < hid) J.out2[(int64_t)c * hid + jj] = a;
      else J.out3[c] = a;
    }
  } else {
    const int heads = J.a, T = J.b, hh = i / T, t = i - hh * T;
    J.out[t * heads + hh] = a;
  }
}

// LayerNorm finish, one block per 8 LayerNorm channels of one job (see linear_mfma.hip: wgrad_ln_finish_kernel)
__global__ void __launch_bounds__(1024) batched_ln_finish_kernel(const FinBatch bt) {
  __shared__ float pg[128][9], pb[128][9], qg[8][9], qb[8][9];
  int kj = 0;
#pragma unroll
  for (int q = 1; q < MAXJ; ++q) kj += (q < bt.n && (int)blockIdx.x >= bt.first[q]) ? 1 : 0;
  const FinJob& J = bt.j[kj];
  const int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;
  const int blk = (int)blockIdx.x - bt.first[kj];
  const int k = blk * 8 + tx;
  const int N = J.N, K = J.K, Kx = J.Kx;
  const float s = J.s;
  const float gk = k < K ? J.gamma[k] : 0.f, bk = k < K ? J.beta[k] : 0.f;
  float ag = 0.f, ab = 0.f;
#pragma unroll 3
  for (int n = ty; n < N; n += 128) {
    const float db = J.G[(int64_t)n * Kx + K];
    if (k < K) {
      const float g = J.G[(int64_t)n * Kx + k], w = J.Wt[(int64_t)n * K + k];
      if (J.dW) J.dW[(int64_t)n * K + k] = s * fmaf(gk, g, bk * db);
      ag = fmaf(w, g, ag);
      ab = fmaf(w, db, ab);
    }
    if (blk == 0 && tx == 0 && J.dbias) J.dbias[n] = s * db;
  }
  pg[ty][tx] = ag;
  pb[ty][tx] = ab;
  __syncthreads();
  if (ty < 8) {   // two fixed-order levels: 8 partial sums of 16 rows, then their sum
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) { a += pg[ty * 16 + j][tx]; b += pb[ty * 16 + j][tx]; }
    qg[ty][tx] = a;
    qb[ty][tx] = b;
  }
  __syncthreads();
  if (ty == 0 && k < K) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { a += qg[j][tx]; b += qb[j][tx]; }
    if (J.dgamma) J.dgamma[k] = s * a;
    if (J.dbeta) J.dbeta[k] = s * b;
  }
}

int launch_sums(const SumJob* jobs, int n, hipStream_t st) {
  for (int base = 0; base < n; base += MAXJ) {
    SumBatch bt{};
    bt.n = n - base < MAXJ ? n - base : MAXJ;
    int blocks = 0;
    for (int q = 0; q < bt.n; ++q) {
      bt.j[q] = jobs[base + q];
      bt.first[q] = blocks;
      blocks += (bt.j[q].tot + 31) / 32;
    }
    bt.first[bt.n] = blocks;
    if (blocks == 0) continue;
    hipLaunchKernelGGL(batched_sum_kernel, dim3((unsigned)blocks), dim3(256), 0, st, bt);
    if (int rc = rdst_launch_status("batched_sum")) return rc;
  }
  return 0;
}
int launch_fins(const FinJob* jobs, int n, hipStream_t st) {
  for (int base = 0; base < n; base += MAXJ) {
    FinBatch bt{};
    bt.n = n - base < MAXJ ? n - base : MAXJ;
    int blocks = 0;
    for (int q = 0; q < bt.n; ++q) {
      bt.j[q] = jobs[base + q];
      bt.first[q] = blocks;
      blocks += (bt.j[q].K + 7) / 8;
    }
    bt.first[bt.n] = blocks;
    if (blocks == 0) continue;
    hipLaunchKernelGGL(batched_ln_finish_kernel, dim3((unsigned)blocks), dim3(1024), 0, st, bt);
    if (int rc = rdst_launch_status("batched_ln_finish")) return rc;
  }
  return 0;
}

}  // namespace

int sum(const SumJob& j, hipStream_t st) {
  if (g_active) { g_sums.push_back(j); return 0; }
  return launch_sums(&j, 1, st);
}
int finish(const FinJob& j, hipStream_t st) {
  if (g_active) { g_fins.push_back(j); return 0; }
  return launch_fins(&j, 1, st);
}

}  // namespace rbatch

extern "C" int rdst_reduce_batch_begin(void) {
  if (rbatch::g_active) return rdst_fail(RDST_EINVAL, "rdst_reduce_batch_begin: a batch is already open on this thread");
  rbatch::g_active = true;
  rbatch::g_sums.clear();
  rbatch::g_fins.clear();
  return 0;
}

extern "C" int rdst_reduce_batch_end(void* stream) {
  if (!rbatch::g_active) return rdst_fail(RDST_EINVAL, "rdst_reduce_batch_end: no open batch");
  rbatch::g_active = false;
  hipStream_t st = (hipStream_t)stream;
  int rc = rbatch::launch_sums(rbatch::g_sums.data(), (int)rbatch::g_sums.size(), st);
  if (!rc) rc = rbatch::launch_fins(rbatch::g_fins.data(), (int)rbatch::g_fins.size(), st);
  rbatch::g_sums.clear();
  rbatch::g_fins.clear();
  return rc;
}
