// See reduce_batch.h.
#include "reduce_batch.h"
#include <vector>

namespace rbatch {
namespace {

constexpr int MAXJ = 16;   // jobs per launch: a DenseSTLayer's backward (2 Swin blocks + its tail Linear) queues 11 sums and 5 finishes
struct SumBatch { SumJob j[MAXJ]; int first[MAXJ + 1]; int n; };   // first[k] = first block of job k
struct FinBatch { FinJob j[MAXJ]; int first[MAXJ + 1]; int n; };

thread_local bool g_active = false;
thread_local std::vector<SumJob> g_sums;
thread_local std::vector<FinJob> g_fins;

// one summed element -> its destination(s)
__device__ __forceinline__ void emit(const SumJob& J, int n, int c, int i, float a) {
  if (J.map == MAP_COPY) {
    J.out[J.g4 ? n * J.b2 + c : i] = a;
  } else if (J.map == MAP_LINEAR) {
    const int K = J.a, Kx = J.b;
    if (!J.g4) { n = i / Kx; c = i - n * Kx; }
    if (c < K) { if (J.out) J.out[(int64_t)n * K + c] = a * J.s; }
    else if (c == K && J.out2) J.out2[n] = a * J.s;
  } else if (J.map == MAP_MLP) {
    const int C = J.a, hid = J.b, n1 = hid * (C + 1);
    if (i < n1) J.out[i] = a;
    else {
      const int i2 = i - n1, jj = i2 / C, cc = i2 - jj * C;
      if (jj < hid) J.out2[(int64_t)cc * hid + jj] = a;
      else J.out3[cc] = a;
    }
  } else if (J.map == MAP_T) {
    const int rows = J.a;   // n = j, c = channel
    if (n < rows) J.out[(int64_t)c * rows + n] = a;
    else if (n == rows) J.out2[c] = a;
  } else {
    const int heads = J.a, T = J.b, hh = i / T, t = i - hh * T;
    J.out[t * heads + hh] = a;
  }
}

// sum of the slabs in fixed order: 32 outputs (fp32) or 32 groups of 4 (G4) per block, 8 slab groups x 8 loads in flight,
// two fixed-order levels
__global__ void __launch_bounds__(256) batched_sum_kernel(const SumBatch bt) {
  __shared__ float part[4][8][33];
  int k = 0;
#pragma unroll
  for (int q = 1; q < MAXJ; ++q) k += (q < bt.n && (int)blockIdx.x >= bt.first[q]) ? 1 : 0;
  const SumJob& J = bt.j[k];
  const int o = threadIdx.x & 31, sg = threadIdx.x >> 5;
  const int i = ((int)blockIdx.x - bt.first[k]) * 32 + o;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (i < J.tot) {
    if (J.g4) {
      const uint2* sl = reinterpret_cast<const uint2*>(J.slab);
      for (int w0 = sg; w0 < J.nwg; w0 += 8 * 8) {
        uint2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = (w0 + 8 * u < J.nwg) ? sl[(int64_t)(w0 + 8 * u) * J.stride + i] : make_uint2(0u, 0u);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          a0 += __uint_as_float(v[u].x << 16); a1 += __uint_as_float(v[u].x & 0xffff0000u);
          a2 += __uint_as_float(v[u].y << 16); a3 += __uint_as_float(v[u].y & 0xffff0000u);
        }
      }
    } else {
      for (int w0 = sg; w0 < J.nwg; w0 += 8 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = (w0 + 8 * u < J.nwg) ? J.slab[(int64_t)(w0 + 8 * u) * J.stride + i] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) a0 += v[u];
      }
    }
  }
  part[0][sg][o] = a0; part[1][sg][o] = a1; part[2][sg][o] = a2; part[3][sg][o] = a3;
  __syncthreads();
  if (sg != 0 || i >= J.tot) return;
  float r[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int q = 0; q < 8; ++q) r[e] += part[e][q][o];
  if (J.g4) {
    const int W = J.b2, ng = i / W, c = i - ng * W;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (4 * ng + e < J.a2) emit(J, 4 * ng + e, c, 0, r[e]);
  } else {
    emit(J, 0, 0, i, r[0]);
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

extern "C" int rdst_reduce_batch_abort(void) {
  // drop whatever is queued WITHOUT running it (the workspaces and outputs of a failed backward may be gone)
  rbatch::g_active = false;
  rbatch::g_sums.clear();
  rbatch::g_fins.clear();
  return 0;
}
