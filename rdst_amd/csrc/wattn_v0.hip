// K1/K2 generic window-attention kernels ("v0"): one workgroup per (window, head), VALU math in
// fp32, any window size / head dim the MFMA kernels do not cover.  Correctness baseline and
// fallback; the tuned gfx950 kernels live in wattn_mfma.hip.
//
// Reference sequence replaced: networks/swin_transformer_sr.py:244-267 + :117-138 (see
// include/rdst_hip.h, rdst_wattn_fwd).
#include "common.h"
#include "wattn.h"

namespace {

// smem carve (floats): Ks[N*D] Vs[N*D] tab[T] | ints: trow[N] reg[N]
template <typename T, int D>
__global__ void __launch_bounds__(256)
wattn_fwd_v0(const T* __restrict__ qkv, int64_t ld, const float* __restrict__ table,
             T* __restrict__ out, int64_t ldo, WinGeom g, float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int N = g.N, TT = g.T, ws = g.ws;
  float* Ks = reinterpret_cast<float*>(smem_raw);
  float* Vs = Ks + N * D;
  float* tab = Vs + N * D;
  int* trow = reinterpret_cast<int*>(tab + TT);
  int* reg = trow + N;

  const int tid = threadIdx.x, bd = blockDim.x;
  const int head = blockIdx.x % g.heads;
  const int win = blockIdx.x / g.heads;
  const int nW = g.nWh * g.nWw;
  const int b = win / nW, wi = win - b * nW;
  const int wr = wi / g.nWw, wc = wi - wr * g.nWw;

  for (int t = tid; t < N; t += bd) {
    trow[t] = (int)win_token(b, wr, wc, t, g);
    reg[t] = win_region(wr, wc, t, g);
  }
  for (int t = tid; t < TT; t += bd) tab[t] = table[t * g.heads + head];
  __syncthreads();
  for (int idx = tid; idx < N * D; idx += bd) {
    const int j = idx / D, e = idx - j * D;
    const T* row = qkv + (int64_t)trow[j] * ld + head * D + e;
    Ks[idx] = to_f32<T>(row[g.C]);
    Vs[idx] = to_f32<T>(row[2 * g.C]);
  }
  __syncthreads();

  const int tw = 2 * ws - 1;
  const bool drop = g.pdrop > 0.f;
  const unsigned long long seed = drop ? *g.seed : 0ull;
  const float rkeep = drop ? 1.0f / (1.0f - g.pdrop) : 1.0f;
  for (int i = tid; i < N; i += bd) {
    float q[D];
    const T* qrow = qkv + (int64_t)trow[i] * ld + head * D;
#pragma unroll
    for (int e = 0; e < D; ++e) q[e] = to_f32<T>(qrow[e]) * scale;
    const int yi = i / ws, xi = i - yi * ws, ri = reg[i];
    const float* mrow = g.mask ? g.mask + ((int64_t)(win % g.mask_nw) * N + i) * N : nullptr;
    const int base = (yi + ws - 1) * tw + xi + ws - 1;
    float m = -INFINITY;
    for (int yj = 0, j = 0; yj < ws; ++yj)
      for (int xj = 0; xj < ws; ++xj, ++j) {
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < D; ++e) s = fmaf(q[e], Ks[j * D + e], s);
        s += tab[base - yj * tw - xj];
        if (mrow) s += mrow[j]; else if (g.shift > 0 && reg[j] != ri) s += -100.0f;
        m = fmaxf(m, s);
      }
    float l = 0.f, acc[D];
#pragma unroll
    for (int e = 0; e < D; ++e) acc[e] = 0.f;
    for (int yj = 0, j = 0; yj < ws; ++yj)
      for (int xj = 0; xj < ws; ++xj, ++j) {
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < D; ++e) s = fmaf(q[e], Ks[j * D + e], s);
        s += tab[base - yj * tw - xj];
        if (mrow) s += mrow[j]; else if (g.shift > 0 && reg[j] != ri) s += -100.0f;
        const float p = expf(s - m);
        l += p;   // the softmax is normalised over ALL keys; dropout acts on the normalised weights
        const float pm = drop ? p * wattn_drop_mul(seed, blockIdx.x, i, j, N, g.pdrop, rkeep) : p;
#pragma unroll
        for (int e = 0; e < D; ++e) acc[e] = fmaf(pm, Vs[j * D + e], acc[e]);
      }
    const float inv = 1.0f / l;
    T* orow = out + (int64_t)trow[i] * ldo + head * D;
#pragma unroll
    for (int e = 0; e < D; ++e) orow[e] = from_f32<T>(acc[e] * inv);
  }
}

// smem carve (floats): Qs Ks Vs dOs [N*D each] lse[N] delta[N] tab[T] dtab[T] | ints trow[N] reg[N]
template <typename T, int D>
__global__ void __launch_bounds__(256)
wattn_bwd_v0(const T* __restrict__ qkv, int64_t ld, const float* __restrict__ table,
             const T* __restrict__ dout, int64_t ldd, T* __restrict__ dqkv, int64_t ldq,
             float* __restrict__ slab, WinGeom g, float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int N = g.N, TT = g.T, ws = g.ws;
  float* Qs = reinterpret_cast<float*>(smem_raw);
  float* Ks = Qs + N * D;
  float* Vs = Ks + N * D;
  float* dOs = Vs + N * D;
  float* lse = dOs + N * D;
  float* delta = lse + N;
  float* tab = delta + N;
  float* dtab = tab + TT;
  int* trow = reinterpret_cast<int*>(dtab + TT);
  int* reg = trow + N;

  const int tid = threadIdx.x, bd = blockDim.x;
  const int head = blockIdx.x % g.heads;
  const int win = blockIdx.x / g.heads;
  const int nW = g.nWh * g.nWw;
  const int b = win / nW, wi = win - b * nW;
  const int wr = wi / g.nWw, wc = wi - wr * g.nWw;

  for (int t = tid; t < N; t += bd) {
    trow[t] = (int)win_token(b, wr, wc, t, g);
    reg[t] = win_region(wr, wc, t, g);
  }
  for (int t = tid; t < TT; t += bd) {
    tab[t] = table[t * g.heads + head];
    dtab[t] = 0.f;
  }
  __syncthreads();
  for (int idx = tid; idx < N * D; idx += bd) {
    const int j = idx / D, e = idx - j * D;
    const T* row = qkv + (int64_t)trow[j] * ld + head * D + e;
    Qs[idx] = to_f32<T>(row[0]) * scale;
    Ks[idx] = to_f32<T>(row[g.C]);
    Vs[idx] = to_f32<T>(row[2 * g.C]);
    dOs[idx] = to_f32<T>(dout[(int64_t)trow[j] * ldd + head * D + e]);
  }
  __syncthreads();

  const int tw = 2 * ws - 1;
  // attention dropout: out_i = sum_j a_ij m_ij v_j with a = softmax, m = 0 or 1 / (1 - p).  Then dV_j = sum_i a_ij m_ij dO_i,
  // dS_ij = a_ij (m_ij dP_ij - delta_i) with dP = dO.V^T and delta_i = sum_k a_ik m_ik dP_ik = dO_i . out_i as without dropout
  const bool drop = g.pdrop > 0.f;
  const unsigned long long seed = drop ? *g.seed : 0ull;
  const float rkeep = drop ? 1.0f / (1.0f - g.pdrop) : 1.0f;
  // ---- pass A: one lane per query row: lse, delta, dQ, d(table) -------------------------------
  for (int i = tid; i < N; i += bd) {
    float q[D], dO[D];
#pragma unroll
    for (int e = 0; e < D; ++e) { q[e] = Qs[i * D + e]; dO[e] = dOs[i * D + e]; }
    const int yi = i / ws, xi = i - yi * ws, ri = reg[i];
    const float* mrow = g.mask ? g.mask + ((int64_t)(win % g.mask_nw) * N + i) * N : nullptr;
    const int base = (yi + ws - 1) * tw + xi + ws - 1;
    float m = -INFINITY;
    for (int yj = 0, j = 0; yj < ws; ++yj)
      for (int xj = 0; xj < ws; ++xj, ++j) {
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < D; ++e) s = fmaf(q[e], Ks[j * D + e], s);
        s += tab[base - yj * tw - xj];
        if (mrow) s += mrow[j]; else if (g.shift > 0 && reg[j] != ri) s += -100.0f;
        m = fmaxf(m, s);
      }
    float l = 0.f, acc[D];
#pragma unroll
    for (int e = 0; e < D; ++e) acc[e] = 0.f;
    for (int yj = 0, j = 0; yj < ws; ++yj)
      for (int xj = 0; xj < ws; ++xj, ++j) {
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < D; ++e) s = fmaf(q[e], Ks[j * D + e], s);
        s += tab[base - yj * tw - xj];
        if (mrow) s += mrow[j]; else if (g.shift > 0 && reg[j] != ri) s += -100.0f;
        const float p = expf(s - m);
        l += p;
        const float pm = drop ? p * wattn_drop_mul(seed, blockIdx.x, i, j, N, g.pdrop, rkeep) : p;
#pragma unroll
        for (int e = 0; e < D; ++e) acc[e] = fmaf(pm, Vs[j * D + e], acc[e]);
      }
    float dl = 0.f;
#pragma unroll
    for (int e = 0; e < D; ++e) dl = fmaf(dO[e], acc[e], dl);
    dl /= l;                       // delta_i = dO_i . O_i = sum_j P_ij dP_ij
    const float ls = m + logf(l);  // log-sum-exp of row i
    lse[i] = ls;
    delta[i] = dl;
    float dq[D];
#pragma unroll
    for (int e = 0; e < D; ++e) dq[e] = 0.f;
    for (int yj = 0, j = 0; yj < ws; ++yj)
      for (int xj = 0; xj < ws; ++xj, ++j) {
        float s = 0.f, dp = 0.f;
#pragma unroll
        for (int e = 0; e < D; ++e) {
          s = fmaf(q[e], Ks[j * D + e], s);
          dp = fmaf(dO[e], Vs[j * D + e], dp);
        }
        const int ti = base - yj * tw - xj;
        s += tab[ti];
        if (mrow) s += mrow[j]; else if (g.shift > 0 && reg[j] != ri) s += -100.0f;
        const float p = expf(s - ls);
        if (drop) dp *= wattn_drop_mul(seed, blockIdx.x, i, j, N, g.pdrop, rkeep);
        const float ds = p * (dp - dl);
#pragma unroll
        for (int e = 0; e < D; ++e) dq[e] = fmaf(ds, Ks[j * D + e], dq[e]);
        atomicAdd(&dtab[ti], ds);
      }
    T* qrow = dqkv + (int64_t)trow[i] * ldq + head * D;
#pragma unroll
    for (int e = 0; e < D; ++e) qrow[e] = from_f32<T>(dq[e] * scale);
  }
  __syncthreads();
  // ---- pass B: one lane per key row: dK, dV -----------------------------------------------------
  for (int j = tid; j < N; j += bd) {
    float k[D], v[D], dk[D], dv[D];
#pragma unroll
    for (int e = 0; e < D; ++e) { k[e] = Ks[j * D + e]; v[e] = Vs[j * D + e]; dk[e] = 0.f; dv[e] = 0.f; }
    const int yj = j / ws, xj = j - yj * ws, rj = reg[j];
    const float* mcol = g.mask ? g.mask + (int64_t)(win % g.mask_nw) * N * N + j : nullptr;
    const int base = (ws - 1 - yj) * tw + ws - 1 - xj;
    for (int yi = 0, i = 0; yi < ws; ++yi)
      for (int xi = 0; xi < ws; ++xi, ++i) {
        float s = 0.f, dp = 0.f;
#pragma unroll
        for (int e = 0; e < D; ++e) {
          s = fmaf(Qs[i * D + e], k[e], s);
          dp = fmaf(dOs[i * D + e], v[e], dp);
        }
        s += tab[base + yi * tw + xi];
        if (mcol) s += mcol[(int64_t)i * N]; else if (g.shift > 0 && reg[i] != rj) s += -100.0f;
        const float p = expf(s - lse[i]);
        const float mm = drop ? wattn_drop_mul(seed, blockIdx.x, i, j, N, g.pdrop, rkeep) : 1.0f;
        const float ds = p * (mm * dp - delta[i]);
        const float pm = p * mm;
#pragma unroll
        for (int e = 0; e < D; ++e) {
          dv[e] = fmaf(pm, dOs[i * D + e], dv[e]);
          dk[e] = fmaf(ds, Qs[i * D + e], dk[e]);  // Qs already carries `scale`
        }
      }
    T* krow = dqkv + (int64_t)trow[j] * ldq + g.C + head * D;
    T* vrow = krow + g.C;
#pragma unroll
    for (int e = 0; e < D; ++e) { krow[e] = from_f32<T>(dk[e]); vrow[e] = from_f32<T>(dv[e]); }
  }
  __syncthreads();
  float* my = slab + (int64_t)blockIdx.x * TT;
  for (int t = tid; t < TT; t += bd) my[t] = dtab[t];
}

template <typename T>
int launch_fwd_d(int D, const T* qkv, int64_t ld, const float* table, T* out, int64_t ldo, const WinGeom& g,
                 float scale, hipStream_t st) {
  const int N = g.N;
  const size_t smem = (size_t)(2 * N * D + g.T) * 4 + (size_t)2 * N * 4;
  const int bd = N >= 256 ? 256 : ((N + 63) / 64) * 64;
  const dim3 grid((unsigned)((int64_t)g.B * g.nWh * g.nWw * g.heads));
  if (smem > 160 * 1024) return rdst_fail(RDST_ENOTSUP, "rdst_wattn_fwd: window %d x head dim %d exceeds LDS", g.ws, D);
#define RDST_FWD_CASE(DD)                                                                        \
  case DD: {                                                                                     \
    auto kern = wattn_fwd_v0<T, DD>;                                                             \
    if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
    hipLaunchKernelGGL(kern, grid, dim3(bd), smem, st, qkv, ld, table, out, ldo, g, scale);      \
  } break;
  switch (D) {
    RDST_FWD_CASE(4) RDST_FWD_CASE(5) RDST_FWD_CASE(8) RDST_FWD_CASE(10) RDST_FWD_CASE(12) RDST_FWD_CASE(15)
    RDST_FWD_CASE(16) RDST_FWD_CASE(20) RDST_FWD_CASE(24) RDST_FWD_CASE(30) RDST_FWD_CASE(32)
    default:
      return rdst_fail(RDST_ENOTSUP, "rdst_wattn_fwd: head dim %d not supported by the generic kernel", D);
  }
#undef RDST_FWD_CASE
  return rdst_launch_status("wattn_fwd_v0");
}

template <typename T>
int launch_bwd_d(int D, const T* qkv, int64_t ld, const float* table, const T* dout, int64_t ldd, T* dqkv,
                 int64_t ldq, float* slab, const WinGeom& g, float scale, hipStream_t st) {
  const int N = g.N;
  const size_t smem = (size_t)(4 * N * D + 2 * N + 2 * g.T) * 4 + (size_t)2 * N * 4;
  if (smem > 160 * 1024) return rdst_fail(RDST_ENOTSUP, "rdst_wattn_bwd: window %d x head dim %d exceeds LDS", g.ws, D);
  const int bd = N >= 256 ? 256 : ((N + 63) / 64) * 64;
  const dim3 grid((unsigned)((int64_t)g.B * g.nWh * g.nWw * g.heads));
#define RDST_BWD_CASE(DD)                                                                         \
  case DD: {                                                                                      \
    auto kern = wattn_bwd_v0<T, DD>;                                                              \
    if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
    hipLaunchKernelGGL(kern, grid, dim3(bd), smem, st, qkv, ld, table, dout, ldd, dqkv, ldq, slab, g, scale); \
  } break;
  switch (D) {
    RDST_BWD_CASE(4) RDST_BWD_CASE(5) RDST_BWD_CASE(8) RDST_BWD_CASE(10) RDST_BWD_CASE(12) RDST_BWD_CASE(15)
    RDST_BWD_CASE(16) RDST_BWD_CASE(20) RDST_BWD_CASE(24) RDST_BWD_CASE(30) RDST_BWD_CASE(32)
    default:
      return rdst_fail(RDST_ENOTSUP, "rdst_wattn_bwd: head dim %d not supported by the generic kernel", D);
  }
#undef RDST_BWD_CASE
  return rdst_launch_status("wattn_bwd_v0");
}

}  // namespace

namespace {
__global__ void wattn_drop_mask_kernel(float* out, int N, float pdrop, const unsigned long long* seedp) {
  const unsigned long long seed = *seedp;
  const float rkeep = 1.0f / (1.0f - pdrop);
  float* my = out + (size_t)blockIdx.x * N * N;
  for (int idx = threadIdx.x; idx < N * N; idx += blockDim.x) {
    const int i = idx / N, j = idx - i * N;
    my[idx] = wattn_drop_mul(seed, blockIdx.x, i, j, N, pdrop, rkeep);
  }
}
}  // namespace

int wattn_drop_mask(float* out, int64_t nblk, int N, float pdrop, const unsigned long long* seed, hipStream_t st) {
  hipLaunchKernelGGL(wattn_drop_mask_kernel, dim3((unsigned)nblk), dim3(256), 0, st, out, N, pdrop, seed);
  return rdst_launch_status("wattn_drop_mask");
}

int wattn_fwd_generic(const void* qkv, int64_t ld, const float* table, void* out, int64_t ldo, const WinGeom& g,
                      float scale, int dtype, hipStream_t st) {
  const int D = g.C / g.heads;
  if (dtype == RDST_F32)
    return launch_fwd_d<float>(D, (const float*)qkv, ld, table, (float*)out, ldo, g, scale, st);
  return launch_fwd_d<bf16>(D, (const bf16*)qkv, ld, table, (bf16*)out, ldo, g, scale, st);
}

int wattn_bwd_generic(const void* qkv, int64_t ld, const float* table, const void* dout, int64_t ldd, void* dqkv,
                      int64_t ldq, float* slab, const WinGeom& g, float scale, int dtype, hipStream_t st) {
  const int D = g.C / g.heads;
  if (dtype == RDST_F32)
    return launch_bwd_d<float>(D, (const float*)qkv, ld, table, (const float*)dout, ldd, (float*)dqkv, ldq, slab, g,
                               scale, st);
  return launch_bwd_d<bf16>(D, (const bf16*)qkv, ld, table, (const bf16*)dout, ldd, (bf16*)dqkv, ldq, slab, g, scale,
                            st);
}
