// Hooks for the MFMA fast paths of K3 (linear_mfma.hip).  Each returns RDST_ENOTSUP when the shape
// or dtype is not covered, in which case linear.hip falls back to the generic functor GEMM.
#pragma once
#include "common.h"

template <typename T>
int linear_fwd_mfma(const T* X, int64_t ldx, const float* ln_w, const float* ln_b, int in_act, const float* Wt,
                    const float* bias, const T* R, int64_t ldr, T* Y, int64_t ldy, float* stats, int64_t M, int K,
                    int N, float s, hipStream_t st);
// dA[M][K] = s * dY[M][N] @ Wt[N][K]; has_ln: write fp32 dA, else dX = dA*act'(X) (+dX)
template <typename T>
int linear_dgrad_mfma(const T* X, int64_t ldx, bool has_ln, int in_act, const float* Wt, const T* dY, int64_t lddy,
                      T* dX, int64_t lddx, const T* acc, int64_t ldacc, float* dA, int64_t M, int K, int N, float s,
                      hipStream_t st);
// dW[N][K] = s * dY^T f(X) and dbias[N] = s * colsum(dY) in one pass (either pointer may be NULL)
template <typename T>
int linear_wgrad_mfma(const T* X, int64_t ldx, const float* ln_w, const float* ln_b, const float* stats, int in_act,
                      const T* dY, int64_t lddy, float* dW, float* dbias, float* slab, int64_t M, int K, int N, float s,
                      hipStream_t st);
size_t linear_wgrad_mfma_slab_floats(int64_t M, int K, int N);
int linear_wgrad_max_wgs(int N);   // slab rows the workspace holds for a Linear of N outputs
// dgrad with the LayerNorm backward fused in (K <= 128): dX written/accumulated, d(gamma)/d(beta) partials in
// slab[*nslab][2][K] for the caller to reduce
template <typename T>
int linear_dgrad_ln_mfma(const T* X, int64_t ldx, const float* stats, const float* gamma, const float* Wt, const T* dY,
                         int64_t lddy, T* dX, int64_t lddx, const T* acc, int64_t ldacc, float* slab, int* nslab,
                         int64_t M, int K, int N, float s, hipStream_t st);
// LayerNorm-fused Linear, weight-gradient side finishing the LayerNorm parameters too: the wgrad pass runs on
// x-hat, the reduction then forms dW = s (gamma G + beta db^T), dbias = s db, d(gamma)[k] = s sum_n W[n][k] G[n][k],
// d(beta)[k] = s sum_n W[n][k] db[n].  G: N*(K+1) floats of scratch.
template <typename T>
int linear_wgrad_ln_mfma(const T* X, int64_t ldx, const float* ln_w, const float* ln_b, const float* stats, const float* Wt,
                         const T* dY, int64_t lddy, float* dW, float* dbias, float* dln_w, float* dln_b, float* slab,
                         float* G, int64_t M, int K, int N, float s, hipStream_t st);
// dgrad + LayerNorm backward producing dX only, everything register resident (K <= 128)
template <typename T>
int linear_dgrad_ln2_mfma(const T* X, int64_t ldx, const float* stats, const float* gamma, const float* Wt, const T* dY,
                          int64_t lddy, T* dX, int64_t lddx, const T* acc, int64_t ldacc, int64_t M, int K, int N, float s,
                          hipStream_t st);
// finish of a LayerNorm-fused weight gradient from G (N, K+1) = [dY^T x-hat | colsum(dY)] (see linear_wgrad_ln_mfma)
int wgrad_ln_finish_launch(const float* G, const float* Wt, const float* ln_w, const float* ln_b, int N, int K, float s,
                           float* dW, float* dbias, float* dln_w, float* dln_b, hipStream_t st);
int wgrad_sum_launch(const float* slab, int nwg, int tot, float* G, hipStream_t st);
int wgrad_reduce_launch(const float* slab, int nwg, int N, int K, float s, float* dW, float* dbias, hipStream_t st);
// LayerNorm + Linear backward (dX, dW, dbias, d(gamma), d(beta)) in ONE pass over (x, dY): mlp_mfma.hip
int linear_ln_bwd_fused_bf16(const bf16* X, int64_t ldx, const float* ln_w, const float* ln_b, const float* stats,
                             const float* Wt, const bf16* dY, int64_t lddy, bf16* dX, int64_t lddx, const bf16* acc,
                             int64_t ldacc, float* dW, float* dbias, float* dln_w, float* dln_b, float* slab, float* G,
                             int64_t M, int K, int N, float s, hipStream_t st, const bf16* acc2 = nullptr, int64_t ldacc2 = 0);

// streaming forward for the E1 shapes, bf16 (lin3_mfma.hip); RDST_ENOTSUP for everything else.
// wpack: lin3_pack_bytes(K, N) bytes of 16-byte aligned device scratch (NULL -> RDST_ENOTSUP).
size_t lin3_pack_bytes(int K, int N);
int lin3_fwd_bf16(const bf16* X, int64_t ldx, const float* ln_w, const float* ln_b, int in_act, const float* Wt, const float* bias,
                  const bf16* R, int64_t ldr, bf16* Y, int64_t ldy, float* stats, int64_t M, int K, int N, float s, void* wpack,
                  bool prepacked, hipStream_t st);
int lin3_pack_launch(const float* W, const float* gamma, const float* beta, const float* bias, void* out, int N, int K, float s,
                     hipStream_t st);
// fused Mlp forward on the same skeleton (mlp3_mfma.hip): wpack = mlp3_pack_bytes(C, hid) bytes = [fc1 image][fc2 image]
size_t mlp3_pack_bytes(int C, int hid);
int mlp3_fwd_bf16(const bf16* X, int64_t ldx, const float* ln_w, const float* ln_b, const float* W1, const float* b1, const float* W2,
                  const float* b2, bf16* Y, int64_t ldy, float* stats, int64_t M, int C, int hid, void* wpack, bool prepacked,
                  hipStream_t st);

// the same streaming forward on fp32 rows in the RDST_F32X3 arithmetic (lin3x_mfma.hip); wpack: lin3x_pack_bytes(K, N) bytes (pack.h)
int lin3x_kind(int K, int N, bool ln, bool res, int in_act);   // 0 = not covered
int lin3x_fwd_f32(const float* X, int64_t ldx, const float* ln_w, const float* ln_b, int in_act, const float* Wt, const float* bias,
                  const float* R, int64_t ldr, float* Y, int64_t ldy, float* stats, int64_t M, int K, int N, float s, void* wpack,
                  bool prepacked, hipStream_t st);
int lin3x_pack_launch(const float* W, const float* gamma, const float* beta, const float* bias, void* out, int N, int K, float s,
                      hipStream_t st);
// one-pass Linear backward on fp32 rows in the RDST_F32X3 arithmetic (lnlin3x_mfma.hip): dX (+ LayerNorm backward / GELU' / addends), dW,
// dbias, d(gamma), d(beta); slab: linear_wgrad_mfma_slab_floats(M, K, N) floats, G: N (K + 1) floats (LayerNorm only)
int lnlin3x_bwd_kind(int K, int N, bool ln, int in_act);   // 0 = not covered
int lnlin3x_bwd_f32(const float* X, int64_t ldx, const float* ln_w, const float* ln_b, const float* stats, int in_act, const float* Wt,
                    const float* dY, int64_t lddy, float* dX, int64_t lddx, const float* acc, int64_t ldacc, const float* acc2,
                    int64_t ldacc2, float* dW, float* dbias, float* dln_w, float* dln_b, float* slab, float* G, int64_t M, int K, int N,
                    float s, hipStream_t st);
