// Window geometry shared by the window-attention kernels (K1/K2).
#pragma once
#include "common.h"

struct WinGeom {
  int B, H, W, C, heads, ws, shift;
  int nWh, nWw;  // windows per column / row
  int N;         // tokens per window = ws*ws
  int T;         // relative-position table rows = (2ws-1)^2
  const float* mask;  // optional explicit additive mask (mask_nw, N, N), else NULL
  int mask_nw;
  // attention dropout (swin_transformer_sr.py:136, training only): probability and the DEVICE address of the 64-bit seed
  // of this call; 0 / NULL everywhere but in rdst_wattn_{fwd,bwd}_drop (the generic kernels of wattn_v0.hip apply it)
  float pdrop;
  const unsigned long long* seed;
};

// Multiplier of attention weight (query i, key j) of workgroup `blk` = (window, head) under dropout: 0 with probability
// pdrop, else 1 / (1 - pdrop); a pure function of (seed, blk, i, j) (splitmix64 of a counter), so the forward, both passes of
// the backward and rdst_wattn_drop_mask see the same mask without storing it.
__device__ __forceinline__ float wattn_drop_mul(unsigned long long seed, unsigned blk, int i, int j, int N, float pdrop,
                                                float rkeep) {
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (((unsigned long long)blk * N + i) * N + j + 1ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  const float u = (float)(unsigned)(z >> 40) * (1.0f / 16777216.0f);   // 24 uniform bits
  return u >= pdrop ? rkeep : 0.f;
}

// Row (token) index in the (B*H*W) activation of token t of window (wr,wc) of image b.
// The cyclic shift of networks/swin_transformer_sr.py:244-247 / :264-267 is folded in here:
// shifted[r][c] = x[(r+shift)%H][(c+shift)%W], and the result goes back to the same place.
__device__ __forceinline__ int64_t win_token(int b, int wr, int wc, int t, const WinGeom& g) {
  const int y = t / g.ws, x = t - y * g.ws;
  int r = wr * g.ws + y + g.shift;
  if (r >= g.H) r -= g.H;
  int c = wc * g.ws + x + g.shift;
  if (c >= g.W) c -= g.W;
  return ((int64_t)b * g.H + r) * g.W + c;
}

// Same for 8x8 windows (shifts instead of divisions; the MFMA kernels only handle ws == 8).
__device__ __forceinline__ int64_t win_token8(int b, int wr, int wc, int t, const WinGeom& g) {
  int r = wr * 8 + (t >> 3) + g.shift;
  if (r >= g.H) r -= g.H;
  int c = wc * 8 + (t & 7) + g.shift;
  if (c >= g.W) c -= g.W;
  return ((int64_t)b * g.H + r) * g.W + c;
}

// Region id (0..8) of a token of the SHIFTED image: networks/swin_transformer_sr.py:215-225.
// Tokens of one window with different ids are masked with -100 (:227-230).
__device__ __forceinline__ int win_region(int wr, int wc, int t, const WinGeom& g) {
  if (g.shift == 0) return 0;
  const int y = t / g.ws, x = t - y * g.ws;
  const int r = wr * g.ws + y, c = wc * g.ws + x;
  const int rr = r < g.H - g.ws ? 0 : (r < g.H - g.shift ? 1 : 2);
  const int cr = c < g.W - g.ws ? 0 : (c < g.W - g.shift ? 1 : 2);
  return rr * 3 + cr;
}

int wattn_drop_mask(float* out, int64_t nblk, int N, float pdrop, const unsigned long long* seed, hipStream_t st);
int wattn_fwd_generic(const void* qkv, int64_t ld, const float* table, void* out, int64_t ldo, const WinGeom& g,
                      float scale, int dtype, hipStream_t st);
int wattn_bwd_generic(const void* qkv, int64_t ld, const float* table, const void* dout, int64_t ldd, void* dqkv,
                      int64_t ldq, float* slab, const WinGeom& g, float scale, int dtype, hipStream_t st);

// gfx950 matrix-core kernels (wattn_mfma.hip): RDST_ENOTSUP when the shape is not covered
int wattn_fwd_mfma(const void* qkv, int64_t ld, const float* table, void* out, int64_t ldo, const WinGeom& g,
                   float scale, int dtype, hipStream_t st);
// K1 specialised for bf16, ws 8, 6 heads of dim 10/15/20 (wattn_mfma_hd.hip); RDST_ENOTSUP otherwise
int wattn_fwd_mfma_hd(const void* qkv, int64_t ld, const float* table, void* out, int64_t ldo, const WinGeom& g,
                      float scale, hipStream_t st);
// 16x16 windows, bf16, 6 heads of dim 10/15/20 (wattn16_mfma.hip); RDST_ENOTSUP otherwise
// nlse (optional): [B H W][6] floats, the forward's row statistics -(scale2 . max + log2 sum); o / ldo / nlse (optional, all
// or none): the forward's output rows and statistics — with them the backward runs its streaming first pass (round 5)
int wattn16_fwd_mfma(const void* qkv, int64_t ld, const float* table, void* out, int64_t ldo, const WinGeom& g,
                     float scale, hipStream_t st, float* nlse = nullptr);
int wattn16_bwd_mfma(const void* qkv, int64_t ld, const float* table, const void* dout, int64_t ldd, void* dqkv,
                     int64_t ldq, float* slab, int slab_rows, const WinGeom& g, float scale, int* nslab, hipStream_t st,
                     const void* o = nullptr, int64_t ldo = 0, const float* nlse = nullptr);
// 16x16 windows in exact fp32 on the matrix cores (wattn16_f32.hip); RDST_ENOTSUP otherwise
int wattn16_fwd_f32(const float* qkv, int64_t ld, const float* table, float* out, int64_t ldo, const WinGeom& g, float scale,
                    hipStream_t st);
int wattn16_bwd_f32(const float* qkv, int64_t ld, const float* table, const float* dout, int64_t ldd, float* dqkv, int64_t ldq,
                    float* slab, int slab_rows, const WinGeom& g, float scale, int* nslab, hipStream_t st);
int wattn_bwd_pair(const void* qkv, int64_t ld, const float* table, const void* dout, int64_t ldd, void* dqkv, int64_t ldq,
                   float* slab, int slab_rows, const WinGeom& g, float scale, int* nslab, hipStream_t st);
int wattn_bwd_mfma_hd(const void* qkv, int64_t ld, const float* table, const void* dout, int64_t ldd, void* dqkv,
                      int64_t ldq, float* slab, int slab_rows, const WinGeom& g, float scale, int* nslab, hipStream_t st);
int wattn_bwd_mfma(const void* qkv, int64_t ld, const float* table, const void* dout, int64_t ldd, void* dqkv,
                   int64_t ldq, float* slab, int slab_rows, const WinGeom& g, float scale, int dtype, int* nslab,
                   hipStream_t st);

// K8 (swinattn_fwd.hip): LayerNorm + qkv -> window attention -> proj + shortcut in one launch (bf16, ws 8, 6 heads, C = 60 / 90 / 120)
size_t swinattn_pack_bytes(int C);
bool swinattn_supported(int C, int heads, int ws);
int swinattn_fwd_bf16(const bf16* X, int64_t ldx, const float* ln_w, const float* ln_b, const float* Wqkv, const float* bqkv,
                      const float* table, const float* Wproj, const float* bproj, bf16* qkv, int64_t ldq, bf16* a, int64_t lda,
                      bf16* x1, int64_t ld1, float* stats, void* wpack, bool prepacked, const WinGeom& g, float scale, hipStream_t st);
