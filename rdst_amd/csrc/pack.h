// Weight packing for the register-stationary kernels (lin3_mfma.hip, conv3_mfma.hip): the device bodies, shared by the
// per-call pack kernels and by the batched one (pack_batch.hip: every layer of a network in a handful of launches).
#pragma once
#include "common.h"
#include "mfma.h"

constexpr int PK_FWD = 0, PK_DGRAD = 1, PK_DGRAD_UNSHUF = 2;

// fragment (nt, ks) = 64 lanes x 8 bf16: lane (r, h): output n = 32 nt + r, input k = 16 ks + 8 h + e: W[n][k] gamma[k] s.
// sb[0][n] = S[n] = sum_k of the rounded values, sb[1][n] = b'[n] = (bias[n] + sum_k W[n][k] beta[k]) s.
__device__ __forceinline__ void lin3_pack_block(int bid, const float* __restrict__ W, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ bias,
                                                        bf16* __restrict__ wp, float* __restrict__ sb, int N, int K, int ksteps,
                                                        int ntiles, float s) {
  const int nfr = ntiles * ksteps * 64, nb1 = (nfr + 255) / 256;
  if (bid < nb1) {
    const int i = bid * 256 + threadIdx.x;
    if (i >= nfr) return;
    const int lane = i & 63, f = i >> 6;
    const int ks = f % ksteps, nt = f / ksteps;
    const int n = nt * 32 + (lane & 31), k0 = ks * 16 + (lane >> 5) * 8;
    uint32_t w[4];
#pragma unroll
    for (int e2 = 0; e2 < 4; ++e2) {
      float v[2];
#pragma unroll
      for (int e1 = 0; e1 < 2; ++e1) {
        const int k = k0 + 2 * e2 + e1;
        v[e1] = (n < N && k < K) ? W[(int64_t)n * K + k] * (gamma ? gamma[k] : 1.f) * s : 0.f;
      }
      w[e2] = pack_bf16x2(v[0], v[1]);
    }
    u32x4_a4 o;
    o.x = w[0]; o.y = w[1]; o.z = w[2]; o.w = w[3];
    *reinterpret_cast<u32x4_a4*>(wp + (int64_t)i * 8) = o;
    return;
  }
  // S / b': one wave per output row (coalesced row reads, fixed shuffle tree)
  const int n = (bid - nb1) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int NP = ntiles * 32;
  if (n >= NP) return;
  float S = 0.f, bb = 0.f;
  if (n < N) {
    for (int k = lane; k < K; k += 64) {
      const float w = W[(int64_t)n * K + k];
      S += __bfloat162float(__float2bfloat16(w * (gamma ? gamma[k] : 1.f) * s));
      if (beta) bb = fmaf(w, beta[k], bb);
    }
    S = wave_sum(S);
    bb = wave_sum(bb);
    bb = (bb + (bias ? bias[n] : 0.f)) * s;
  }
  if (lane == 0) {
    sb[n] = S;
    sb[NP + n] = bb;
  }
}


// The same image for a Linear whose N outputs are `nsec` equal SECTIONS (qkv: q | k | v, N = 3 C), each padded to whole
// 32-row tiles of its own: tile nt = sec * spt + j holds outputs sec * Cs + 32 j + r (zero rows where 32 j + r >= Cs), so a
// tile never straddles two sections and the consumer (swinattn_fwd.hip) writes each tile into one LDS section.
// sb[0][32 nt + r] = S, sb[1][32 nt + r] = b' of that output (0 for the pad rows).
__device__ __forceinline__ void lin3sec_pack_block(int bid, const float* __restrict__ W, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, const float* __restrict__ bias,
                                                   bf16* __restrict__ wp, float* __restrict__ sb, int N, int K, int nsec, float s) {
  const int Cs = N / nsec, spt = (Cs + 31) / 32, ntiles = nsec * spt, ksteps = (K + 15) / 16;
  const int nfr = ntiles * ksteps * 64, nb1 = (nfr + 255) / 256;
  if (bid < nb1) {
    const int i = bid * 256 + threadIdx.x;
    if (i >= nfr) return;
    const int lane = i & 63, f = i >> 6;
    const int ks = f % ksteps, nt = f / ksteps;
    const int sec = nt / spt, c = (nt - sec * spt) * 32 + (lane & 31);
    const int n = sec * Cs + c, k0 = ks * 16 + (lane >> 5) * 8;
    uint32_t w[4];
#pragma unroll
    for (int e2 = 0; e2 < 4; ++e2) {
      float v[2];
#pragma unroll
      for (int e1 = 0; e1 < 2; ++e1) {
        const int k = k0 + 2 * e2 + e1;
        v[e1] = (c < Cs && k < K) ? W[(int64_t)n * K + k] * (gamma ? gamma[k] : 1.f) * s : 0.f;
      }
      w[e2] = pack_bf16x2(v[0], v[1]);
    }
    u32x4_a4 o;
    o.x = w[0]; o.y = w[1]; o.z = w[2]; o.w = w[3];
    *reinterpret_cast<u32x4_a4*>(wp + (int64_t)i * 8) = o;
    return;
  }
  const int np = (bid - nb1) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;   // padded output index
  const int NP = ntiles * 32;
  if (np >= NP) return;
  const int nt = np >> 5, sec = nt / spt, c = (nt - sec * spt) * 32 + (np & 31), n = sec * Cs + c;
  float S = 0.f, bb = 0.f;
  if (c < Cs) {
    for (int k = lane; k < K; k += 64) {
      const float w = W[(int64_t)n * K + k];
      S += __bfloat162float(__float2bfloat16(w * (gamma ? gamma[k] : 1.f) * s));
      if (beta) bb = fmaf(w, beta[k], bb);
    }
    S = wave_sum(S);
    bb = wave_sum(bb);
    bb = (bb + (bias ? bias[n] : 0.f)) * s;
  }
  if (lane == 0) {
    sb[np] = S;
    sb[NP + np] = bb;
  }
}
static __host__ __device__ inline int lin3sec_tiles(int K, int N, int nsec) { (void)K; return nsec * ((N / nsec + 31) / 32); }
static inline int lin3sec_pack_blocks(int K, int N, int nsec) {
  const int nt = lin3sec_tiles(K, N, nsec), ks = (K + 15) / 16;
  return (nt * ks * 64 + 255) / 256 + (nt * 32 + 3) / 4;
}
static __host__ __device__ inline size_t lin3sec_pack_bytes(int K, int N, int nsec) {
  const int nt = lin3sec_tiles(K, N, nsec), ks = (K + 15) / 16;
  return (size_t)nt * ks * 1024 + (size_t)2 * nt * 32 * 4 + 256;
}


// packed weights: fragment (ct, tap, ks) = 64 lanes x 8 bf16; lane (r, h): output channel n = 32 ct + r,
// contraction k = 16 ks + 8 h + e.
//   PK_FWD          : Wc[n][k][tap]
//   PK_DGRAD        : Wc[k][n][8 - tap]                       (contraction over co, mirrored tap)
//   PK_DGRAD_UNSHUF : Wc[4 c' + q][n][8 - tap], k = 60 q + c' (conv channel 4c'+q is sub-pixel q of channel c')
__device__ __forceinline__ void conv3_pack_block(int bid, const float* __restrict__ Wc, bf16* __restrict__ out, int Cin,
                                                         int Cout, int K, int N, int ksteps, int ctiles, int mode, float s) {
  const int i = bid * 256 + threadIdx.x;          // one thread per 8 packed elements (16 B)
  const int total = ctiles * 9 * ksteps * 64;
  if (i >= total) return;
  const int lane = i & 63, f = i >> 6;
  const int ks = f % ksteps, tap = (f / ksteps) % 9, ct = f / (ksteps * 9);
  const int n = ct * 32 + (lane & 31);
  uint32_t w[4];
#pragma unroll
  for (int e2 = 0; e2 < 4; ++e2) {
    float v[2];
#pragma unroll
    for (int e1 = 0; e1 < 2; ++e1) {
      const int k = ks * 16 + (lane >> 5) * 8 + 2 * e2 + e1;
      float x = 0.f;
      if (n < N && k < K) {
        if (mode == PK_FWD) x = Wc[((int64_t)n * Cin + k) * 9 + tap];
        else if (mode == PK_DGRAD) x = Wc[((int64_t)k * Cin + n) * 9 + (8 - tap)];
        else {
          const int cq = Cout / 4, q = k / cq, c = k - q * cq;
          x = Wc[((int64_t)(4 * c + q) * Cin + n) * 9 + (8 - tap)];
        }
      }
      v[e1] = x * s;   // out_scale rides on the weights: (conv(W) + bias) s = conv(s W) + s bias
    }
    w[e2] = pack_bf16x2(v[0], v[1]);
  }
  u32x4_a4 o;
  o.x = w[0]; o.y = w[1]; o.z = w[2]; o.w = w[3];
  *reinterpret_cast<u32x4_a4*>(out + (int64_t)i * 8) = o;
}


static inline int lin3_pack_blocks(int K, int N) {
  const int nt = (N + 31) / 32, ks = (K + 15) / 16;
  return (nt * ks * 64 + 255) / 256 + (nt * 32 + 3) / 4;
}
static inline int conv3_pack_blocks(int K, int N) {
  const int ks = (K + 15) / 16, ct = (N + 31) / 32;
  return (ct * 9 * ks * 64 + 255) / 256;
}


// ---- RDST_F32X3 image of a Linear layer (lin3x_mfma.hip): every weight as TWO bf16 terms, hi = bf16(w), lo = bf16(w - hi) ----
// fragment (nt, ks) = [hi: 64 lanes x 8 bf16][lo: 64 lanes x 8 bf16] (2 KB); lane (r, h): output n = 32 nt + r, inputs k = 16 ks + 8 h + e of
// w = W[n][k] gamma[k] s.  Behind the fragments: b'[32 nt + r] = (bias[n] + sum_k W[n][k] beta[k]) s (fp32; 0 for the pad rows).
// The kernels that read it normalise the rows BEFORE the product (x-hat = (x - mean) rstd, split into hi + lo), so no S[n] term exists.
__device__ __forceinline__ void lin3x_pack_block(int bid, const float* __restrict__ W, const float* __restrict__ gamma,
                                                 const float* __restrict__ beta, const float* __restrict__ bias, uint32_t* __restrict__ wp,
                                                 float* __restrict__ bp, int N, int K, int ksteps, int ntiles, float s) {
  const int nfr = ntiles * ksteps * 64, nb1 = (nfr + 255) / 256;
  if (bid < nb1) {
    const int i = bid * 256 + threadIdx.x;
    if (i >= nfr) return;
    const int lane = i & 63, f = i >> 6;
    const int ks = f % ksteps, nt = f / ksteps;
    const int n = nt * 32 + (lane & 31), k0 = ks * 16 + (lane >> 5) * 8;
    u32x4_a4 hi, lo;
    uint32_t* ph = reinterpret_cast<uint32_t*>(&hi);
    uint32_t* pl = reinterpret_cast<uint32_t*>(&lo);
#pragma unroll
    for (int e2 = 0; e2 < 4; ++e2) {
      float v[2];
#pragma unroll
      for (int e1 = 0; e1 < 2; ++e1) {
        const int k = k0 + 2 * e2 + e1;
        v[e1] = (n < N && k < K) ? W[(int64_t)n * K + k] * (gamma ? gamma[k] : 1.f) * s : 0.f;
      }
      ph[e2] = pack_bf16x2(v[0], v[1]);
      pl[e2] = pack_bf16x2(v[0] - bf16lo(ph[e2]), v[1] - bf16hi(ph[e2]));
    }
    uint32_t* dst = wp + ((int64_t)f * 128 + lane) * 4;
    *reinterpret_cast<u32x4_a4*>(dst) = hi;
    *reinterpret_cast<u32x4_a4*>(dst + 256) = lo;
    return;
  }
  const int n = (bid - nb1) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;   // b': one wave per output row
  const int NP = ntiles * 32;
  if (n >= NP) return;
  float bb = 0.f;
  if (n < N) {
    if (beta)
      for (int k = lane; k < K; k += 64) bb = fmaf(W[(int64_t)n * K + k], beta[k], bb);
    bb = wave_sum(bb);
    bb = (bb + (bias ? bias[n] : 0.f)) * s;
  }
  if (lane == 0) bp[n] = bb;
}
static inline int lin3x_pack_blocks(int K, int N) {
  const int nt = (N + 31) / 32, ks = (K + 15) / 16;
  return (nt * ks * 64 + 255) / 256 + (nt * 32 + 3) / 4;
}
static __host__ __device__ inline size_t lin3x_pack_bytes(int K, int N) {
  const int nt = (N + 31) / 32, ks = (K + 15) / 16;
  return (size_t)nt * ks * 2048 + (size_t)nt * 32 * 4 + 256;
}


// ---- RDST_F32X3 image of a 3x3 convolution (conv3x_mfma.hip): fragment (ct, tap, ks) = [hi: 64 lanes x 8 bf16][lo: 64 lanes x 8 bf16];
// lane (r, h): output channel n = 32 ct + r, contraction k = 16 ks + 8 h + e; same PK_ modes as conv3_pack_block
// (cmul, coff): contraction index k of a PK_DGRAD image is conv output channel cmul k + coff (the sub-pixel slices of conv + PixelShuffle)
__device__ __forceinline__ void conv3x_pack_block(int bid, const float* __restrict__ Wc, uint32_t* __restrict__ out, int Cin, int Cout,
                                                  int K, int N, int ksteps, int ctiles, int mode, float s, int cmul = 1, int coff = 0) {
  const int i = bid * 256 + threadIdx.x;          // one thread per (fragment, lane): 16 B of hi and 16 B of lo
  const int total = ctiles * 9 * ksteps * 64;
  if (i >= total) return;
  const int lane = i & 63, f = i >> 6;
  const int ks = f % ksteps, tap = (f / ksteps) % 9, ct = f / (ksteps * 9);
  const int n = ct * 32 + (lane & 31);
  u32x4_a4 hi, lo;
  uint32_t* ph = reinterpret_cast<uint32_t*>(&hi);
  uint32_t* pl = reinterpret_cast<uint32_t*>(&lo);
#pragma unroll
  for (int e2 = 0; e2 < 4; ++e2) {
    float v[2];
#pragma unroll
    for (int e1 = 0; e1 < 2; ++e1) {
      const int k = ks * 16 + (lane >> 5) * 8 + 2 * e2 + e1;
      float x = 0.f;
      if (n < N && k < K) {
        if (mode == PK_FWD) x = Wc[((int64_t)n * Cin + k) * 9 + tap];
        else x = Wc[((int64_t)(cmul * k + coff) * Cin + n) * 9 + (8 - tap)];   // PK_DGRAD: contraction over co, mirrored tap
      }
      v[e1] = x * s;
    }
    ph[e2] = pack_bf16x2(v[0], v[1]);
    pl[e2] = pack_bf16x2(v[0] - bf16lo(ph[e2]), v[1] - bf16hi(ph[e2]));
  }
  uint32_t* dst = out + ((int64_t)f * 128 + lane) * 4;
  *reinterpret_cast<u32x4_a4*>(dst) = hi;
  *reinterpret_cast<u32x4_a4*>(dst + 256) = lo;
}
static inline int conv3x_pack_blocks(int K, int N) {
  const int ks = (K + 15) / 16, ct = (N + 31) / 32;
  return (ct * 9 * ks * 64 + 255) / 256;
}
