// gfx950 MFMA building blocks shared by the GEMM-shaped kernels (linear / conv / attention).
//
// One abstraction covers both arithmetic modes: a "pack" is the 16 bytes one lane feeds to the matrix
// core per k-step (4 fp32 or 8 bf16).  With lane = (r = lane & 31, h = lane >> 5):
//   bf16: v_mfma_f32_32x32x16_bf16  — lane holds A[r][16t + 8h + j], j = 0..7     (one MFMA per pack)
//   fp32: v_mfma_f32_32x32x2_f32 x4 — lane holds A[r][ 8t + 4h + e], e = 0..3; MFMA e consumes
//         element e of both operands, i.e. physical k = 8t + 4h + e.  The k order is permuted the
//         same way for A and B, so the sum is unchanged; fp32 MFMA is an exact fmaf chain
//         (cdna_hip_programming.md §3), which is what the fp32 parity mode needs.
// In both cases a k-step covers 32 BYTES of a k-contiguous row and lane half h owns bytes
// [32t + 16h, +16) — so LDS tiles and fragment addressing are dtype-agnostic in bytes.
// C/D layout (both): col = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5).
#pragma once
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));  // 16-B load, dword aligned
typedef uint32_t u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));

struct alignas(16) Pack16 {
  uint32_t w[4];
};

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  bf16x2_t v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float bf16lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// SP (float only): the SPLIT arithmetic of the RDST_F32X3 mode — fp32 rows in memory, a matrix operand is a pack of
// [4 bf16 hi | 4 bf16 lo] (hi = bf16(x), lo = bf16(x - hi): 16 mantissa bits, same 16 bytes as the 4 floats); see Mma<float, true>.
// pack() / unpack() always speak the MEMORY format; op() turns a memory pack into an MFMA operand (identity without SP),
// pack_op() = op(pack()).  Every operand of mma() must have gone through op() / pack_op() exactly once.
template <typename E, bool SP = false> struct Mma;

template <> struct Mma<float, false> {
  static constexpr int KP = 8;  // k elements per k-step (both lane halves)
  static constexpr int HP = 4;  // elements per lane pack
  static constexpr bool SPLIT = false;
  static __device__ __forceinline__ Pack16 op(const Pack16& p) { return p; }
  static __device__ __forceinline__ Pack16 pack_op(const float* f) { return pack(f); }
  static __device__ __forceinline__ void mma(f32x16& acc, const Pack16& a, const Pack16& b) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w[e]), __uint_as_float(b.w[e]), acc, 0, 0, 0);
  }
  static __device__ __forceinline__ void mma_da(f32x16& acc, const Pack16& a, const Pack16& b) { mma(acc, a, b); }
  static __device__ __forceinline__ void unpack(const Pack16& p, float* f) {
#pragma unroll
    for (int e = 0; e < 4; ++e) f[e] = __uint_as_float(p.w[e]);
  }
  static __device__ __forceinline__ Pack16 pack(const float* f) {
    Pack16 p;
#pragma unroll
    for (int e = 0; e < 4; ++e) p.w[e] = __float_as_uint(f[e]);
    return p;
  }
};

template <> struct Mma<bf16, false> {
  static constexpr int KP = 16;
  static constexpr int HP = 8;
  static constexpr bool SPLIT = false;
  static __device__ __forceinline__ Pack16 op(const Pack16& p) { return p; }
  static __device__ __forceinline__ Pack16 pack_op(const float* f) { return pack(f); }
  static __device__ __forceinline__ void mma(f32x16& acc, const Pack16& a, const Pack16& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc,
                                                  0, 0, 0);
  }
  static __device__ __forceinline__ void mma_da(f32x16& acc, const Pack16& a, const Pack16& b) { mma(acc, a, b); }
  static __device__ __forceinline__ void unpack(const Pack16& p, float* f) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      f[2 * e] = bf16lo(p.w[e]);
      f[2 * e + 1] = bf16hi(p.w[e]);
    }
  }
  static __device__ __forceinline__ Pack16 pack(const float* f) {
    Pack16 p;
#pragma unroll
    for (int e = 0; e < 4; ++e) p.w[e] = pack_bf16x2(f[2 * e], f[2 * e + 1]);
    return p;
  }
};

// RDST_F32X3 on the network's GEMMs.  With A = [a_hi | a_lo] in the 8 k-slots of a lane and B = [b_hi | b_hi] resp. [b_lo | b_lo],
// two v_mfma_f32_32x32x16_bf16 add up all four partial products (a_hi + a_lo)(b_hi + b_lo) of 4 k per lane half, i.e. the
// 8 k of one fp32 k-step: 64 cycles of the matrix pipe where 4 x v_mfma_f32_32x32x2_f32 take 256.  The only error is the
// 16-bit representation of each operand (<= 2^-17 relative, against 2^-9 of a bf16 operand); accumulation stays fp32.
__device__ __forceinline__ Pack16 split_pack4(const float* f) {
  Pack16 p;
  p.w[0] = pack_bf16x2(f[0], f[1]);
  p.w[1] = pack_bf16x2(f[2], f[3]);
  p.w[2] = pack_bf16x2(f[0] - bf16lo(p.w[0]), f[1] - bf16hi(p.w[0]));
  p.w[3] = pack_bf16x2(f[2] - bf16lo(p.w[1]), f[3] - bf16hi(p.w[1]));
  return p;
}
template <> struct Mma<float, true> : Mma<float, false> {
  static constexpr bool SPLIT = true;
  static __device__ __forceinline__ Pack16 op(const Pack16& p) {
    float f[4];
    unpack(p, f);
    return split_pack4(f);
  }
  static __device__ __forceinline__ Pack16 pack_op(const float* f) { return split_pack4(f); }
  static __device__ __forceinline__ void mma(f32x16& acc, const Pack16& a, const Pack16& b) {
    Pack16 bh, bl;
    bh.w[0] = b.w[0]; bh.w[1] = b.w[1]; bh.w[2] = b.w[0]; bh.w[3] = b.w[1];
    bl.w[0] = b.w[2]; bl.w[1] = b.w[3]; bl.w[2] = b.w[2]; bl.w[3] = b.w[3];
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, bh), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, bl), acc, 0, 0, 0);
  }
  // the same with the FIRST operand duplicated (for call sites whose second operand is the one re-read per tile)
  static __device__ __forceinline__ void mma_da(f32x16& acc, const Pack16& a, const Pack16& b) {
    Pack16 ah, al;
    ah.w[0] = a.w[0]; ah.w[1] = a.w[1]; ah.w[2] = a.w[0]; ah.w[3] = a.w[1];
    al.w[0] = a.w[2]; al.w[1] = a.w[3]; al.w[2] = a.w[2]; al.w[3] = a.w[3];
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, ah), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, al), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
  }
};

// Guarded load of one lane pack (HP elements starting at element k0 of a row of K valid elements).
// Rows need only dword alignment (the dense-concat slices of an RDSTB start at 60/90/120 elements).
template <typename T>
__device__ __forceinline__ Pack16 load_pack(const T* __restrict__ rowp, int k0, int K, bool valid) {
  constexpr int HP = Mma<T>::HP;
  Pack16 p;
  if (valid && k0 + HP <= K && (reinterpret_cast<uintptr_t>(rowp + k0) & 3) == 0) {
    const u32x4_a4 v = *reinterpret_cast<const u32x4_a4*>(rowp + k0);
    p.w[0] = v.x; p.w[1] = v.y; p.w[2] = v.z; p.w[3] = v.w;
    return p;
  }
  float f[HP];
#pragma unroll
  for (int e = 0; e < HP; ++e) f[e] = (valid && k0 + e < K) ? to_f32<T>(rowp[k0 + e]) : 0.f;
  return Mma<T>::pack(f);
}

// Pack HP fp32 parameters (converted to T) starting at src[k0*stride], guarded by k < K.
template <typename T>
__device__ __forceinline__ Pack16 pack_from_f32(const float* __restrict__ src, int k0, int K, int64_t stride, bool valid) {
  constexpr int HP = Mma<T>::HP;
  float f[HP];
#pragma unroll
  for (int e = 0; e < HP; ++e) f[e] = (valid && k0 + e < K) ? src[(int64_t)(k0 + e) * stride] : 0.f;
  return Mma<T>::pack(f);
}

// Stage converted fp32 parameters into an LDS image of lane packs, BATCHED: every thread issues the
// loads of U items before it converts and stores any of them, so a workgroup pays U-fold fewer
// global-load latencies (the one-item-at-a-time loop cost 10-25 us per launch, measured).
//   item idx -> addr(idx, base, dst, ok):  base = pointer of element 0 of the item's pack (k0 = first k),
//   element e of the pack is base[e * stride]; elements with k0 + e >= K (or !ok) are zero.
// F: void(int idx, const float*& base, int& k0, char*& dst, bool& ok)
template <typename T, int U, bool SP = false, class F>
__device__ __forceinline__ void stage_packs_batched(int total, int K, int64_t stride, int tid, int nthreads, F addr, int rot = 0) {
  constexpr int HP = Mma<T>::HP;
  // `rot` rotates the item order per workgroup: every workgroup stages the SAME parameters at the same moment, and
  // without it all of them hit the same few L2 channels in lockstep
  for (int base_idx = tid; base_idx < total; base_idx += nthreads * U) {
    float f[U][HP];
    char* dst[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int idx = base_idx + nthreads * u;
      const float* src = nullptr;
      int k0 = 0;
      bool ok = false;
      dst[u] = nullptr;
      if (idx < total) {
        idx += rot;
        if (idx >= total) idx -= total;
        addr(idx, src, k0, dst[u], ok);
      }
      if (ok && stride == 1 && k0 + HP <= K) {   // contiguous pack: 16-B loads (fp32 parameters are dword aligned)
#pragma unroll
        for (int q = 0; q < HP / 4; ++q) {
          const u32x4_a4 v = *reinterpret_cast<const u32x4_a4*>(src + 4 * q);
          f[u][4 * q] = __uint_as_float(v.x); f[u][4 * q + 1] = __uint_as_float(v.y);
          f[u][4 * q + 2] = __uint_as_float(v.z); f[u][4 * q + 3] = __uint_as_float(v.w);
        }
      } else {
#pragma unroll
        for (int e = 0; e < HP; ++e) f[u][e] = (ok && k0 + e < K) ? src[(int64_t)e * stride] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (dst[u]) *reinterpret_cast<Pack16*>(dst[u]) = Mma<T, SP>::pack_op(f[u]);
  }
}

// Stage a strided-transposed view of fp32 parameters into LDS with COALESCED global reads: the source is
// read as R rows of L contiguous floats (row stride S floats; consecutive lanes read consecutive floats,
// 16-B loads when the geometry allows), every element is converted and written on its own to
// lds + dst(rr, j) (negative = dropped).  A gather with one pack per lane made every lane touch its own
// cache line (measured: 40-90 us of prologue per conv launch, 7-12 us per Linear dgrad launch).
// The destination region must have been zero-filled (padding rows / columns) before.
template <typename T, class F>
__device__ __forceinline__ void stage_scatter(const float* __restrict__ src, int R, int L, int64_t S, int tid, int nthreads,
                                              char* lds, F dst) {
  auto put = [&](int rr, int j, float v) {
    const int off = dst(rr, j);
    if (off >= 0) *reinterpret_cast<T*>(lds + off) = from_f32<T>(v);
  };
  if (S == L && (L & 3) != 0) {
    // contiguous block whose rows are not whole float4s: read it flat, 16 B per lane, and split each group over
    // the (at most two) rows it touches — one division per group instead of the scalar path's 4x more loads
    const int64_t tot = (int64_t)R * L;
    const int n4 = (int)(tot >> 2);
    constexpr int U = 8;
    for (int base = tid; base < n4; base += nthreads * U) {
      u32x4_a4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = base + nthreads * u;
        v[u] = *reinterpret_cast<const u32x4_a4*>(src + 4 * (int64_t)(i < n4 ? i : n4 - 1));
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = base + nthreads * u;
        if (i < n4) {
          const int f = 4 * i;
          int rr = f / L, j = f - rr * L;
          const float e4[4] = {__uint_as_float(v[u].x), __uint_as_float(v[u].y), __uint_as_float(v[u].z), __uint_as_float(v[u].w)};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            put(rr, j, e4[e]);
            if (++j == L) { j = 0; ++rr; }
          }
        }
      }
    }
    for (int64_t f = 4 * (int64_t)n4 + tid; f < tot; f += nthreads) put((int)(f / L), (int)(f % L), src[f]);
    return;
  }
  // thread -> (row, column group) by shift / mask (no integer divisions): a row is padded to a power of two of groups
  const bool vec = (L & 3) == 0;
  const int G = vec ? (L >> 2) : L;            // column groups per row (of 4 floats / of 1 float)
  int sh = 0;
  while ((1 << sh) < G) ++sh;
  const int total = R << sh;
  if (vec) {
    constexpr int U = 16;   // loads in flight per thread before the first conversion (the loop is latency bound)
    for (int base = tid; base < total; base += nthreads * U) {
      u32x4_a4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = base + nthreads * u;
        int rr = i >> sh, j4 = i & ((1 << sh) - 1);
        rr = rr < R ? rr : R - 1;
        j4 = j4 < G ? j4 : G - 1;
        v[u] = *reinterpret_cast<const u32x4_a4*>(src + (int64_t)rr * S + 4 * j4);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = base + nthreads * u;
        const int rr = i >> sh, j4 = i & ((1 << sh) - 1);
        if (rr < R && j4 < G) {
          put(rr, 4 * j4, __uint_as_float(v[u].x));
          put(rr, 4 * j4 + 1, __uint_as_float(v[u].y));
          put(rr, 4 * j4 + 2, __uint_as_float(v[u].z));
          put(rr, 4 * j4 + 3, __uint_as_float(v[u].w));
        }
      }
    }
  } else {
    constexpr int U = 16;
    for (int base = tid; base < total; base += nthreads * U) {
      float v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = base + nthreads * u;
        int rr = i >> sh, j = i & ((1 << sh) - 1);
        rr = rr < R ? rr : R - 1;
        j = j < G ? j : G - 1;
        v[u] = src[(int64_t)rr * S + j];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = base + nthreads * u;
        const int rr = i >> sh, j = i & ((1 << sh) - 1);
        if (rr < R && j < G) put(rr, j, v[u]);
      }
    }
  }
}
__device__ __forceinline__ void lds_zero16(char* lds, int bytes, int tid, int nthreads) {   // bytes % 16 == 0
  for (int i = tid * 16; i < bytes; i += nthreads * 16) *reinterpret_cast<float4*>(lds + i) = make_float4(0.f, 0.f, 0.f, 0.f);
}

// row of accumulator register v for lane half h (32x32 C/D layout)
__device__ __forceinline__ int acc_row(int v, int h) { return (v & 3) + 8 * (v >> 2) + 4 * h; }

// LDS row stride (bytes) for a k-contiguous tile of `kelems` elements of `esize` bytes read with 16-B
// packs: rounded up to whole 32-B k-steps plus one 16-B slot, so (stride/16) is odd and the 16 rows
// of a ds_read_b128 lane group land on distinct 16-B slots of the 256-B bank row.
static inline int lds_row_bytes(int kelems, int esize) {
  const int b = ((kelems * esize + 31) / 32) * 32;
  return b + 16;
}

// ---------------------------------------------------------------------------------------------------
// Row-wise epilogue for a 32x32 accumulator tile.
// In the C/D layout a lane owns ONE column, so storing from the accumulators writes 2-4 bytes per
// lane (128-256 B per store instruction) and the kernel becomes store-ISSUE-bound (~1.2 TB/s measured).
// Instead each wave bounces the tile through a private 32x32 fp32 LDS buffer (row stride 128 B: the two
// rows of a ds_read_b128 lane group fill the 64 banks exactly) and finishes it row-major: lane ->
// (row = lane>>3 + 8*pass, 4 consecutive columns), with the residual / activation-gradient / accumulate
// operands read as 8-16-B row segments and the result stored 8 B (bf16) or 16 B (fp32) per lane.
// DS operations of one wave execute in order, so no barrier is needed around the bounce.
struct TileEpilogue {
  const void* R; int64_t ldr;        // + R[row][col]                        (forward residual)
  const void* Xa; int64_t ldxa; int act;  // * act'(Xa[row][col])            (dgrad through an input activation)
  void* Y; int64_t ldy;                   // Y = value (+ ...)
  const void* Acc; int64_t ldacc;         // + Acc[row][col]  (dX_add of the backward; may alias Y)
  float* Yf32; int64_t ldf;          // alternative fp32 destination (dgrad in front of a LayerNorm)
};

template <typename T>
__device__ __forceinline__ void tile_store_rows(float* __restrict__ eps, const float (&vals)[16], int lane, int64_t row0,
                                                int64_t M, int col0, int N, const TileEpilogue& e) {
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int v = 0; v < 16; ++v) eps[acc_row(v, h) * 32 + r] = vals[v];
  __builtin_amdgcn_wave_barrier();
  const int chunk = lane & 7, col = col0 + chunk * 4;
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int row = (lane >> 3) + 8 * pass;
    const int64_t rr = row0 + row;
    const float4 f4 = *reinterpret_cast<const float4*>(eps + row * 32 + chunk * 4);
    if (rr < M && col < N) {
      float f[4] = {f4.x, f4.y, f4.z, f4.w};
      const bool full = col + 4 <= N;
      if (e.Yf32) {
        float* dst = e.Yf32 + rr * e.ldf + col;
        if (full && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) *reinterpret_cast<float4*>(dst) = f4;
        else
#pragma unroll
          for (int q = 0; q < 4; ++q) if (col + q < N) dst[q] = f[q];
        continue;
      }
      T* dst = reinterpret_cast<T*>(e.Y) + rr * e.ldy + col;
      const T* rp = e.R ? reinterpret_cast<const T*>(e.R) + rr * e.ldr + col : nullptr;
      const T* xp = (e.Xa && e.act) ? reinterpret_cast<const T*>(e.Xa) + rr * e.ldxa + col : nullptr;
      auto add4 = [&](const T* src, bool mul_actgrad) {
        float g[4];
        if (full && (reinterpret_cast<uintptr_t>(src) & 3) == 0) {
          if (sizeof(T) == 2) {
            const u32x2_a4 u = *reinterpret_cast<const u32x2_a4*>(src);
            g[0] = bf16lo(u.x); g[1] = bf16hi(u.x); g[2] = bf16lo(u.y); g[3] = bf16hi(u.y);
          } else {
            const u32x4_a4 u = *reinterpret_cast<const u32x4_a4*>(src);
            g[0] = __uint_as_float(u.x); g[1] = __uint_as_float(u.y); g[2] = __uint_as_float(u.z); g[3] = __uint_as_float(u.w);
          }
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) g[q] = (col + q < N) ? to_f32<T>(src[q]) : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) f[q] = mul_actgrad ? f[q] * act_grad(g[q], e.act) : f[q] + g[q];
      };
      if (xp) add4(xp, true);
      if (rp) add4(rp, false);
      if (e.Acc) add4(reinterpret_cast<const T*>(e.Acc) + rr * e.ldacc + col, false);
      if (full && (reinterpret_cast<uintptr_t>(dst) & 3) == 0) {
        if (sizeof(T) == 2) {
          u32x2_a4 u;
          u.x = pack_bf16x2(f[0], f[1]); u.y = pack_bf16x2(f[2], f[3]);
          *reinterpret_cast<u32x2_a4*>(dst) = u;
        } else {
          u32x4_a4 u;
          u.x = __float_as_uint(f[0]); u.y = __float_as_uint(f[1]); u.z = __float_as_uint(f[2]); u.w = __float_as_uint(f[3]);
          *reinterpret_cast<u32x4_a4*>(dst) = u;
        }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) if (col + q < N) dst[q] = from_f32<T>(f[q]);
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
}
