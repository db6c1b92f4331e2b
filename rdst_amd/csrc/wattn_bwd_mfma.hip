// K2 on the gfx950 matrix cores: fused window attention backward for 8x8 windows, 6 heads.
//
// Persistent workgroups of 2*heads = 12 waves; wave (t, hd) owns tile t (32 tokens) of head hd.
// Per window the qkv rows and the dOut rows are read once from HBM into four LDS sections (Q is
// pre-multiplied by scale [* log2 e in bf16 mode] on the way in, so S^T = K Qs^T needs no per-element
// scaling and the relative-position bias is simply the INITIAL ACCUMULATOR of the MFMA chain).
//   pass T (keys in the accumulator registers, query on the lane; = the forward's orientation):
//     S^T -> softmax statistics (m, 1/l per query: in-lane + one cross-half shuffle) -> P^T;
//     dP^T = V dO_h^T;  delta = rowsum(P dP);  dS^T = P^T (dP^T - delta);
//     d(table) += dS^T by conflict-free LDS atomics (accumulated across all windows of the workgroup);
//     dQ_h = scale * (dS^T)^T K_h   (accumulator tile as the A operand, K read transposed)
//   pass N (queries in the registers, key on the lane): S, dP recomputed with the roles swapped,
//     P / dS rebuilt from the per-query statistics in LDS;
//     dV_h = P^T dO_h,  dK_h = dS^T Qs_h   (again accumulator-as-operand + transposed reads)
// Heads are selected by masking packs to the head's channels (head dims 10/15/20 are not k-step
// multiples).  dQ/dK/dV land in three more LDS sections and go back to HBM as full dqkv rows.
// d(table) partials: one slab row per workgroup, summed in fixed order by dtable_reduce.
#include "common.h"
#include "wattn.h"
#include "mfma.h"
#include <stdlib.h>

namespace {

constexpr int TS = 24;
constexpr int HEADS = 6;
constexpr int NUNITS = 2 * HEADS;   // (query / key tile, head) units of a window

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
template <int GRAN> struct Chunk;
template <> struct Chunk<16> { typedef u32x4_t type; };
template <> struct Chunk<8> { typedef u32x2_t type; };
template <> struct Chunk<4> { typedef uint32_t type; };

template <typename T>
struct WbArgs {
  const T* qkv; int64_t ld;
  const float* table;
  const T* dout; int64_t ldd;
  T* dqkv; int64_t ldq;
  float* slab;
  WinGeom g;
  float scale;
  int d, ldt;
  int dbg;  // ablation switches (RDST_K2_DEBUG): 1 = skip d(table) atomics, 2 = skip pass T, 4 = skip pass N
};

__device__ __forceinline__ uint32_t mask_bits_bf16(int c0, int lo, int hi) {
  return ((c0 >= lo && c0 < hi) ? 0x0000ffffu : 0u) | ((c0 + 1 >= lo && c0 + 1 < hi) ? 0xffff0000u : 0u);
}

template <typename T, bool SP = false>
__device__ __forceinline__ Pack16 masked_pack(const char* base, int c0, int c_lo, int c_hi) {
  Pack16 p = *reinterpret_cast<const Pack16*>(base + (size_t)c0 * sizeof(T));
  if (SP) {   // [4 bf16 hi | 4 bf16 lo] of channels c0 .. c0 + 3
    const uint32_t m0 = mask_bits_bf16(c0, c_lo, c_hi), m1 = mask_bits_bf16(c0 + 2, c_lo, c_hi);
    p.w[0] &= m0; p.w[1] &= m1; p.w[2] &= m0; p.w[3] &= m1;
  } else if (sizeof(T) == 2) {
#pragma unroll
    for (int e = 0; e < 4; ++e) p.w[e] &= mask_bits_bf16(c0 + 2 * e, c_lo, c_hi);
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) p.w[e] = (c0 + e >= c_lo && c0 + e < c_hi) ? p.w[e] : 0u;
  }
  return p;
}

// acc (32 x 32, rows = tokens of tile `rt`, cols = channel tile ct) += X^T . Bsec
//   X[kk][v]: accumulator tiles whose ROW index (tokens of tile kk) is contracted; Bsec rows = those tokens;
//   col0 = first of the 32 columns (bf16: a multiple of 32 — the transposed reads want aligned chunks; fp32: any).
template <typename T, bool SP = false>
__device__ __forceinline__ void acc_xt_b(f32x16& acc, const f32x16 (&X)[2], const char* Bsec, int ldt, int col0, bool colin,
                                         int lane) {
  const int r = lane & 31, h = lane >> 5;
  if constexpr (SP) {
    // split mode: the section rows are packs [4 hi | 4 lo] of 4 channels (16 B); a transposed read takes the hi (or, 8 bytes
    // on, the lo) halves of 4 token rows x 16 channels; a k-step is 8 token rows: accumulator registers 4s .. 4s + 3 of both halves
    const int gq = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const int colB = col0 + 16 * (gq & 1) + 4 * pp;   // a multiple of 4 (col0 is a multiple of 32)
    const uint32_t cm = colin ? 0xffffffffu : 0u;
    typedef __attribute__((address_space(3))) s16x4_t* lds_p;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        float f[4] = {X[kk][4 * s4], X[kk][4 * s4 + 1], X[kk][4 * s4 + 2], X[kk][4 * s4 + 3]};
        const Pack16 a = split_pack4(f);
        const int rowb = kk * 32 + 8 * s4 + 4 * h + q;
        const char* bp = Bsec + (size_t)rowb * ldt + colB * 4;
        const s16x4_t b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)bp);
        const s16x4_t b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(bp + 8));
        const uint2 u0 = __builtin_bit_cast(uint2, b0), u1 = __builtin_bit_cast(uint2, b1);
        Pack16 bb;
        bb.w[0] = u0.x & cm; bb.w[1] = u0.y & cm; bb.w[2] = u1.x & cm; bb.w[3] = u1.y & cm;
        Mma<float, true>::mma(acc, a, bb);
      }
  } else if constexpr (sizeof(T) == 2) {
    const int gq = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const int colB = col0 + 16 * (gq & 1) + 4 * pp;
    const uint32_t cm = colin ? 0xffffffffu : 0u;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        Pack16 a;
#pragma unroll
        for (int e = 0; e < 4; ++e) a.w[e] = pack_bf16x2(X[kk][8 * s + 2 * e], X[kk][8 * s + 2 * e + 1]);
        const int rowb = kk * 32 + 16 * s + 4 * h + q;
        typedef __attribute__((address_space(3))) s16x4_t* lds_p;
        const s16x4_t b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(Bsec + (size_t)rowb * ldt + colB * 2));
        const s16x4_t b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(Bsec + (size_t)(rowb + 8) * ldt + colB * 2));
        const uint2 u0 = __builtin_bit_cast(uint2, b0), u1 = __builtin_bit_cast(uint2, b1);
        Pack16 bb;
        bb.w[0] = u0.x & cm; bb.w[1] = u0.y & cm; bb.w[2] = u1.x & cm; bb.w[3] = u1.y & cm;
        Mma<T>::mma(acc, a, bb);
      }
  } else {
    const float* Bf = reinterpret_cast<const float*>(Bsec);
    const int ldb = ldt / 4;
    const int col = col0 + r;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const float bv = Bf[(kk * 32 + acc_row(v, h)) * ldb + col];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(X[kk][v], colin ? bv : 0.f, acc, 0, 0, 0);
      }
  }
}

// NWAVES = 12: one unit per wave (bf16: 170 registers per wave are enough).  NWAVES = 8 (fp32): two waves per SIMD with
// 256 registers each, waves 0-3 own two units — the 12-wave form of the fp32 kernel spilled 160-170 registers per lane
// (3.4 GB of scratch traffic per launch); the matrix-core work per SIMD is the same three units either way.
// CT > 0: the channel count is a compile-time constant (LDS row stride and head width with it): with run-time strides
// every one of the ~100 LDS positions of a unit is its own multiply and its own register, all hoisted out of the window
// loop and spilled (440 vector instructions and 136 scratch stores in front of the loop in the fp32 kernel).
template <typename T> constexpr int wb_ldt(int C) {
  int ldt = ((C * (int)sizeof(T) + 31) / 32) * 32;
  return (ldt / 16) % 2 == 0 ? ldt + 16 : ldt;
}
template <typename T, int GRAN, int ITERS, int NWAVES, int CT, bool SP = false>
__global__ void __launch_bounds__(64 * NWAVES) wattn_bwd_mfma_kernel(const WbArgs<T> p) {
  constexpr int NTHREADS = 64 * NWAVES;
  constexpr int MAXR = (64 + NWAVES - 1) / NWAVES;   // token rows staged per wave
  constexpr int NU = (NUNITS + NWAVES - 1) / NWAVES;  // units per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using MM = Mma<T, SP>;   // SP: RDST_F32X3 — the four sections are split pack by pack after they are staged (mfma.h)
  using CH = typename Chunk<GRAN>::type;
  constexpr int KP = MM::KP, HP = MM::HP;
  constexpr bool BF = sizeof(T) == 2;
  constexpr bool L2D = BF || SP;   // softmax in the log2 domain on the hardware exp2 (the split mode is not bit-exact fp32 anyway; exact fp32 keeps expf)
  // fp32: the 32-column window of the accumulate products starts at the head's first channel (one window per head; with
  // windows at multiples of 32 half of the heads of C = 90 / 120 straddle two: 32 more MFMAs per product)
  // (split mode: windows at multiples of 32 — the transposed reads take whole 4-channel packs)
  constexpr bool HEADCOL = !BF && !SP;
  const WinGeom g = p.g;
  const int C = CT > 0 ? CT : g.C, d = CT > 0 ? CT / HEADS : p.d, ldt = CT > 0 ? wb_ldt<T>(CT) : p.ldt;
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int secsz = 64 * ldt;
  // sections: Q * scale (* log2 e) | K | V | dOut, 64 rows of ldt bytes each
  // (the dQ / dK / dV tiles are stored to HBM straight from the accumulator registers — lane = channel, register = token:
  // a store instruction covers 32 consecutive channels of two token rows — instead of through three more LDS sections and
  // a row copy-out: four sections instead of seven, so the fp32 kernel fits the LDS at C = 90 and 120 too, where it used to
  // hand over to the scalar kernel: 0.9 / 1.18 ms per launch)
  float* tab0 = reinterpret_cast<float*>(smem + 4 * secsz);   // [HEADS][15][TS]
  float* dtabL = tab0 + HEADS * 15 * TS;                       // [HEADS][15][TS]
  // then float4 stats[HEADS][64] {m, 1/l, delta, -}

  constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
  const float tabscale = L2D ? LOG2E : 1.0f;
  const float qscale = p.scale * tabscale;
  const int nW = g.nWh * g.nWw;
  const int nwin = g.B * nW;

  for (int i = tid; i < HEADS * 15 * TS; i += NTHREADS) dtabL[i] = 0.f;
  for (int i = tid; i < HEADS * 225; i += NTHREADS) {
    const int hd = i / 225, rem = i - hd * 225;
    const int dy = rem / 15, dx = rem - dy * 15;
    tab0[(hd * 15 + dy) * TS + dx] = p.table[rem * HEADS + hd] * tabscale;
  }
  {  // zero the pad columns of the four input sections once
    const int padw = (ldt - C * (int)sizeof(T)) / 4;
    for (int idx = tid; idx < 4 * 64 * padw; idx += NTHREADS) {
      const int row = idx / padw, w = idx - row * padw;
      *reinterpret_cast<uint32_t*>(smem + (size_t)row * ldt + C * sizeof(T) + 4 * w) = 0u;
    }
  }
  const int secb = C * (int)sizeof(T);
  const int cps = secb / GRAN;
  const int per_row = 3 * cps;

  const int thr = g.ws - g.shift;
  const float NEG = -100.0f * tabscale;

  // d(table): dS^T summed element-wise over all windows of this workgroup in registers (LDS float atomics
  // cost ~180 cycles per wave-instruction: 230 of 370 us when issued per window); binned once at the end.
  f32x16 Dsum[NU][2];
#pragma unroll
  for (int ui = 0; ui < NU; ++ui)
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int v = 0; v < 16; ++v) Dsum[ui][kt][v] = 0.f;
  // unit ui of this wave: tile t (32 tokens) of head hd
#define RDST_WB_UNIT(ui)                                                                       \
  const int u = wv + NWAVES * (ui);                                                            \
  const int t = u & 1, hd = u >> 1;                                                            \
  const int c_lo = hd * d, c_hi = c_lo + d;                                                    \
  const int t_lo = c_lo / KP, t_hi = (c_hi - 1) / KP;                                          \
  const int ct_lo = c_lo / 32, ct_hi = (c_hi - 1) / 32;                                        \
  const int yl = t * 4 + (r >> 3); /* window coords of this lane's token (tile t): (yl, xl) */ \
  const bool fyl = yl < thr, fxl = xl < thr;

  for (int win = blockIdx.x; win < nwin; win += gridDim.x) {
    const int b = win / nW, wi = win - b * nW;
    const int wr = wi / g.nWw, wc = wi - wr * g.nWw;
    // dqkv rows of the lane's 16 accumulator rows of tile t (token (y, x) = (4 t + (v >> 2), 4 h + (v & 3)) of the window): the image row
    // part is WAVE-UNIFORM (scalar registers), the column part one offset per (v & 3) — computed once per unit instead of a 64-bit
    // token-index product in front of every one of the 16 x 3 (x 2) stores (64 v_mad_u64_u32 + 64 v_mul_lo_u32 per pass in the ISA)
    auto out_rows = [&](int t, T* (&rowp)[4], int (&cofs)[4]) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int ry = wr * 8 + t * 4 + j + g.shift;
        ry = ry >= g.H ? ry - g.H : ry;
        rowp[j] = p.dqkv + ((int64_t)b * g.H + ry) * g.W * p.ldq;
        int cx = wc * 8 + j + 4 * h + g.shift;
        cx = cx >= g.W ? cx - g.W : cx;
        cofs[j] = cx * (int)p.ldq;
      }
    };
    // the lane index is made opaque per window: the ~200 LDS positions of a unit are window-invariant, and hipcc
    // otherwise computes each of them (section + row x stride + lane part as its own register, not as an instruction
    // offset) in front of the window loop and spills them; recomputed per window they are base + immediate
    int tidw = tid;
    asm volatile("" : "+v"(tidw));
    const int lane = tidw & 63, r = lane & 31, h = lane >> 5;
    const int xl = r & 7;
    char* const smw = smem;
    char* const Qs = smw;
    char* const Ks = Qs + secsz;
    char* const Vs = Ks + secsz;
    char* const Os = Vs + secsz;
    const float* const tabL = reinterpret_cast<const float*>(Os + secsz);
    float4* const stats = reinterpret_cast<float4*>(const_cast<float*>(tabL) + 2 * HEADS * 15 * TS);
    // ---- HBM -> LDS: qkv rows (3 sections, Q scaled) and dOut rows ---------------------------------
    {
      CH regs[MAXR][ITERS];
      CH rego[MAXR];
#pragma unroll
      for (int i = 0; i < MAXR; ++i) {
        int row = wv + NWAVES * i;
        row = row < 64 ? row : 63;
        const int64_t tok = win_token8(b, wr, wc, row, g);
        const char* src = reinterpret_cast<const char*>(p.qkv + tok * p.ld);
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
          int c = lane + 64 * it;
          c = c < per_row ? c : per_row - 1;
          regs[i][it] = *reinterpret_cast<const CH*>(src + (size_t)c * GRAN);
        }
        const int co = lane < cps ? lane : cps - 1;
        rego[i] = *reinterpret_cast<const CH*>(reinterpret_cast<const char*>(p.dout + tok * p.ldd) + (size_t)co * GRAN);
      }
      __syncthreads();  // every wave is done with the previous window's sections
#pragma unroll
      for (int i = 0; i < MAXR; ++i) {
        const int row = wv + NWAVES * i;
        if (row < 64) {
#pragma unroll
          for (int it = 0; it < ITERS; ++it) {
            const int c = lane + 64 * it;
            if (c < per_row) {
              const int off = c * GRAN;
              const int sec = (off >= secb) + (off >= 2 * secb);
              CH v = regs[i][it];
              if (sec == 0) {  // Q <- Q * scale (* log2 e)
                uint32_t w[GRAN / 4];
                __builtin_memcpy(w, &v, GRAN);
#pragma unroll
                for (int e = 0; e < GRAN / 4; ++e) {
                  if (BF) w[e] = pack_bf16x2(bf16lo(w[e]) * qscale, bf16hi(w[e]) * qscale);
                  else w[e] = __float_as_uint(__uint_as_float(w[e]) * qscale);
                }
                __builtin_memcpy(&v, w, GRAN);
              }
              if constexpr (SP && GRAN == 16) {   // a 16-byte chunk IS a pack of 4 channels: split on the way in (no second pass over the LDS)
                float f[4];
                __builtin_memcpy(f, &v, 16);
                const Pack16 pk = MM::pack_op(f);
                __builtin_memcpy(&v, &pk, 16);
              }
              *reinterpret_cast<CH*>(smw + sec * (secsz - secb) + row * ldt + off) = v;
            }
          }
          if (lane < cps) {
            CH v = rego[i];
            if constexpr (SP && GRAN == 16) {
              float f[4];
              __builtin_memcpy(f, &v, 16);
              const Pack16 pk = MM::pack_op(f);
              __builtin_memcpy(&v, &pk, 16);
            }
            *reinterpret_cast<CH*>(Os + row * ldt + lane * GRAN) = v;
          }
        }
      }
    }
    __syncthreads();
    if constexpr (SP && GRAN != 16) {   // (C = 90: 8-byte chunks, packs straddle them)
      const int ppr = (C + 3) / 4;   // packs per row; the channels past C inside the last one are zeroed (they held lo halves)
      for (int i = tid; i < 4 * 64 * ppr; i += NTHREADS) {
        const int row = i / ppr, pk = i - row * ppr;
        Pack16* q = reinterpret_cast<Pack16*>(smw + (size_t)row * ldt + pk * 16);
        float f[4];
        MM::unpack(*q, f);
#pragma unroll
        for (int e = 0; e < 4; ++e) f[e] = (4 * pk + e < C) ? f[e] : 0.f;
        *q = MM::pack_op(f);
      }
      __syncthreads();
    }

    const bool mrow = g.shift > 0 && wr == g.nWh - 1, mcol = g.shift > 0 && wc == g.nWw - 1;
    const bool masked = __builtin_amdgcn_readfirstlane((int)(mrow || mcol)) != 0;
    // the shift mask as a comparison of region numbers (2 [row >= thr] + [column >= thr], each only where the window is the
    // last of its row / column): mr / mc are made opaque per window, otherwise hipcc turns the 64 per-element predicates
    // into window-invariant lane masks, computes all of them in front of the window loop and spills them
    int mr = mrow ? 1 : 0, mc = mcol ? 1 : 0;
    asm volatile("" : "+v"(mr), "+v"(mc));
    int rx4[4];   // column part of the region of token column (v & 3) + 4 h
#pragma unroll
    for (int j = 0; j < 4; ++j) rx4[j] = mc & (int)(j + 4 * h >= thr);

    // ================= pass T: keys on registers, query (tile t) on the lane ======================
#pragma unroll
    for (int ui = 0; ui < NU; ++ui) {
      if (NU > 1 && wv + NWAVES * ui >= NUNITS) break;
      if (RDST_DBGV(p.dbg) & 2) break;
      RDST_WB_UNIT(ui)
      f32x16 X[2], D[2];
      const float* tb = tabL + hd * 15 * TS + (yl + 7) * TS + (xl + 7) - 4 * h;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          X[kt][v] = tb[-((kt * 4 + (v >> 2)) * TS + (v & 3))];  // bias = initial accumulator
          D[kt][v] = 0.f;
        }
      for (int ts = t_lo; ts <= t_hi; ++ts) {
        const int c0 = ts * KP + h * HP;
        const Pack16 qb = masked_pack<T, SP>(Qs + (size_t)(t * 32 + r) * ldt, c0, c_lo, c_hi);
        const Pack16 ob = masked_pack<T, SP>(Os + (size_t)(t * 32 + r) * ldt, c0, c_lo, c_hi);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
          const Pack16 ka = *reinterpret_cast<const Pack16*>(Ks + (size_t)(kt * 32 + r) * ldt + (size_t)c0 * sizeof(T));
          const Pack16 va = *reinterpret_cast<const Pack16*>(Vs + (size_t)(kt * 32 + r) * ldt + (size_t)c0 * sizeof(T));
          MM::mma(X[kt], ka, qb);
          MM::mma(D[kt], va, ob);
        }
      }
      if (masked) {
        const int rl = 2 * (mr & (int)!fyl) + (mc & (int)!fxl);   // shift region of the lane's token
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int v = 0; v < 16; ++v) {
            const int yj = kt * 4 + (v >> 2);
            X[kt][v] += rl != 2 * (mr & (int)(yj >= thr)) + rx4[v & 3] ? NEG : 0.f;
          }
      }
      float m = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int v = 0; v < 16; ++v) m = fmaxf(m, X[kt][v]);
      m = fmaxf(m, __shfl_xor(m, 32, 64));
      float l = 0.f;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const float e = L2D ? __builtin_amdgcn_exp2f(X[kt][v] - m) : expf(X[kt][v] - m);
          X[kt][v] = e;
          l += e;
        }
      l += __shfl_xor(l, 32, 64);
      const float inv = 1.0f / l;
      float dl = 0.f;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          X[kt][v] *= inv;                       // P^T
          dl = fmaf(X[kt][v], D[kt][v], dl);
        }
      dl += __shfl_xor(dl, 32, 64);              // delta_i
      if (h == 0) stats[hd * 64 + t * 32 + r] = make_float4(m, inv, dl, 0.f);
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const float ds = X[kt][v] * (D[kt][v] - dl);  // dS^T
          D[kt][v] = ds;
          Dsum[ui][kt][v] += ds;
        }
      // dQ_h (tile t) = scale * (dS^T)^T K_h
      for (int ci = 0; ci < (HEADCOL ? 1 : ct_hi - ct_lo + 1); ++ci) {
        f32x16 acc;
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[v] = 0.f;
        const int col0 = HEADCOL ? c_lo : (ct_lo + ci) * 32;
        const int col = col0 + r;
        const bool colin = col >= c_lo && col < c_hi;
        acc_xt_b<T, SP>(acc, D, Ks, ldt, col0, colin, lane);
        if (colin) {
          T* rowp[4];
          int cofs[4];
          out_rows(t, rowp, cofs);
#pragma unroll
          for (int v = 0; v < 16; ++v)
            rowp[v >> 2][cofs[v & 3] + col] = from_f32<T>(acc[v] * p.scale);
        }
      }
    }
    __syncthreads();  // statistics of both query tiles are in LDS

    // ================= pass N: queries on registers, key (tile t) on the lane ======================
#pragma unroll
    for (int ui = 0; ui < NU; ++ui) {
      if (NU > 1 && wv + NWAVES * ui >= NUNITS) break;
      if (RDST_DBGV(p.dbg) & 4) break;
      RDST_WB_UNIT(ui)
      f32x16 Y[2], E[2];
      const float* tb = tabL + hd * 15 * TS + (7 - yl) * TS + (7 - xl) + 4 * h;
#pragma unroll
      for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          Y[it][v] = tb[(it * 4 + (v >> 2)) * TS + (v & 3)];
          E[it][v] = 0.f;
        }
      for (int ts = t_lo; ts <= t_hi; ++ts) {
        const int c0 = ts * KP + h * HP;
        const Pack16 kb = *reinterpret_cast<const Pack16*>(Ks + (size_t)(t * 32 + r) * ldt + (size_t)c0 * sizeof(T));
        const Pack16 vb = *reinterpret_cast<const Pack16*>(Vs + (size_t)(t * 32 + r) * ldt + (size_t)c0 * sizeof(T));
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const Pack16 qa = masked_pack<T, SP>(Qs + (size_t)(it * 32 + r) * ldt, c0, c_lo, c_hi);
          const Pack16 oa = masked_pack<T, SP>(Os + (size_t)(it * 32 + r) * ldt, c0, c_lo, c_hi);
          MM::mma(Y[it], qa, kb);
          MM::mma(E[it], oa, vb);
        }
      }
      if (masked) {
        const int rl = 2 * (mr & (int)!fyl) + (mc & (int)!fxl);
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
          for (int v = 0; v < 16; ++v) {
            const int yi = it * 4 + (v >> 2);
            Y[it][v] += rl != 2 * (mr & (int)(yi >= thr)) + rx4[v & 3] ? NEG : 0.f;
          }
      }
#pragma unroll
      for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const float4 st = stats[hd * 64 + it * 32 + acc_row(v, h)];
          const float pr = (L2D ? __builtin_amdgcn_exp2f(Y[it][v] - st.x) : expf(Y[it][v] - st.x)) * st.y;  // P
          Y[it][v] = pr;
          E[it][v] = pr * (E[it][v] - st.z);                                                                // dS
        }
      for (int ci = 0; ci < (HEADCOL ? 1 : ct_hi - ct_lo + 1); ++ci) {
        const int col0 = HEADCOL ? c_lo : (ct_lo + ci) * 32;
        const int col = col0 + r;
        const bool colin = col >= c_lo && col < c_hi;
        const float kfix = L2D ? LN2 : 1.0f;
        {
          f32x16 av;
#pragma unroll
          for (int v = 0; v < 16; ++v) av[v] = 0.f;
          acc_xt_b<T, SP>(av, Y, Os, ldt, col0, colin, lane);   // dV_h = P^T dO_h
          if (colin) {
            T* rowp[4];
            int cofs[4];
            out_rows(t, rowp, cofs);
#pragma unroll
            for (int v = 0; v < 16; ++v)
              rowp[v >> 2][cofs[v & 3] + 2 * C + col] = from_f32<T>(av[v]);
          }
        }
        {
          f32x16 ak;
#pragma unroll
          for (int v = 0; v < 16; ++v) ak[v] = 0.f;
          acc_xt_b<T, SP>(ak, E, Qs, ldt, col0, colin, lane);   // dK_h = dS^T Qs_h (Qs carries scale [* log2 e])
          if (colin) {
            T* rowp[4];
            int cofs[4];
            out_rows(t, rowp, cofs);
#pragma unroll
            for (int v = 0; v < 16; ++v)
              rowp[v >> 2][cofs[v & 3] + C + col] = from_f32<T>(ak[v] * kfix);
          }
        }
      }
    }
  }
#pragma unroll
  for (int ui = 0; ui < NU; ++ui) {
    if (NU > 1 && wv + NWAVES * ui >= NUNITS) break;
    const int xl = r & 7;
    RDST_WB_UNIT(ui)
    (void)c_hi; (void)t_lo; (void)t_hi; (void)ct_lo; (void)ct_hi; (void)fyl; (void)fxl;
    float* dtb = dtabL + hd * 15 * TS + (yl + 7) * TS + (xl + 7) - 4 * h;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int v = 0; v < 16; ++v) atomicAdd(&dtb[-((kt * 4 + (v >> 2)) * TS + (v & 3))], Dsum[ui][kt][v]);
  }
#undef RDST_WB_UNIT
  __syncthreads();
  float* my = p.slab + (int64_t)blockIdx.x * HEADS * 225;
  for (int i = tid; i < HEADS * 225; i += NTHREADS) {
    const int hh = i / 225, rem = i - hh * 225;
    const int dy = rem / 15, dx = rem - dy * 15;
    my[i] = dtabL[(hh * 15 + dy) * TS + dx];
  }
}

bool mfma_disabled() {
  static int v = -1;
  if (v < 0) {
    const char* e = rdst_dbg_getenv("RDST_DISABLE_MFMA");
    v = (e && e[0] == '1') ? 1 : 0;
  }
  return v == 1;
}

template <typename T>
int launch_bwd(const T* qkv, int64_t ld, const float* table, const T* dout, int64_t ldd, T* dqkv, int64_t ldq,
               float* slab, int slab_rows, const WinGeom& g, float scale, int* nslab, hipStream_t st) {
  const int d = g.C / g.heads;
  if (g.ws != 8 || g.heads != HEADS || d > 32 || g.C > 128 || g.mask) return RDST_ENOTSUP;
  WbArgs<T> p{};
  p.qkv = qkv; p.ld = ld; p.table = table; p.dout = dout; p.ldd = ldd; p.dqkv = dqkv; p.ldq = ldq; p.slab = slab;
  p.g = g; p.scale = scale; p.d = d;
  { const char* e = rdst_dbg_getenv("RDST_K2_DEBUG"); p.dbg = e ? atoi(e) : 0; }
  const int sec = g.C * (int)sizeof(T);
  int gran = 0;
  for (int gs = 16; gs >= 4; gs >>= 1) {
    const auto ok = [&](const void* q, int64_t l) { return ((uintptr_t)q % gs) == 0 && (l * (int64_t)sizeof(T)) % gs == 0; };
    if (sec % gs == 0 && ok(qkv, ld) && ok(dout, ldd) && ok(dqkv, ldq)) { gran = gs; break; }
  }
  if (!gran) return RDST_ENOTSUP;
  const int ldt = wb_ldt<T>(g.C);
  p.ldt = ldt;
  const size_t smem = (size_t)4 * 64 * ldt + (size_t)2 * HEADS * 15 * TS * 4 + (size_t)HEADS * 64 * 16;
  if (smem > 160 * 1024) return RDST_ENOTSUP;
  const int64_t nwin = (int64_t)g.B * g.nWh * g.nWw;
  int64_t grid = 256;
  if (grid > nwin) grid = nwin;
  if (grid > slab_rows) grid = slab_rows;
  *nslab = (int)grid;
  const int per_row = 3 * sec / gran;
  const int iters = (per_row + 63) / 64;
  constexpr int NWV = sizeof(T) == 4 ? 8 : NUNITS;
#define RDST_WB_LAUNCH(GR, IT, CT)                                                                                   \
  {                                                                                                                  \
    auto kern = wattn_bwd_mfma_kernel<T, GR, IT, NWV, CT>;                                                                 \
    if constexpr (sizeof(T) == 4)                                                                                    \
      if (rdst_split()) kern = wattn_bwd_mfma_kernel<T, GR, IT, NWV, CT, true>;                                      \
    if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * NWV), smem, st, p);                                        \
  }
  bool done = false;
  if constexpr (sizeof(T) == 4) {   // the fp32 parity mode's three widths, channel count at compile time
    done = true;
    if (g.C == 60 && gran == 16) RDST_WB_LAUNCH(16, 1, 60)
    else if (g.C == 90 && gran == 8) RDST_WB_LAUNCH(8, 3, 90)
    else if (g.C == 120 && gran == 16) RDST_WB_LAUNCH(16, 2, 120)
    else done = false;
  }
  if (done) {}
  else if (gran == 16 && iters == 1) RDST_WB_LAUNCH(16, 1, 0)
  else if (gran == 16 && iters == 2) RDST_WB_LAUNCH(16, 2, 0)
  else if (gran == 8 && iters == 1) RDST_WB_LAUNCH(8, 1, 0)
  else if (gran == 8 && iters <= 3) RDST_WB_LAUNCH(8, 3, 0)
  else if (gran == 4 && iters <= 3) RDST_WB_LAUNCH(4, 3, 0)
  else return RDST_ENOTSUP;
#undef RDST_WB_LAUNCH
  return rdst_launch_status("wattn_bwd_mfma");
}

}  // namespace

// slab: [>= 256][heads][225] floats; *nslab = number of slab rows written (to be reduced by the caller)
int wattn_bwd_mfma(const void* qkv, int64_t ld, const float* table, const void* dout, int64_t ldd, void* dqkv,
                   int64_t ldq, float* slab, int slab_rows, const WinGeom& g, float scale, int dtype, int* nslab,
                   hipStream_t st) {
  if (mfma_disabled()) return RDST_ENOTSUP;
  if (dtype == RDST_F32) {
    const int rc = wattn16_bwd_f32((const float*)qkv, ld, table, (const float*)dout, ldd, (float*)dqkv, ldq, slab, slab_rows, g,
                                   scale, nslab, st);   // 16x16 windows
    if (rc != RDST_ENOTSUP) return rc;
    return launch_bwd<float>((const float*)qkv, ld, table, (const float*)dout, ldd, (float*)dqkv, ldq, slab, slab_rows, g,
                             scale, nslab, st);
  }
#ifndef RDST_K2_DMA
#define RDST_K2_DMA 6   // bit mask over the head dims 10 / 15 / 20 (1 / 2 / 4): which widths take the round-4 re-cut (wattn_bwd_pair.hip:
                        // LDS-DMA ring of section sets, loader / storer waves, passes T / N without the P / dS images).  It hides the row
                        // traffic but is VALU-bound by its recomputed key-tile pass.  Same box, cold, "K2 + table reduce" per call at
                        // C = 60 / 90 / 120: 77.6-80.7 / 78.9 / 78.8 us against 76.0 / 80.5 / 82.5 for the kernel below; inside the step
                        // (rocprofv3, profiles/r04f_*, before the loaders' cheaper issue code) 65.7 / 69.9 / 69.1 against 64.6 / 70.0 / 72.7:
                        // C = 90 and C = 120 take it, C = 60 (where the old kernel's arithmetic is the cheaper one) does not
#endif
  if (g.heads == 6 && g.C % 6 == 0) {
    const int d6 = g.C / 6;
    const int bit = d6 == 10 ? 1 : d6 == 15 ? 2 : d6 == 20 ? 4 : 0;
    if (RDST_K2_DMA & bit) {
      const int rc = wattn_bwd_pair(qkv, ld, table, dout, ldd, dqkv, ldq, slab, slab_rows, g, scale, nslab, st);
      if (rc != RDST_ENOTSUP) return rc;
    }
  }
  {  // the compile-time-specialised kernel (6 heads of dim 10/15/20) where it applies
    const int rc = wattn_bwd_mfma_hd(qkv, ld, table, dout, ldd, dqkv, ldq, slab, slab_rows, g, scale, nslab, st);
    if (rc != RDST_ENOTSUP) return rc;
  }
  {  // 16x16 windows
    const int rc = wattn16_bwd_mfma(qkv, ld, table, dout, ldd, dqkv, ldq, slab, slab_rows, g, scale, nslab, st);
    if (rc != RDST_ENOTSUP) return rc;
  }
  return launch_bwd<bf16>((const bf16*)qkv, ld, table, (const bf16*)dout, ldd, (bf16*)dqkv, ldq, slab, slab_rows, g, scale,
                          nslab, st);
}
