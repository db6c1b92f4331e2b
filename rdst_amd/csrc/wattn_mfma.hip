// K1 on the gfx950 matrix cores: fused window attention forward for 8x8 windows (N = 64 tokens).
//
// One 4-wave workgroup per window.  The window's 64 token rows are read once from HBM (the cyclic
// shift and window partition are index math) and split into Q / K / V sections in LDS, each 16-B
// aligned with an odd number of 16-B slots per row (conflict-free ds_read_b128).  Head dims are
// 10 / 15 / 20 — never a multiple of the MFMA k-step — so heads are NOT extracted: the k-steps that
// overlap a head are run on full aligned packs and the Q pack is masked to the head's channels
// (zero x anything = 0), which costs a few v_and and wastes < 1 k-step per head.
//   wave w: query tile qt = w & 1 (32 queries), head group hg = w >> 1 (heads hg, hg+2, ...)
//   S^T tile [key j][query i] = K . Q^T   -> query on the lane, keys in the 16 accumulator registers,
//   so softmax is in-register (+ one cross-half shuffle), and P^T is already the A operand of the
//   next MFMA (O = (P^T)^T V, "accumulator as operand": bf16 packs straight from the registers, fp32
//   registers as they are).  V is the B operand read transposed from LDS (ds_read_b64_tr_b16 / element
//   reads in fp32), masked to the head's output columns.  O_h overwrites the dead Q_h channels in LDS;
//   at the end full rows go back to HBM coalesced (window reverse + un-shift = same index math).
// HBM traffic = the algorithmic 4*C*elt bytes per token (+ the bias table once per workgroup).
#include "common.h"
#include "wattn.h"
#include "mfma.h"
#include <stdlib.h>

namespace {

constexpr int TS = 24;  // LDS row stride of the relative-position table (15 used): conflict-free gathers

template <typename T>
struct WaArgs {
  const T* qkv; int64_t ld;
  const float* table;
  T* out; int64_t ldo;
  WinGeom g;
  float scale;
  int d;          // head dim
  int ldt;        // LDS row stride of each Q/K/V section, bytes
  int gran;       // copy granule in bytes (16, 8 or 4)
  int dbg;        // ablation switches (RDST_K1_DEBUG): 1 = skip compute, 2 = skip HBM loads, 4 = skip HBM stores
};

__device__ __forceinline__ uint32_t mask_bits_bf16(int c0, int lo, int hi) {
  // 2 bf16 per dword: element c0 (low half) and c0+1 (high half)
  return ((c0 >= lo && c0 < hi) ? 0x0000ffffu : 0u) | ((c0 + 1 >= lo && c0 + 1 < hi) ? 0xffff0000u : 0u);
}

template <int GRAN> struct Chunk;
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
template <> struct Chunk<16> { typedef u32x4_t type; };
template <> struct Chunk<8> { typedef u32x2_t type; };
template <> struct Chunk<4> { typedef uint32_t type; };

// Persistent: each workgroup walks windows win = blockIdx.x, +gridDim.x, ...  The relative-position
// table is staged once.  The copy plan (which (token row, byte offset) each thread moves, and where it
// lands in the Q/K/V sections) is window independent, so it is computed once with the integer
// divisions it needs and kept in registers; the NEXT window's rows are loaded into registers while
// the current window is being computed (HBM latency hides behind the MFMA/softmax work).
// NW waves: wave w owns query tile w & 1 and the heads (w >> 1), + NW / 2, ...  NW = 4 with up to three workgroups per
// CU; NW = 8 where only one workgroup fits the LDS (fp32, C = 90 / 120: four waves were ONE wave per SIMD, every LDS
// and matrix-core latency exposed).
// SP (fp32 rows): RDST_F32X3.  The K and V sections are split pack by pack after they are staged ([4 bf16 hi | 4 bf16 lo] per
// 4 channels, mfma.h); the Q section stays fp32 — finished heads write their output over their dead Q channels, which would
// tear the packs that straddle two heads — and a Q fragment is split when it is read (once per head and k-step).
template <typename T, int GRAN, int ITERS, int NW, bool SP = false>
__global__ void __launch_bounds__(64 * NW, NW == 4 ? (sizeof(T) == 2 ? 3 : 2) : 1) wattn_fwd_mfma_kernel(const WaArgs<T> p) {
  constexpr int NT = 64 * NW;      // threads
  constexpr int RW = 64 / NW;      // token rows a wave stages and copies out
  constexpr bool HEADCOL = sizeof(T) == 4 && !SP;   // fp32: the 32-column window of P.V starts at the head's first channel
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using MM = Mma<T, SP>;
  using CH = typename Chunk<GRAN>::type;
  constexpr int KP = MM::KP, HP = MM::HP;
  constexpr bool BF = sizeof(T) == 2;
  constexpr bool L2D = BF || SP;   // softmax in the log2 domain on the hardware exp2 (the split mode is not bit-exact fp32 anyway; exact fp32 keeps expf)
  const WinGeom g = p.g;
  const int C = g.C, heads = g.heads, d = p.d;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  const int ldt = p.ldt;
  char* Qs = smem;
  char* Ks = Qs + 64 * ldt;
  char* Vs = Ks + 64 * ldt;
  float* tabL = reinterpret_cast<float*>(Vs + 64 * ldt);  // [heads][15][TS]

  constexpr float LOG2E = 1.4426950408889634f;
  const float tabscale = L2D ? LOG2E : 1.0f;
  const int nW = g.nWh * g.nWw;
  const int nwin = g.B * nW;

  for (int i = tid; i < heads * 15 * 15; i += NT) {
    const int hd = i / 225, rem = i - hd * 225;
    const int dy = rem / 15, dx = rem - dy * 15;
    tabL[(hd * 15 + dy) * TS + dx] = p.table[rem * heads + hd] * tabscale;
  }
  {  // zero the pad columns [C*elt, ldt) of every section once: padded k-steps must read zeros
    const int padw = (ldt - C * (int)sizeof(T)) / 4;
    for (int idx = tid; idx < 3 * 64 * padw; idx += NT) {
      const int row = idx / padw, w = idx - row * padw;
      *reinterpret_cast<uint32_t*>(Qs + (size_t)row * ldt + C * sizeof(T) + 4 * w) = 0u;
    }
  }
  // copy geometry: a token row is 3 sections of `secb` bytes; wave w moves token rows 16w .. 16w+15,
  // lanes move GRAN-byte chunks of a row (row base = wave-uniform scalar math, no per-lane divisions)
  const int cps = C * (int)sizeof(T) / GRAN;  // chunks per section
  const int per_row = 3 * cps;
  const int secb = C * (int)sizeof(T);
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  int win = blockIdx.x;
  const int qt = wave & 1, hg = wave >> 1;
  const int yi = qt * 4 + (r >> 3), xi = r & 7;
  const int thr = g.ws - g.shift;
  const bool fyi = yi < thr, fxi = xi < thr;
  const float NEG = -100.0f * tabscale;
  const float qscale = p.scale * tabscale;

  for (; win < nwin; win += gridDim.x) {
    const int b = win / nW, wi = win - b * nW;
    const int wr = wi / g.nWw, wc = wi - wr * g.nWw;
    // token rows of this wave: window rows y = 2*wv, 2*wv+1 (8 tokens each).  Row index of token (y, x):
    // rowbase(y) + col(x), col(x) = c0 + x (- W when it wraps: only the last window column of a shifted block)
    const int c0 = wc * 8 + g.shift;
    constexpr int NY = RW / 8;   // window rows of this wave's tokens
    int64_t rbase[NY];
#pragma unroll
    for (int yy = 0; yy < NY; ++yy) {
      int rr = wr * 8 + wv * NY + yy + g.shift;
      if (rr >= g.H) rr -= g.H;
      rbase[yy] = ((int64_t)b * g.H + rr) * g.W;
    }
    constexpr int NPART = ITERS >= 3 ? 4 : 2;   // stage the wave's token rows in parts to bound the staging registers
    constexpr int RPP = RW / NPART;
#pragma unroll
    for (int part = 0; part < NPART; ++part) {
      CH regs[RPP][ITERS];
#pragma unroll
      for (int i = 0; i < RPP; ++i) {
        const int ri = part * RPP + i;          // 0..RW-1 within the wave
        int col = c0 + (ri & 7);
        if (col >= g.W) col -= g.W;
        const int64_t tok = (RDST_DBGV(p.dbg) & 2) ? 0 : rbase[ri >> 3] + col;
        const char* src = reinterpret_cast<const char*>(p.qkv + tok * p.ld);
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
          int c = lane + 64 * it;
          c = c < per_row ? c : per_row - 1;  // clamp instead of predicate: keeps the staging registers SROA-able
          regs[i][it] = *reinterpret_cast<const CH*>(src + (size_t)c * GRAN);
        }
      }
      if (part == 0) __syncthreads();  // previous window's O has been copied out; the tile may be overwritten
#pragma unroll
      for (int i = 0; i < RPP; ++i) {
        const int row = wv * RW + part * RPP + i;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
          const int c = lane + 64 * it;
          if (c < per_row) {
            const int off = c * GRAN;
            const int sec = (off >= secb) + (off >= 2 * secb);
            CH v = regs[i][it];
            if (sec == 0) {  // Q <- Q * scale (* log2 e): the bias can then be the initial accumulator
              uint32_t w[GRAN / 4];
              __builtin_memcpy(w, &v, GRAN);
#pragma unroll
              for (int e = 0; e < GRAN / 4; ++e) {
                if (BF) w[e] = pack_bf16x2(bf16lo(w[e]) * qscale, bf16hi(w[e]) * qscale);
                else w[e] = __float_as_uint(__uint_as_float(w[e]) * qscale);
              }
              __builtin_memcpy(&v, w, GRAN);
            }
            if constexpr (SP && GRAN == 16) {   // K / V: a 16-byte chunk is a pack of 4 channels, split on the way in
              if (sec != 0) {
                float f[4];
                __builtin_memcpy(f, &v, 16);
                const Pack16 pk = MM::pack_op(f);
                __builtin_memcpy(&v, &pk, 16);
              }
            }
            *reinterpret_cast<CH*>(smem + sec * (64 * ldt - secb) + row * ldt + off) = v;
          }
        }
      }
    }
    __syncthreads();
    if constexpr (SP && GRAN != 16) {
      const int ppr = (C + 3) / 4;   // packs per row; the channels past C inside the last one are zeroed (they held lo halves)
      for (int i = tid; i < 2 * 64 * ppr; i += NT) {
        const int row = i / ppr, pk = i - row * ppr;
        Pack16* q = reinterpret_cast<Pack16*>(Ks + (size_t)row * ldt + pk * 16);
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
    for (int hd = hg; hd < ((RDST_DBGV(p.dbg) & 1) ? 0 : heads); hd += NW / 2) {
      const int c_lo = hd * d, c_hi = c_lo + d;
      const int t_lo = c_lo / KP, t_hi = (c_hi - 1) / KP;
      f32x16 X[2];
      const float* tb = tabL + hd * 15 * TS + (yi + 7) * TS + (xi + 7) - 4 * h;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int v = 0; v < 16; ++v) X[kt][v] = tb[-((kt * 4 + (v >> 2)) * TS + (v & 3))];  // bias = initial accumulator
      for (int t = t_lo; t <= t_hi; ++t) {
        const int c0 = t * KP + h * HP;
        Pack16 qb = *reinterpret_cast<const Pack16*>(Qs + (size_t)(qt * 32 + r) * ldt + (size_t)c0 * sizeof(T));
        if (BF) {
#pragma unroll
          for (int e = 0; e < 4; ++e) qb.w[e] &= mask_bits_bf16(c0 + 2 * e, c_lo, c_hi);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) qb.w[e] = (c0 + e >= c_lo && c0 + e < c_hi) ? qb.w[e] : 0u;
        }
        qb = MM::op(qb);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
          const Pack16 ka = *reinterpret_cast<const Pack16*>(Ks + (size_t)(kt * 32 + r) * ldt + (size_t)c0 * sizeof(T));
          MM::mma(X[kt], ka, qb);
        }
      }
      // X[kt][v] = logit (log2 domain in bf16 mode) of key j = kt*32 + acc_row(v,h), query i = qt*32 + r
      if (masked) {  // wave-uniform: only the last window row / column of a shifted block
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int v = 0; v < 16; ++v) {
            const int yj = kt * 4 + (v >> 2), xj = (v & 3) + 4 * h;
            const bool dyf = mrow & (fyi != (yj < thr)), dxf = mcol & (fxi != (xj < thr));
            X[kt][v] += (dyf | dxf) ? NEG : 0.f;
          }
      }
      float m = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int v = 0; v < 16; ++v) m = fmaxf(m, X[kt][v]);
      m = fmaxf(m, __shfl_xor(m, 32, 64));
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      f32x2 l2 = {0.f, 0.f};
      const f32x2 m2 = {m, m};
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int v = 0; v < 16; v += 2) {
          f32x2 x2 = {X[kt][v], X[kt][v + 1]};
          x2 -= m2;                                   // v_pk_add_f32
          f32x2 e2;
          e2.x = L2D ? __builtin_amdgcn_exp2f(x2.x) : expf(x2.x);
          e2.y = L2D ? __builtin_amdgcn_exp2f(x2.y) : expf(x2.y);
          l2 += e2;
          X[kt][v] = e2.x;
          X[kt][v + 1] = e2.y;
        }
      float l = l2.x + l2.y;
      l += __shfl_xor(l, 32, 64);
      const float inv = 1.0f / l;
      const f32x2 inv2 = {inv, inv};
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int v = 0; v < 16; v += 2) {
          f32x2 p2 = {X[kt][v], X[kt][v + 1]};
          p2 *= inv2;                                  // v_pk_mul_f32
          X[kt][v] = p2.x;
          X[kt][v + 1] = p2.y;
        }

      // O_h = P V_h over the column tiles that overlap the head
      const int ct_lo = c_lo / 32, ct_hi = (c_hi - 1) / 32;
      for (int ci = 0; ci < (HEADCOL ? 1 : ct_hi - ct_lo + 1); ++ci) {
        const int ct = ct_lo + ci;
        f32x16 acc;
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[v] = 0.f;
        const int col = (HEADCOL ? c_lo : ct * 32) + r;
        const bool colin = col >= c_lo && col < c_hi;
        const uint32_t cmask = colin ? 0xffffffffu : 0u;
        if constexpr (SP) {
          // the V rows are packs [4 hi | 4 lo] of 4 channels: a transposed read takes the hi (8 bytes on: the lo) halves of
          // 4 key rows x 16 channels; a k-step is 8 keys: accumulator registers 4s .. 4s + 3 of both lane halves
          const int gq = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
          const int colB = ct * 32 + 16 * (gq & 1) + 4 * pp;
          typedef __attribute__((address_space(3))) s16x4_t* lds_p;
#pragma unroll
          for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
              const float f[4] = {X[kt][4 * s4], X[kt][4 * s4 + 1], X[kt][4 * s4 + 2], X[kt][4 * s4 + 3]};
              const Pack16 a = split_pack4(f);
              const char* bp = Vs + (size_t)(kt * 32 + 8 * s4 + 4 * h + q) * ldt + colB * 4;
              const s16x4_t b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)bp);
              const s16x4_t b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(bp + 8));
              const uint2 u0 = __builtin_bit_cast(uint2, b0), u1 = __builtin_bit_cast(uint2, b1);
              Pack16 bb;
              bb.w[0] = u0.x & cmask; bb.w[1] = u0.y & cmask; bb.w[2] = u1.x & cmask; bb.w[3] = u1.y & cmask;
              MM::mma(acc, a, bb);
            }
        } else if constexpr (BF) {
          const int gq = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
          const int colB = ct * 32 + 16 * (gq & 1) + 4 * pp;
#pragma unroll
          for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
              Pack16 a;
#pragma unroll
              for (int e = 0; e < 4; ++e) a.w[e] = pack_bf16x2(X[kt][8 * s + 2 * e], X[kt][8 * s + 2 * e + 1]);
              // element jj of lane half h is key 16s + 8(jj>>2) + 4h + (jj&3) of the tile
              const int rowb = kt * 32 + 16 * s + 4 * h + q;
              typedef __attribute__((address_space(3))) s16x4_t* lds_p;
              const s16x4_t b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(Vs + (size_t)rowb * ldt + colB * 2));
              const s16x4_t b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(Vs + (size_t)(rowb + 8) * ldt + colB * 2));
              const uint2 u0 = __builtin_bit_cast(uint2, b0), u1 = __builtin_bit_cast(uint2, b1);
              Pack16 bb;
              bb.w[0] = u0.x & cmask; bb.w[1] = u0.y & cmask; bb.w[2] = u1.x & cmask; bb.w[3] = u1.y & cmask;
              MM::mma(acc, a, bb);
            }
        } else {
          const float* Vf = reinterpret_cast<const float*>(Vs);
          const int ldv = ldt / 4;
#pragma unroll
          for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
              const float bv = Vf[(kt * 32 + acc_row(v, h)) * ldv + col];
              acc = __builtin_amdgcn_mfma_f32_32x32x2f32(X[kt][v], colin ? bv : 0.f, acc, 0, 0, 0);
            }
        }
        // O tile rows = queries qt*32 + acc_row(v,h), column = col: overwrite the dead Q_h channels
        if (colin) {
#pragma unroll
          for (int v = 0; v < 16; ++v) {
            T* dst = reinterpret_cast<T*>(Qs + (size_t)(qt * 32 + acc_row(v, h)) * ldt) + col;
            *dst = from_f32<T>(acc[v]);
          }
        }
      }
    }
    __syncthreads();
    // LDS (Q section now holds O) -> global rows: wave w copies rows w, w+4, ...; no divisions
    if (!(RDST_DBGV(p.dbg) & 4)) {
#pragma unroll
      for (int i = 0; i < RW; ++i) {
        int col = c0 + (i & 7);
        if (col >= g.W) col -= g.W;
        const int64_t tok = rbase[i >> 3] + col;
        char* dst = reinterpret_cast<char*>(p.out + tok * p.ldo);
        const char* src = Qs + (size_t)(wv * RW + i) * ldt;
        if (lane < cps) *reinterpret_cast<CH*>(dst + (size_t)lane * GRAN) = *reinterpret_cast<const CH*>(src + (size_t)lane * GRAN);
      }
    }
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

int pick_gran(uintptr_t a, uintptr_t b, int64_t lda_bytes, int64_t ldb_bytes, int sec_bytes) {
  for (int gsz = 16; gsz >= 4; gsz >>= 1)
    if (a % gsz == 0 && b % gsz == 0 && lda_bytes % gsz == 0 && ldb_bytes % gsz == 0 && sec_bytes % gsz == 0) return gsz;
  return 0;
}

template <typename T>
int launch_fwd(const T* qkv, int64_t ld, const float* table, T* out, int64_t ldo, const WinGeom& g, float scale,
               hipStream_t st) {
  const int d = g.C / g.heads;
  if (g.ws != 8 || d > 32 || g.C > 128 || g.mask) return RDST_ENOTSUP;
  WaArgs<T> p{};
  p.qkv = qkv; p.ld = ld; p.table = table; p.out = out; p.ldo = ldo; p.g = g; p.scale = scale; p.d = d;
  { const char* e = rdst_dbg_getenv("RDST_K1_DEBUG"); p.dbg = e ? atoi(e) : 0; }
  const int sec = g.C * (int)sizeof(T);
  p.gran = pick_gran((uintptr_t)qkv, (uintptr_t)out, ld * (int64_t)sizeof(T), ldo * (int64_t)sizeof(T), sec);
  if (!p.gran) return RDST_ENOTSUP;
  // section row stride: whole k-steps, 16-B aligned, odd number of 16-B slots
  int ldt = ((sec + 31) / 32) * 32;
  if ((ldt / 16) % 2 == 0) ldt += 16;
  p.ldt = ldt;
  const size_t smem = (size_t)3 * 64 * ldt + (size_t)g.heads * 15 * TS * 4;
  if (smem > 160 * 1024) return RDST_ENOTSUP;
  const int64_t nwin = (int64_t)g.B * g.nWh * g.nWw;
  const int per_row = 3 * sec / p.gran;
  const int iters = (per_row + 63) / 64;
  int wg_per_cu = (int)((160 * 1024) / smem);
  if (wg_per_cu > 4) wg_per_cu = 4;
  if (wg_per_cu < 1) wg_per_cu = 1;
  int64_t grid = 256 * wg_per_cu;
  if (grid > nwin) grid = nwin;
#define RDST_WA_LAUNCH1(GR, KM, NWV)                                                                                  \
  {                                                                                                                  \
    auto kern = wattn_fwd_mfma_kernel<T, GR, KM, NWV>;                                                               \
    if constexpr (sizeof(T) == 4)                                                                                    \
      if (rdst_split()) kern = wattn_fwd_mfma_kernel<T, GR, KM, NWV, true>;                                          \
    if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * NWV), smem, st, p);                                     \
  }
#define RDST_WA_LAUNCH(GR, KM)                                                                                        \
  {                                                                                                                  \
    if (sizeof(T) == 4 && wg_per_cu == 1) RDST_WA_LAUNCH1(GR, KM, 8) else RDST_WA_LAUNCH1(GR, KM, 4)                 \
  }
  if (p.gran == 16 && iters == 1) RDST_WA_LAUNCH(16, 1)
  else if (p.gran == 16 && iters == 2) RDST_WA_LAUNCH(16, 2)
  else if (p.gran == 8 && iters == 1) RDST_WA_LAUNCH(8, 1)
  else if (p.gran == 8 && iters <= 3) RDST_WA_LAUNCH(8, 3)
  else if (p.gran == 4 && iters <= 3) RDST_WA_LAUNCH(4, 3)
  else return RDST_ENOTSUP;
#undef RDST_WA_LAUNCH
#undef RDST_WA_LAUNCH1
  return rdst_launch_status("wattn_fwd_mfma");
}

}  // namespace

int wattn_fwd_mfma(const void* qkv, int64_t ld, const float* table, void* out, int64_t ldo, const WinGeom& g,
                   float scale, int dtype, hipStream_t st) {
  if (mfma_disabled()) return RDST_ENOTSUP;
  if (dtype == RDST_F32) {
    const int rc = wattn16_fwd_f32((const float*)qkv, ld, table, (float*)out, ldo, g, scale, st);   // 16x16 windows
    if (rc != RDST_ENOTSUP) return rc;
    return launch_fwd<float>((const float*)qkv, ld, table, (float*)out, ldo, g, scale, st);
  }
  {  // the compile-time-specialised kernel (6 heads of dim 10/15/20) where it applies
    const int rc = wattn_fwd_mfma_hd(qkv, ld, table, out, ldo, g, scale, st);
    if (rc != RDST_ENOTSUP) return rc;
  }
  {  // 16x16 windows
    const int rc = wattn16_fwd_mfma(qkv, ld, table, out, ldo, g, scale, st);
    if (rc != RDST_ENOTSUP) return rc;
  }
  return launch_fwd<bf16>((const bf16*)qkv, ld, table, (bf16*)out, ldo, g, scale, st);
}
