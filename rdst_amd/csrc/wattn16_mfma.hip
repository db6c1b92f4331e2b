// Window attention for 16x16 windows (N = 256 tokens per window) on the matrix cores: bf16, 6 heads of dim
// 10 / 15 / 20 (the RDST widths C = 60 / 90 / 120 with window_size 16 — BASELINE.json configs[3], the SwinIR-style
// large-window configuration; reference: networks/swin_transformer_sr.py:110-141 with N = 256, mask :211-232).
//
// What changes against the 8x8 kernels (wattn_mfma_hd.hip): a query now has 256 keys, so
//   * the Q/K/V rows of a whole window do not fit the LDS for all heads (256 x 3 x 240 B = 184 KB at C = 120):
//     a workgroup owns (window, head PAIR) — 40 / 60 / 80 B pieces of each row section, 37 - 61 KB of LDS — and the
//     three workgroups of a window are placed on the same XCD so the shared cache lines are fetched from HBM once;
//   * a wave owns a query tile (32 queries on the lanes) and ALL 256 keys of a head live in its accumulator
//     registers (8 tiles x 16 registers): S^T = K.Q^T with bias/scale as the initial accumulator, the shift mask as one
//     more k-step of one-hot region vectors, softmax entirely in registers (no online rescaling: every key is
//     there), P^T repacked as the B operand of O^T = V^T.P^T; 4 waves x 2 query tiles, two workgroups per CU;
//   * the relative-position table of a head (31 x 31) is staged REVERSED in both directions, so that the bias of
//     key (yj, xj) for the lane's query sits at lane base + compile-time offset, in two copies shifted by one float
//     (every lane reads aligned 8-byte pairs).
// The kernel is bound by the softmax on the vector ALUs (256 x 256 x 6 exponentials per window), not by HBM.
#include "wattn_hd.h"
#include "wattn16.h"

namespace wahd {
template <> struct Chunk<4> { typedef uint32_t type; };   // one head of dim 10: 20-byte pieces, rows only dword aligned
}

namespace {
using namespace wahd;
using namespace w16c;

struct W16Args {
  const bf16* qkv; int64_t ld;
  const float* table;
  bf16* out; int64_t ldo;
  const bf16* dout; int64_t ldd;     // backward only
  bf16* dqkv; int64_t ldq;
  float* slab;                        // backward: [window][heads][961] partial d(table)
  WinGeom g;
  float scale;
  // round 5: the forward can leave its row statistics, nlse[token][head] = -(scale2 . max + log2 sum) (so that
  // exp2(scale2 . S + nlse) IS the attention weight); with them and the attention output the backward's first pass needs no
  // row reductions (wattn16_bwd2_kernel)
  float* nlse_out;                    // forward: [B H W][6] or NULL
  const float* nlse; const bf16* o; int64_t ldo2;   // backward v2: the statistics and the forward's output rows, or NULL
};

// HPG_ = 2: a workgroup owns a head PAIR (every kernel but wattn16_bwd1_kernel); HPG_ = 1: a single head (D = 10 / 20 only: the
// pieces of a head of dim 15 start on 2-byte boundaries)
template <int D, int HPG_ = 2>
struct W16 {
  static constexpr int HEADS = 6, HPG = HPG_, NG = HEADS / HPG;
  static constexpr int C = HEADS * D;
  static constexpr int GC = HPG * D;                 // channels of a head pair (of the head)
  // bytes of its piece of a row section: 40 / 60 / 80 (one head: 20 / 32 / 40 — a head of dim 15 starts on a 2-byte boundary when
  // its index is odd: its piece is the dword-aligned 32-byte window around its 30 bytes, one channel of a neighbour included)
  static constexpr int PB = (HPG == 1 && D == 15) ? 32 : GC * 2;
  static constexpr int GRAN = HPG == 2 ? (D == 10 ? 8 : D == 15 ? 12 : 16) : (D == 20 ? 8 : 4);  // chunk size (D = 15: 12-B chunks, dword aligned)
  static constexpr int CPS = PB / GRAN;              // chunks per piece
  static constexpr int LDT = HPG == 2 ? (D == 10 ? 48 : 80) : (D == 20 ? 48 : 32);   // LDS row stride (pairs: odd number of 16-B slots)
  static constexpr int SEC = 256 * LDT;              // bytes of one staged section
  static constexpr int TROW = 32;                    // floats per staged table row (31 used)
  // floats per head and copy: 31 rows + 16 floats of padding, so that the copies lie 16 banks apart (mod 64).  A 32-lane
  // ds_read_b64 group holds two query (key) rows, 32 floats = 32 banks apart, and in each the even lanes read copy 0 and the
  // odd lanes copy 1: with the copies 32 banks apart as well, (row 0, odd) met (row 1, even) on the same 16 banks: 2-way
  // on every bias read (tools/lds_banks.py)
  static constexpr int TABF = 31 * TROW + 16;
  static constexpr int NKS = (GC + 15) / 16;         // k-steps over the pair's channels
};

// workgroup -> (window, head pair): the three pairs of a window run on the same XCD (blockIdx round-robins over 8 XCDs)
__device__ __forceinline__ void w16_locate(int nwin, int& win, int& grp) {
  const int b = blockIdx.x;
  if ((nwin & 7) == 0) {
    const int xcd = b & 7, slot = b >> 3;
    grp = slot % 3;
    win = (slot / 3) * 8 + xcd;
  } else {
    grp = b % 3;
    win = b / 3;
  }
}

struct W16Ctx {
  float* lse;               // forward: the lane's slot nlse_out[token][pair's first head], or NULL
  lds_cp Qp, Kp, Vp, Orow;
  const LDS_AS f32x2* tb;   // lane base into the reversed table copy of its parity (head 0 of the pair)
  int h, r;
  bool masked, mrow, mcol;
  int thr, qt;
  float scale2;
  uint32_t cbits;
};

__device__ __forceinline__ void onehot4(int reg, uint32_t v, int h, Pack16& q) {   // k = 0..3 of lane half 0
  q.w[0] = h ? 0u : ((reg == 0 ? v : 0u) | (reg == 1 ? v << 16 : 0u));
  q.w[1] = h ? 0u : ((reg == 2 ? v : 0u) | (reg == 3 ? v << 16 : 0u));
  q.w[2] = 0u;
  q.w[3] = 0u;
}

// S'^T tiles of one (query tile, head): X[kt][v] = q.k + bias/scale (+ 100/scale where the shift regions agree);
// key j = 32 kt + acc_row(v, h) = (yj = 2 kt + (v >> 3), xj = 8 ((v >> 2) & 1) + 4 h + (v & 3)), query = the lane's.
template <int D, int HL>
__device__ __forceinline__ void w16_scores(f32x16 (&X)[8], const W16Ctx& c) {
  using CF = W16<D>;
  constexpr int ldt = CF::LDT;
  constexpr int c_lo = HL * D, c_hi = c_lo + D;
  constexpr int t_lo = c_lo / 16, t_hi = (c_hi - 1) / 16;
  const LDS_AS f32x2* tbh = c.tb + HL * (2 * CF::TABF / 2);
#pragma unroll
  for (int kt = 0; kt < 8; ++kt)
#pragma unroll
    for (int v = 0; v < 16; v += 2) {
      const f32x2 b2 = lds_read_f32x2(tbh + ((2 * kt + (v >> 3)) * CF::TROW + 8 * ((v >> 2) & 1) + (v & 3)) / 2);
      X[kt][v] = b2.x;
      X[kt][v + 1] = b2.y;
    }
#pragma unroll
  for (int t = t_lo; t <= t_hi; ++t) {
    Pack16 qb = lds_pack(c.Qp + t * 32);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t mA = qmask_bits(t * 16 + 2 * e, c_lo, c_hi), mB = qmask_bits(t * 16 + 8 + 2 * e, c_lo, c_hi);
      qb.w[e] &= c.h ? mB : mA;
    }
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
      const Pack16 ka = lds_pack(c.Kp + kt * 32 * ldt + t * 32);
      Mma<bf16>::mma(X[kt], ka, qb);
    }
  }
  if (c.masked) {   // wave-uniform: last window row / column of a shifted block
    const int xi = c.r & 15, yq = 2 * c.qt + (c.r >> 4);
    const int rx = (c.mcol && xi >= c.thr) ? 1 : 0;   // r & 15 is also the key column of row r of a key tile
    Pack16 mQ;
    onehot4(2 * ((c.mrow && yq >= c.thr) ? 1 : 0) + rx, c.cbits, c.h, mQ);
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
      Pack16 mK;
      onehot4(2 * ((c.mrow && 2 * kt + (c.r >> 4) >= c.thr) ? 1 : 0) + rx, 0x3f80u, c.h, mK);
      Mma<bf16>::mma(X[kt], mK, mQ);
    }
  }
}

// One head of one query tile, forward: softmax over the 256 keys in registers, O^T = V^T.P^T, rows [c_lo, c_hi) of the
// result overwrite the (dead) Q channels of the wave's own query rows.
template <int D, int HL>
__device__ __forceinline__ void w16_head_fwd(const W16Ctx& c) {
  using CF = W16<D>;
  constexpr int ldt = CF::LDT;
  constexpr int c_lo = HL * D, c_hi = c_lo + D;
  constexpr int RL = c_lo & ~3;   // first channel row of the output tile (8-byte aligned for the transposed reads)
  f32x16 X[8];
  w16_scores<D, HL>(X, c);
  float m = X[0][0];
#pragma unroll
  for (int kt = 0; kt < 8; ++kt)
#pragma unroll
    for (int v = 0; v < 16; ++v) m = __builtin_fmaxf(m, X[kt][v]);
  m = half_swap_max(m);
  const float nm = -c.scale2 * m;
  float l0 = 0.f, l1 = 0.f;
  f32x16 acc;
#pragma unroll
  for (int v = 0; v < 16; ++v) acc[v] = 0.f;
  typedef LDS_AS s16x4_t* lds_tr_p;
#pragma unroll
  for (int kt = 0; kt < 8; ++kt) {
#pragma unroll
    for (int v = 0; v < 16; v += 2) {
      const float e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(X[kt][v], c.scale2, nm));
      const float e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(X[kt][v + 1], c.scale2, nm));
      l0 += e0;
      l1 += e1;
      X[kt][v] = e0;
      X[kt][v + 1] = e1;
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {   // un-normalised P^T as the B operand: element jj of lane half h is key 16s + 8(jj>>2) + 4h + (jj&3)
      Pack16 pb;
#pragma unroll
      for (int e = 0; e < 4; ++e) pb.w[e] = pack_bf16x2(X[kt][8 * s + 2 * e], X[kt][8 * s + 2 * e + 1]);
      const lds_cp vb = c.Vp + RL * 2 + (kt * 32 + 16 * s) * ldt;
      const s16x4_t b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_p)(vb));
      const s16x4_t b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_p)(vb + 8 * ldt));
      const u32x2_t u0 = __builtin_bit_cast(u32x2_t, b0), u1 = __builtin_bit_cast(u32x2_t, b1);
      Pack16 va;
      va.w[0] = u0.x; va.w[1] = u0.y; va.w[2] = u1.x; va.w[3] = u1.y;
      Mma<bf16>::mma(acc, va, pb);   // rows = channels RL.. (V^T), cols = queries
    }
  }
  const float inv = __builtin_amdgcn_rcpf(half_swap_sum(l0 + l1));
  if (c.lse && c.h == 0) c.lse[HL] = nm + __builtin_amdgcn_logf(inv);   // (what pass 1 of the backward would compute)
  store_tile_rows<RL, c_lo, c_hi>(c.Orow, acc, inv, c.h);
}

template <int D>
__global__ void __launch_bounds__(256, 2) wattn16_fwd_kernel(const W16Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using CF = W16<D>;
  using CH = typename Chunk<CF::GRAN>::type;
  constexpr int ldt = CF::LDT, GRAN = CF::GRAN, CPS = CF::CPS, CPR = 3 * CPS, RPI = 64 / CPR, NI = 64 / RPI;
  const WinGeom g = p.g;
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* Qs = smem;
  char* Ks = Qs + CF::SEC;
  char* Vs = Ks + CF::SEC;
  float* tabL = reinterpret_cast<float*>(Vs + CF::SEC + 64);   // [head of the pair][copy A | copy B][31][32]

  const int nW = g.nWh * g.nWw, nwin = g.B * nW;
  int win, grp;
  w16_locate(nwin, win, grp);
  const int b = win / nW, wi = win - b * nW, wr = wi / g.nWw, wc = wi - wr * g.nWw;

  // ---- this wave's 64 token rows -> registers (all in flight), then the table, then LDS
  const int lr0 = lane / CPR, ch = lane - lr0 * CPR;
  const bool act = lr0 < RPI;
  const int lr = act ? lr0 : RPI - 1;
  const int sec = ch / CPS, cw = ch - sec * CPS;
  CH regs[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int t = wv * 64 + i * RPI + lr;
    const int64_t tok = win_token16(b, wr, wc, t, g);
    const char* src = reinterpret_cast<const char*>(p.qkv + tok * p.ld) + sec * (CF::C * 2) + grp * CF::PB + cw * GRAN;
    regs[i] = *reinterpret_cast<const CH*>(src);
  }
  constexpr float LOG2E = 1.4426950408889634f;
  const float rscale = 1.0f / p.scale;
  {  // table of the pair's two heads, reversed in y and x: A[dy'][u'] = T[30 - dy'][30 - u'] / scale, B[k] = A[k + 1]
    constexpr int NSRC = 2 * 961, NLD = (NSRC + 255) / 256;
    float tv[NLD];
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int j = tid + 256 * k;
      const int jj = j < NSRC ? j : NSRC - 1;
      const int hl = jj >= 961 ? 1 : 0, rel = jj - 961 * hl;
      tv[k] = p.table[rel * CF::HEADS + grp * 2 + hl];
    }
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int j = tid + 256 * k;
      if (j < NSRC) {
        const int hl = j >= 961 ? 1 : 0, rel = j - 961 * hl;
        const int ry = rel / 31, rx = rel - ry * 31;
        const int idx = (30 - ry) * CF::TROW + (30 - rx);
        const float v = tv[k] * rscale;
        float* A = tabL + hl * 2 * CF::TABF;
        A[idx] = v;
        if (idx >= 1) A[CF::TABF + idx - 1] = v;
      }
    }
  }
  if constexpr (ldt > CF::PB) {   // zero the pad bytes of every row: a padded k-step must read zeros (0 x NaN)
    constexpr int padw = (ldt - CF::PB) / 4;
    for (int idx = tid; idx < 3 * 256 * padw; idx += 256) {
      const int row = idx / padw, w = idx - row * padw;
      *reinterpret_cast<uint32_t*>(Qs + (size_t)row * ldt + CF::PB + 4 * w) = 0u;
    }
  }
  if (tid < 16) *reinterpret_cast<uint32_t*>(Vs + CF::SEC + 4 * tid) = 0u;   // guard behind the last V row
  if (act) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int t = wv * 64 + i * RPI + lr;
      chunk_to_lds<CH>(smem + sec * CF::SEC + t * ldt + cw * GRAN, regs[i]);
    }
  }
  __syncthreads();

  W16Ctx c;
  c.h = h; c.r = r;
  c.Kp = (lds_cp)(Ks + r * ldt + h * 16);
  {
    const int gq = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    c.Vp = (lds_cp)(Vs + (4 * h + q) * ldt + (16 * (gq & 1) + 4 * pp) * 2);
  }
  c.thr = g.ws - g.shift;
  c.mrow = g.shift > 0 && wr == g.nWh - 1;
  c.mcol = g.shift > 0 && wc == g.nWw - 1;
  c.masked = c.mrow || c.mcol;
  c.cbits = (uint32_t)__builtin_bit_cast(uint16_t, (__bf16)(100.0f * rscale));
  c.scale2 = p.scale * LOG2E;
#pragma unroll 1
  for (int qi = 0; qi < 2; ++qi) {
    const int qt = wv + 4 * qi;
    c.qt = qt;
    c.Qp = (lds_cp)(Qs + (qt * 32 + r) * ldt + h * 16);
    c.Orow = (lds_cp)(Qs + (qt * 32 + r) * ldt);
    c.lse = p.nlse_out ? p.nlse_out + win_token16(b, wr, wc, qt * 32 + r, g) * CF::HEADS + grp * 2 : nullptr;
    {
      const int yi = 2 * qt + (r >> 4), xi = r & 15;
      const int u0 = (15 - yi) * CF::TROW + 15 - xi + 4 * h;
      const float* tb = (u0 & 1) ? tabL + CF::TABF + (u0 - 1) : tabL + u0;
      c.tb = (const LDS_AS f32x2*)tb;
    }
    w16_head_fwd<D, 0>(c);
    w16_head_fwd<D, 1>(c);
  }
  __syncthreads();
  // O (in the Q section) -> global rows: this wave's 64 query rows (tiles wv and wv + 4), CPS chunks per row
#pragma unroll 1
  for (int idx = lane; idx < 64 * CPS; idx += 64) {
    const int row = idx / CPS, k = idx - row * CPS;
    const int t = (row < 32 ? wv * 32 : (wv + 4) * 32 - 32) + row;
    const int64_t tok = win_token16(b, wr, wc, t, g);
    char* dst = reinterpret_cast<char*>(p.out + tok * p.ldo) + grp * CF::PB + k * GRAN;
    *reinterpret_cast<CH*>(dst) = chunk_from_lds<CH>(Qs + (size_t)t * ldt + k * GRAN);
  }
}

template <int D>
constexpr size_t w16_fwd_smem() { return (size_t)3 * W16<D>::SEC + 64 + (size_t)2 * 2 * W16<D>::TABF * 4; }

template <int D>
int launch_fwd16(const W16Args& p, hipStream_t st) {
  auto kern = wattn16_fwd_kernel<D>;
  constexpr size_t smem = w16_fwd_smem<D>();
  static_assert(smem <= 80 * 1024, "two workgroups per CU");
  if (smem > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  const int64_t nwin = (int64_t)p.g.B * p.g.nWh * p.g.nWw;
  hipLaunchKernelGGL(kern, dim3((unsigned)(3 * nwin)), dim3(256), smem, st, p);
  return rdst_launch_status("wattn16_fwd");
}

// ------------------------------------------------------------------------------------------------------------------
// Backward.  One 8-wave workgroup per (window, head pair), all of Q / K / V / dO of the pair in LDS, two passes:
//   pass 1 (wave = query tile, the forward's orientation: keys in the accumulator registers): S^T -> P^T (normalised,
//     128 registers), dP^T = V.dO^T per key tile, delta = rowsum(P.dP), dS^T = P^T.(dP^T - delta) -> dQ^T += K^T.dS^T
//     and d(table) (LDS float adds into the reversed layout of the bias table); (-scale2.max - log2 l, -delta) per query
//     go to LDS for pass 2 (the exponent offset that makes exp2 return P itself, the initial accumulator of dP); dP^T is formed twice (1-2 MFMAs per tile) instead of being kept (128 more registers);
//   pass 2 (wave = key tile, queries in the accumulator registers, keys on the lanes): S, P, dP, dS recomputed tile by
//     tile from the saved row statistics, dV^T += dO^T.P and dK^T += Q^T.dS accumulate in registers over the 8 query
//     tiles (the tiles ARE the B operands; Q / dO are read transposed).
// dQ^T tiles stay in registers across pass 2 and overwrite the wave's own (then dead) Q rows, dK / dV its K / V rows;
// rows leave coalesced.
// d(table)[dy][dx] = sum of dS over the pairs with (yi - yj, xi - xj) = (dy, dx), deterministic (LDS float atomics measured
// 180 cycles per wave instruction: 370 thousand cycles per workgroup): in pass 1 a lane row is one query row (fixed yi,
// xi = lane & 15) and a register one key, so the x-diagonals are lane SHIFTS by a compile-time amount: two DPP adds per
// element (row_shr / row_shl into a low and a high 16-column accumulator per key row), the two lane halves (keys 4
// columns apart) merged by one half swap + shifted adds, and the 31-column row sums of (wave, query row, key row) go to
// a private LDS slot; after a barrier one thread per table entry adds its <= 16 slots in fixed order.
template <int D>
struct W16B {
  using CF = W16<D>;
  static constexpr int SEC = CF::SEC;
  static constexpr int OFF_TABR = 4 * SEC + 64;                       // reversed table (pass 1): [head][copy][31][32]
  static constexpr int OFF_TABN = OFF_TABR + 2 * 2 * CF::TABF * 4;    // natural table (pass 2)
  static constexpr int OFF_STAT = OFF_TABN + 2 * 2 * CF::TABF * 4;    // [head][nm - log2 l | -delta | (unused)][256]
  static constexpr int OFF_PART = OFF_STAT + 2 * 3 * 256 * 4;         // d(table) row sums of one head: [yi 16][yj 16][32]
  static constexpr size_t SMEM = (size_t)OFF_PART + 16 * 16 * 32 * 4;
};

struct W16BCtx {
  W16Ctx sc;                     // pass 1 score context (Qp = own query rows, Kp = key rows, tb = reversed table)
  lds_cp dOp, Vrow, Ktr;         // pass 1: dO^T pack base, V rows as A operand, K^T transposed reads
  LDS_AS float* part;            // pass 1: the lane's slot row [yi][.][lane & 15] of the d(table) row sums
  LDS_AS float* stat;            // statistics base
  lds_cp QA, dOA, Qtr, dOtr, Kown, Vown, Kst, Vst;   // pass 2
  const LDS_AS f32x2* tb2;       // pass 2: lane base into the natural table copy of its parity
  int kt;
  float scale;
};

__device__ __forceinline__ Pack16 tr_pack(lds_cp p, int ldt) { return lds_tr_pack(p, p + 8 * ldt); }

template <int D, int HL>
__device__ __forceinline__ void w16_bwd_p1(const W16BCtx& c, f32x16& dq) {
  using CF = W16<D>;
  constexpr int ldt = CF::LDT;
  constexpr int c_lo = HL * D, c_hi = c_lo + D;
  constexpr int t_lo = c_lo / 16, t_hi = (c_hi - 1) / 16, NT = t_hi - t_lo + 1;
  constexpr int RL = c_lo & ~3;
  const int h = c.sc.h;
  f32x16 X[8];
  w16_scores<D, HL>(X, c.sc);
  float m = X[0][0];
#pragma unroll
  for (int kt = 0; kt < 8; ++kt)
#pragma unroll
    for (int v = 0; v < 16; ++v) m = __builtin_fmaxf(m, X[kt][v]);
  m = half_swap_max(m);
  const float nm = -c.sc.scale2 * m;
  float l0 = 0.f, l1 = 0.f;
#pragma unroll
  for (int kt = 0; kt < 8; ++kt)
#pragma unroll
    for (int v = 0; v < 16; v += 2) {
      const float e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(X[kt][v], c.sc.scale2, nm));
      const float e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(X[kt][v + 1], c.sc.scale2, nm));
      l0 += e0;
      l1 += e1;
      X[kt][v] = e0;
      X[kt][v + 1] = e1;
    }
  const float inv = __builtin_amdgcn_rcpf(half_swap_sum(l0 + l1));
#pragma unroll
  for (int kt = 0; kt < 8; ++kt)
#pragma unroll
    for (int v = 0; v < 16; ++v) X[kt][v] *= inv;

  Pack16 dob[NT];   // dO^T of the lane's query, the head's channels only
#pragma unroll
  for (int t = t_lo; t <= t_hi; ++t) {
    dob[t - t_lo] = lds_pack(c.dOp + t * 32);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t mA = qmask_bits(t * 16 + 2 * e, c_lo, c_hi), mB = qmask_bits(t * 16 + 8 + 2 * e, c_lo, c_hi);
      dob[t - t_lo].w[e] &= h ? mB : mA;
    }
  }
  auto dp_tile = [&](int kt, f32x16& dp, float init) {
#pragma unroll
    for (int v = 0; v < 16; ++v) dp[v] = init;
#pragma unroll
    for (int t = t_lo; t <= t_hi; ++t) Mma<bf16>::mma(dp, lds_pack(c.Vrow + kt * 32 * ldt + t * 32), dob[t - t_lo]);
  };
  float d0 = 0.f, d1 = 0.f;
#pragma unroll
  for (int kt = 0; kt < 8; ++kt) {
    f32x16 dp;
    dp_tile(kt, dp, 0.f);
#pragma unroll
    for (int v = 0; v < 16; v += 2) {
      d0 = __builtin_fmaf(X[kt][v], dp[v], d0);
      d1 = __builtin_fmaf(X[kt][v + 1], dp[v + 1], d1);
    }
  }
  const float delta = half_swap_sum(d0 + d1);
  if (h == 0) {
    LDS_AS float* st = c.stat + HL * 3 * 256 + c.sc.qt * 32 + c.sc.r;
    st[0] = nm + __builtin_amdgcn_logf(inv);   // pass 2: P = exp2(scale2 . S + this) — the 1 / l folded into the exponent
    st[256] = -delta;                          // pass 2: the initial accumulator of dP, so that dS = P . acc
  }
#pragma unroll
  for (int v = 0; v < 16; ++v) dq[v] = 0.f;
#pragma unroll
  for (int kt = 0; kt < 8; ++kt) {
    f32x16 dp;
    dp_tile(kt, dp, -delta);   // dP - delta: -delta is the initial accumulator
    // register v: key (yj = 2 kt + (v >> 3), xj = XL + 4 h) with XL = 8 ((v >> 2) & 1) + (v & 3); column c' = xi - XL + 15
    // of the key row's sums goes to lane c' of `lo` (c' < 16) / lane c' - 16 of `hi`; true column = c' - 4 h
    float lo[2] = {0.f, 0.f}, hi[2] = {0.f, 0.f};
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const float ds = X[kt][v] * dp[v];
      dp[v] = ds;
      const int yl = v >> 3;
      switch (8 * ((v >> 2) & 1) + (v & 3)) {
#define RDST_W16_DIAG(XL) case XL: lo[yl] += dpp_row_shr<15 - XL>(ds); hi[yl] += dpp_row_shl<XL + 1>(ds); break;
        RDST_W16_DIAG(0) RDST_W16_DIAG(1) RDST_W16_DIAG(2) RDST_W16_DIAG(3)
        RDST_W16_DIAG(8) RDST_W16_DIAG(9) RDST_W16_DIAG(10) RDST_W16_DIAG(11)
#undef RDST_W16_DIAG
        default: break;
      }
    }
#pragma unroll
    for (int yl = 0; yl < 2; ++yl) {   // lane half 1 holds the same row sums 4 columns further right: merge into half 0
      const float l1 = other_half(lo[yl], h), h1 = other_half(hi[yl], h);
      const float nl = lo[yl] + dpp_row_shl<4>(l1) + dpp_row_shr<12>(h1);
      const float nh = hi[yl] + dpp_row_shl<4>(h1);
      if (h == 0) {
        c.part[(2 * kt + yl) * 32] = nl;
        c.part[(2 * kt + yl) * 32 + 16] = nh;
      }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      Pack16 pb;
#pragma unroll
      for (int e = 0; e < 4; ++e) pb.w[e] = pack_bf16x2(dp[8 * s + 2 * e], dp[8 * s + 2 * e + 1]);
      Mma<bf16>::mma(dq, tr_pack(c.Ktr + RL * 2 + (kt * 32 + 16 * s) * ldt, ldt), pb);   // rows = channels (K^T), cols = queries
    }
  }
}

// Pass 1 of the backward WITH the forward's row statistics (round 5): nothing about a query's row has to be known before its
// key tiles are visited any more — P = exp2(scale2 . S + nlse) tile by tile, dP - delta with -delta = -rowsum(dO o O) as the
// initial accumulator — so the 8 key tiles are streamed through 16 + 16 registers instead of living in 128: no row maximum,
// no row sum, no normalisation, no delta pass, dP formed ONCE: 5.5 vector instructions per logit instead of 8.5 and half
// the matrix instructions.  (nlse, -delta) of the head are read from `stat`, where the prologue put them for pass 2.
template <int D, int HL, typename PT = float, int HPG = 2>
__device__ __forceinline__ void w16_bwd_p1s(const W16BCtx& c, f32x16& dq) {
  using CF = W16<D, HPG>;
  constexpr int ldt = CF::LDT;
  // pair kernels: HL = the head of the pair (its channels follow the other head's, its tables / statistics are the second set);
  // one head per workgroup: one set, and HL = the first channel of the head inside its staged window (1 for an odd head of dim 15)
  constexpr int HS = HPG == 1 ? 0 : HL;
  constexpr int c_lo = HPG == 1 ? HL : HL * D, c_hi = c_lo + D;
  constexpr int t_lo = c_lo / 16, t_hi = (c_hi - 1) / 16, NT = t_hi - t_lo + 1;
  constexpr int RL = c_lo & ~3;
  const int h = c.sc.h;
  const LDS_AS float* st = c.stat + HS * 3 * 256 + c.sc.qt * 32 + c.sc.r;
  const float nl = st[0], nd = st[256];
  Pack16 qb[NT], dob[NT];   // Q and dO^T of the lane's query, the head's channels only
#pragma unroll
  for (int t = t_lo; t <= t_hi; ++t) {
    qb[t - t_lo] = lds_pack(c.sc.Qp + t * 32);
    dob[t - t_lo] = lds_pack(c.dOp + t * 32);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t mA = qmask_bits(t * 16 + 2 * e, c_lo, c_hi), mB = qmask_bits(t * 16 + 8 + 2 * e, c_lo, c_hi);
      qb[t - t_lo].w[e] &= h ? mB : mA;
      dob[t - t_lo].w[e] &= h ? mB : mA;
    }
  }
  Pack16 mQ;
  const int xi = c.sc.r & 15;
  const int rx = (c.sc.mcol && xi >= c.sc.thr) ? 1 : 0;
  if (c.sc.masked) onehot4(2 * ((c.sc.mrow && 2 * c.sc.qt + (c.sc.r >> 4) >= c.sc.thr) ? 1 : 0) + rx, c.sc.cbits, h, mQ);
  const LDS_AS f32x2* tbh = c.sc.tb + HS * (2 * CF::TABF / 2);
#pragma unroll
  for (int v = 0; v < 16; ++v) dq[v] = 0.f;
#pragma unroll
  for (int kt = 0; kt < 8; ++kt) {
    f32x16 X, dp;
#pragma unroll
    for (int v = 0; v < 16; v += 2) {
      const f32x2 b2 = lds_read_f32x2(tbh + ((2 * kt + (v >> 3)) * CF::TROW + 8 * ((v >> 2) & 1) + (v & 3)) / 2);
      X[v] = b2.x;
      X[v + 1] = b2.y;
    }
#pragma unroll
    for (int v = 0; v < 16; ++v) dp[v] = nd;
#pragma unroll
    for (int t = t_lo; t <= t_hi; ++t) {
      Mma<bf16>::mma(X, lds_pack(c.sc.Kp + kt * 32 * ldt + t * 32), qb[t - t_lo]);
      Mma<bf16>::mma(dp, lds_pack(c.Vrow + kt * 32 * ldt + t * 32), dob[t - t_lo]);
    }
    if (c.sc.masked) {
      Pack16 mK;
      onehot4(2 * ((c.sc.mrow && 2 * kt + (c.sc.r >> 4) >= c.sc.thr) ? 1 : 0) + rx, 0x3f80u, h, mK);
      Mma<bf16>::mma(X, mK, mQ);
    }
    float lo[2] = {0.f, 0.f}, hi[2] = {0.f, 0.f};
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const float ds = __builtin_amdgcn_exp2f(__builtin_fmaf(X[v], c.sc.scale2, nl)) * dp[v];
      dp[v] = ds;
      const int yl = v >> 3;
      switch (8 * ((v >> 2) & 1) + (v & 3)) {
#define RDST_W16_DIAG(XL) case XL: lo[yl] += dpp_row_shr<15 - XL>(ds); hi[yl] += dpp_row_shl<XL + 1>(ds); break;
        RDST_W16_DIAG(0) RDST_W16_DIAG(1) RDST_W16_DIAG(2) RDST_W16_DIAG(3)
        RDST_W16_DIAG(8) RDST_W16_DIAG(9) RDST_W16_DIAG(10) RDST_W16_DIAG(11)
#undef RDST_W16_DIAG
        default: break;
      }
    }
#pragma unroll
    for (int yl = 0; yl < 2; ++yl) {   // lane half 1 holds the same row sums 4 columns further right: merge into half 0
      const float l1 = other_half(lo[yl], h), h1 = other_half(hi[yl], h);
      const float nlo = lo[yl] + dpp_row_shl<4>(l1) + dpp_row_shr<12>(h1);
      const float nhi = hi[yl] + dpp_row_shl<4>(h1);
      if (h == 0) {
        if constexpr (sizeof(PT) == 4) {
          c.part[(2 * kt + yl) * 32] = nlo;
          c.part[(2 * kt + yl) * 32 + 16] = nhi;
        } else {   // bf16 row sums in the 16-wave kernel: both heads' slots have to fit
          LDS_AS uint16_t* pt = (LDS_AS uint16_t*)c.part;
          pt[(2 * kt + yl) * 32] = __builtin_bit_cast(uint16_t, (__bf16)nlo);
          pt[(2 * kt + yl) * 32 + 16] = __builtin_bit_cast(uint16_t, (__bf16)nhi);
        }
      }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      Pack16 pb;
#pragma unroll
      for (int e = 0; e < 4; ++e) pb.w[e] = pack_bf16x2(dp[8 * s + 2 * e], dp[8 * s + 2 * e + 1]);
      Mma<bf16>::mma(dq, tr_pack(c.Ktr + RL * 2 + (kt * 32 + 16 * s) * ldt, ldt), pb);   // rows = channels (K^T), cols = queries
    }
  }
}

template <int D, int HL, int HPG = 2>
__device__ __forceinline__ void w16_bwd_p2(const W16BCtx& c) {
  using CF = W16<D, HPG>;
  constexpr int ldt = CF::LDT;
  constexpr int HS = HPG == 1 ? 0 : HL;   // (see w16_bwd_p1s)
  constexpr int c_lo = HPG == 1 ? HL : HL * D, c_hi = c_lo + D;
  constexpr int t_lo = c_lo / 16, t_hi = (c_hi - 1) / 16, NT = t_hi - t_lo + 1;
  constexpr int RL = c_lo & ~3;
  const int h = c.sc.h, r = c.sc.r;
  Pack16 kb[NT], vb[NT];   // K^T / V^T of the lane's key, the head's channels only
#pragma unroll
  for (int t = t_lo; t <= t_hi; ++t) {
    kb[t - t_lo] = lds_pack(c.Kown + t * 32);
    vb[t - t_lo] = lds_pack(c.Vown + t * 32);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t mA = qmask_bits(t * 16 + 2 * e, c_lo, c_hi), mB = qmask_bits(t * 16 + 8 + 2 * e, c_lo, c_hi);
      kb[t - t_lo].w[e] &= h ? mB : mA;
      vb[t - t_lo].w[e] &= h ? mB : mA;
    }
  }
  Pack16 mKc;   // one-hot region of the lane's key, times 100 / scale
  const int xk = r & 15;
  const int rxk = (c.sc.mcol && xk >= c.sc.thr) ? 1 : 0;
  if (c.sc.masked) onehot4(2 * ((c.sc.mrow && 2 * c.kt + (r >> 4) >= c.sc.thr) ? 1 : 0) + rxk, c.sc.cbits, h, mKc);
  f32x16 dk, dv;
#pragma unroll
  for (int v = 0; v < 16; ++v) { dk[v] = 0.f; dv[v] = 0.f; }
  const LDS_AS f32x2* tbh = c.tb2 + HS * CF::TABF;
  const LDS_AS char* stb = (const LDS_AS char*)(c.stat + HS * 3 * 256) + h * 16;
  typedef float f32x4v __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int qt = 0; qt < 8; ++qt) {
    // S tile: rows = queries of tile qt (yi = 2 qt + (v >> 3), xi = 8 ((v >> 2) & 1) + 4 h + (v & 3)), cols = the lane's key
    f32x16 X, dp;
#pragma unroll
    for (int v = 0; v < 16; v += 2) {
      const f32x2 b2 = lds_read_f32x2(tbh + ((2 * qt + (v >> 3)) * CF::TROW + 8 * ((v >> 2) & 1) + (v & 3)) / 2);
      X[v] = b2.x;
      X[v + 1] = b2.y;
    }
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {   // dP starts at -delta of its query row
      const f32x4v nd = *reinterpret_cast<const LDS_AS f32x4v*>(stb + (256 + qt * 32 + 8 * g4) * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) dp[4 * g4 + e] = nd[e];
    }
#pragma unroll
    for (int t = t_lo; t <= t_hi; ++t) {
      Mma<bf16>::mma(X, lds_pack(c.QA + qt * 32 * ldt + t * 32), kb[t - t_lo]);
      Mma<bf16>::mma(dp, lds_pack(c.dOA + qt * 32 * ldt + t * 32), vb[t - t_lo]);
    }
    if (c.sc.masked) {
      Pack16 mQr;   // rows = queries of tile qt: row r is query (2 qt + (r >> 4), r & 15)
      onehot4(2 * ((c.sc.mrow && 2 * qt + (r >> 4) >= c.sc.thr) ? 1 : 0) + rxk, 0x3f80u, h, mQr);
      Mma<bf16>::mma(X, mQr, mKc);
    }
    f32x16 P, S;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const f32x4v nm = *reinterpret_cast<const LDS_AS f32x4v*>(stb + (qt * 32 + 8 * g4) * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int v = 4 * g4 + e;
        const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(X[v], c.sc.scale2, nm[e]));
        P[v] = pv;
        S[v] = pv * dp[v];
      }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      Pack16 pp, ps;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        pp.w[e] = pack_bf16x2(P[8 * s + 2 * e], P[8 * s + 2 * e + 1]);
        ps.w[e] = pack_bf16x2(S[8 * s + 2 * e], S[8 * s + 2 * e + 1]);
      }
      Mma<bf16>::mma(dv, tr_pack(c.dOtr + RL * 2 + (qt * 32 + 16 * s) * ldt, ldt), pp);   // dV^T += dO^T . P
      Mma<bf16>::mma(dk, tr_pack(c.Qtr + RL * 2 + (qt * 32 + 16 * s) * ldt, ldt), ps);    // dK^T += Q^T . dS
    }
  }
  store_tile_rows<RL, c_lo, c_hi>(c.Kst, dk, c.scale, h);
  store_tile_rows<RL, c_lo, c_hi>(c.Vst, dv, 1.0f, h);
}

template <int D, bool V2>
__global__ void __launch_bounds__(512, 1) wattn16_bwd_kernel(const W16Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using CF = W16<D>;
  using BF = W16B<D>;
  using CH = typename Chunk<CF::GRAN>::type;
  constexpr int ldt = CF::LDT, GRAN = CF::GRAN, CPS = CF::CPS, CPR = 3 * CPS, RPI = 64 / CPR, NI = 32 / RPI;
  constexpr int RPD = 64 / CPS, ND = (32 + RPD - 1) / RPD;   // dO: rows per instruction, instructions
  const WinGeom g = p.g;
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* Qs = smem;
  char* Ks = Qs + CF::SEC;
  char* Vs = Ks + CF::SEC;
  char* dOs = Vs + CF::SEC;
  float* tabR = reinterpret_cast<float*>(smem + BF::OFF_TABR);
  float* tabN = reinterpret_cast<float*>(smem + BF::OFF_TABN);
  float* stat = reinterpret_cast<float*>(smem + BF::OFF_STAT);
  float* part = reinterpret_cast<float*>(smem + BF::OFF_PART);

  const int nW = g.nWh * g.nWw, nwin = g.B * nW;
  int win, grp;
  w16_locate(nwin, win, grp);
  const int b = win / nW, wi = win - b * nW, wr = wi / g.nWw, wc = wi - wr * g.nWw;

  // ---- this wave's 32 token rows of qkv and dO -> registers
  const int lr0 = lane / CPR, ch = lane - lr0 * CPR;
  const bool act = lr0 < RPI;
  const int lr = act ? lr0 : RPI - 1;
  const int sec = ch / CPS, cw = ch - sec * CPS;
  CH regs[NI], dreg[ND];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int t = wv * 32 + i * RPI + lr;
    const int64_t tok = win_token16(b, wr, wc, t, g);
    const char* src = reinterpret_cast<const char*>(p.qkv + tok * p.ld) + sec * (CF::C * 2) + grp * CF::PB + cw * GRAN;
    regs[i] = *reinterpret_cast<const CH*>(src);
  }
  const int dr0 = lane / CPS, dc = lane - dr0 * CPS;
  const bool dact0 = dr0 < RPD;
#pragma unroll
  for (int i = 0; i < ND; ++i) {
    int rr = i * RPD + (dact0 ? dr0 : RPD - 1);
    rr = rr < 32 ? rr : 31;
    const int64_t tok = win_token16(b, wr, wc, wv * 32 + rr, g);
    const char* src = reinterpret_cast<const char*>(p.dout + tok * p.ldd) + grp * CF::PB + dc * GRAN;
    dreg[i] = *reinterpret_cast<const CH*>(src);
  }
  constexpr float LOG2E = 1.4426950408889634f;
  const float rscale = 1.0f / p.scale;
  {  // the pair's bias tables / scale: reversed (pass 1) and natural (pass 2), each with a copy shifted by one float
    constexpr int NSRC = 2 * 961, NLD = (NSRC + 511) / 512;
    float tv[NLD];
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int j = tid + 512 * k;
      const int jj = j < NSRC ? j : NSRC - 1;
      const int hl = jj >= 961 ? 1 : 0, rel = jj - 961 * hl;
      tv[k] = p.table[rel * CF::HEADS + grp * 2 + hl];
    }
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int j = tid + 512 * k;
      if (j < NSRC) {
        const int hl = j >= 961 ? 1 : 0, rel = j - 961 * hl;
        const int ry = rel / 31, rx = rel - ry * 31;
        const float v = tv[k] * rscale;
        const int ir = (30 - ry) * CF::TROW + (30 - rx), in = ry * CF::TROW + rx;
        float* A = tabR + hl * 2 * CF::TABF;
        A[ir] = v;
        if (ir >= 1) A[CF::TABF + ir - 1] = v;
        float* N = tabN + hl * 2 * CF::TABF;
        N[in] = v;
        if (in >= 1) N[CF::TABF + in - 1] = v;
      }
    }
  }
  if (tid < 16) *reinterpret_cast<uint32_t*>(dOs + CF::SEC + 4 * tid) = 0u;   // guard behind the last dO row
  if constexpr (ldt > CF::PB) {   // zero the pad bytes of every staged row
    constexpr int padw = (ldt - CF::PB) / 4;
    for (int idx = tid; idx < 4 * 256 * padw; idx += 512) {
      const int row = idx / padw, w = idx - row * padw;
      *reinterpret_cast<uint32_t*>(Qs + (size_t)row * ldt + CF::PB + 4 * w) = 0u;
    }
  }
  if (act) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int t = wv * 32 + i * RPI + lr;
      chunk_to_lds<CH>(smem + sec * CF::SEC + t * ldt + cw * GRAN, regs[i]);
    }
  }
#pragma unroll
  for (int i = 0; i < ND; ++i) {
    const int rr = i * RPD + dr0;
    if (dact0 && rr < 32) chunk_to_lds<CH>(dOs + (wv * 32 + rr) * ldt + dc * GRAN, dreg[i]);
  }
  __syncthreads();

  if constexpr (V2) {
    // (nlse, -delta) of this wave's 32 queries, lane half h = head h of the pair: delta = sum over the head's channels of
    // dO . O (= sum_j P_ij dP_ij: what pass 1 used to accumulate over all 256 keys), O from the forward's output rows
    const int q = wv * 32 + r;
    const int64_t tok = win_token16(b, wr, wc, q, g);
    const bf16* orow = p.o + tok * p.ldo2 + grp * CF::GC + h * D;
    const char* drow = dOs + q * ldt + h * D * 2;
    float dl = 0.f;
    if constexpr (D % 2 == 0) {
      uint32_t ov[D / 2];
#pragma unroll
      for (int i = 0; i < D / 2; ++i) ov[i] = reinterpret_cast<const uint32_t*>(orow)[i];
#pragma unroll
      for (int i = 0; i < D / 2; ++i) {
        const uint32_t dv = reinterpret_cast<const uint32_t*>(drow)[i];
        dl = fmaf(bf16lo(dv), bf16lo(ov[i]), dl);
        dl = fmaf(bf16hi(dv), bf16hi(ov[i]), dl);
      }
    } else {
      uint16_t ov[D];
#pragma unroll
      for (int i = 0; i < D; ++i) ov[i] = reinterpret_cast<const uint16_t*>(orow)[i];
#pragma unroll
      for (int i = 0; i < D; ++i) {
        const uint16_t dv = reinterpret_cast<const uint16_t*>(drow)[i];
        dl = fmaf(__uint_as_float((uint32_t)dv << 16), __uint_as_float((uint32_t)ov[i] << 16), dl);
      }
    }
    stat[h * 3 * 256 + q] = p.nlse[tok * CF::HEADS + grp * 2 + h];
    stat[h * 3 * 256 + 256 + q] = -dl;
  }
  W16BCtx c;
  c.sc.h = h; c.sc.r = r;
  c.sc.thr = g.ws - g.shift;
  c.sc.mrow = g.shift > 0 && wr == g.nWh - 1;
  c.sc.mcol = g.shift > 0 && wc == g.nWw - 1;
  c.sc.masked = c.sc.mrow || c.sc.mcol;
  c.sc.cbits = (uint32_t)__builtin_bit_cast(uint16_t, (__bf16)(100.0f * rscale));
  c.sc.scale2 = p.scale * LOG2E;
  c.scale = p.scale;
  c.stat = (LDS_AS float*)stat;
  const int trofs = (4 * h + ((lane & 15) >> 2)) * ldt + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;   // transposed-read lane offset
  const int y0 = 2 * wv + (r >> 4), x0 = r & 15;   // the lane's query (pass 1) / key (pass 2)
  const int u0 = (15 - y0) * CF::TROW + 15 - x0 + 4 * h;
  {  // pass 1: wave = query tile wv
    c.sc.qt = wv;
    c.sc.Qp = (lds_cp)(Qs + (wv * 32 + r) * ldt + h * 16);
    c.sc.Kp = (lds_cp)(Ks + r * ldt + h * 16);
    c.sc.tb = (const LDS_AS f32x2*)((u0 & 1) ? tabR + CF::TABF + (u0 - 1) : tabR + u0);
    c.dOp = (lds_cp)(dOs + (wv * 32 + r) * ldt + h * 16);
    c.Vrow = (lds_cp)(Vs + r * ldt + h * 16);
    c.Ktr = (lds_cp)(Ks + trofs);
    c.part = (LDS_AS float*)(part + (2 * wv + (r >> 4)) * 16 * 32 + (r & 15));
  }
  f32x16 dq0, dq1;
  float* slab_row = p.slab + ((int64_t)win * CF::HEADS + grp * 2) * 961;
#ifndef W16_ABL
#define W16_ABL 0   // compile-time ablations (tools/abl_build.sh): 1 no pass 1, 2 no pass 2, 4 no d(table) sums
#endif
#if W16_ABL & 1
#pragma unroll
  for (int v = 0; v < 16; ++v) { dq0[v] = 0.f; dq1[v] = 0.f; }
#else
  if constexpr (V2) w16_bwd_p1s<D, 0>(c, dq0);
  else w16_bwd_p1<D, 0>(c, dq0);
#endif
  __syncthreads();
  if (!(W16_ABL & 4)) w16_dtable_out(part, slab_row, tid);
  __syncthreads();
#if !(W16_ABL & 1)
  if constexpr (V2) w16_bwd_p1s<D, 1>(c, dq1);
  else w16_bwd_p1<D, 1>(c, dq1);
#endif
  __syncthreads();
  if (!(W16_ABL & 4)) w16_dtable_out(part, slab_row + 961, tid);
  {  // pass 2: wave = key tile wv
    c.kt = wv;
    c.QA = (lds_cp)(Qs + r * ldt + h * 16);
    c.dOA = (lds_cp)(dOs + r * ldt + h * 16);
    c.Qtr = (lds_cp)(Qs + trofs);
    c.dOtr = (lds_cp)(dOs + trofs);
    c.Kown = (lds_cp)(Ks + (wv * 32 + r) * ldt + h * 16);
    c.Vown = (lds_cp)(Vs + (wv * 32 + r) * ldt + h * 16);
    c.Kst = (lds_cp)(Ks + (wv * 32 + r) * ldt);
    c.Vst = (lds_cp)(Vs + (wv * 32 + r) * ldt);
    c.tb2 = (const LDS_AS f32x2*)((u0 & 1) ? tabN + CF::TABF + (u0 - 1) : tabN + u0);
#if !(W16_ABL & 2)
    w16_bwd_p2<D, 0>(c);
    w16_bwd_p2<D, 1>(c);
#endif
  }
  __syncthreads();   // every wave is done with Q as an operand: the wave's own query rows take dQ
  {
    const lds_cp qrow = (lds_cp)(Qs + (wv * 32 + r) * ldt);
    store_tile_rows<0, 0, D>(qrow, dq0, p.scale, h);
    store_tile_rows<(D & ~3), D, 2 * D>(qrow, dq1, p.scale, h);
  }
  __syncthreads();
  // dQ | dK | dV (in place of Q / K / V) -> global rows of this wave's 32 tokens
#pragma unroll 1
  for (int idx = lane; idx < 32 * CPR; idx += 64) {
    const int row = idx / CPR, k3 = idx - row * CPR;
    const int s3 = k3 / CPS, k = k3 - s3 * CPS;
    const int t = wv * 32 + row;
    const int64_t tok = win_token16(b, wr, wc, t, g);
    char* dst = reinterpret_cast<char*>(p.dqkv + tok * p.ldq) + s3 * (CF::C * 2) + grp * CF::PB + k * GRAN;
    const char* src = smem + s3 * CF::SEC + (size_t)t * ldt + k * GRAN;
    *reinterpret_cast<CH*>(dst) = chunk_from_lds<CH>(src);
  }
}

// ---- round 5: the backward with the forward's statistics as ONE 16-wave workgroup per (window, head pair) -----------------
// With the streaming first pass (w16_bwd_p1s) a wave needs 125-141 registers instead of 218-232, so both heads of the pair run
// side by side: wave = (tile, head) in both passes, 4 waves per SIMD where the 8-wave kernel had 2 (its vector ALUs were 31 %
// busy).  The d(table) row sums of the two heads sit in LDS as bf16 (2 x 16 KB; fp32 would not fit at D = 15 / 20) and are
// added in fp32 in a fixed order: 2^-9 per slot, 16 slots per entry.
template <int D>
struct W16B3 {
  using CF = W16<D>;
  static constexpr int SEC = CF::SEC;
  static constexpr int OFF_TABR = 4 * SEC + 64;
  static constexpr int OFF_TABN = OFF_TABR + 2 * 2 * CF::TABF * 4;
  static constexpr int OFF_STAT = OFF_TABN + 2 * 2 * CF::TABF * 4;    // [head][nlse | -delta | (unused)][256]
  static constexpr int OFF_PART = OFF_STAT + 2 * 3 * 256 * 4;         // [head][yi 16][yj 16][32] bf16
  static constexpr size_t SMEM = (size_t)OFF_PART + 2 * 16 * 16 * 32 * 2;
};

template <int D>
__global__ void __launch_bounds__(1024, 4) wattn16_bwd3_kernel(const W16Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using CF = W16<D>;
  using BF = W16B3<D>;
  using CH = typename Chunk<CF::GRAN>::type;
  constexpr int ldt = CF::LDT, GRAN = CF::GRAN, CPS = CF::CPS, CPR = 3 * CPS, RPI = 64 / CPR, NI = 16 / RPI;
  constexpr int RPD = 64 / CPS, ND = (16 + RPD - 1) / RPD;
  static_assert(16 % RPI == 0, "a wave stages 16 token rows");
  const WinGeom g = p.g;
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tl = wv & 7, hl = wv >> 3;          // this wave's tile (queries in pass 1, keys in pass 2) and head of the pair
  char* Qs = smem;
  char* Ks = Qs + CF::SEC;
  char* Vs = Ks + CF::SEC;
  char* dOs = Vs + CF::SEC;
  float* tabR = reinterpret_cast<float*>(smem + BF::OFF_TABR);
  float* tabN = reinterpret_cast<float*>(smem + BF::OFF_TABN);
  float* stat = reinterpret_cast<float*>(smem + BF::OFF_STAT);
  bf16* part = reinterpret_cast<bf16*>(smem + BF::OFF_PART);

  const int nW = g.nWh * g.nWw, nwin = g.B * nW;
  int win, grp;
  w16_locate(nwin, win, grp);
  const int b = win / nW, wi = win - b * nW, wr = wi / g.nWw, wc = wi - wr * g.nWw;

  // ---- this wave's 16 token rows of qkv and dO -> registers
  const int lr0 = lane / CPR, ch = lane - lr0 * CPR;
  const bool act = lr0 < RPI;
  const int lr = act ? lr0 : RPI - 1;
  const int sec = ch / CPS, cw = ch - sec * CPS;
  CH regs[NI], dreg[ND];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int t = wv * 16 + i * RPI + lr;
    const int64_t tok = win_token16(b, wr, wc, t, g);
    const char* src = reinterpret_cast<const char*>(p.qkv + tok * p.ld) + sec * (CF::C * 2) + grp * CF::PB + cw * GRAN;
    regs[i] = *reinterpret_cast<const CH*>(src);
  }
  const int dr0 = lane / CPS, dc = lane - dr0 * CPS;
  const bool dact0 = dr0 < RPD;
#pragma unroll
  for (int i = 0; i < ND; ++i) {
    int rr = i * RPD + (dact0 ? dr0 : RPD - 1);
    rr = rr < 16 ? rr : 15;
    const int64_t tok = win_token16(b, wr, wc, wv * 16 + rr, g);
    const char* src = reinterpret_cast<const char*>(p.dout + tok * p.ldd) + grp * CF::PB + dc * GRAN;
    dreg[i] = *reinterpret_cast<const CH*>(src);
  }
  // (nlse, -delta) of query tile tl for head hl: lane r = query, both lane halves compute the same
  const int qown = tl * 32 + r;
  const int64_t tokq = win_token16(b, wr, wc, qown, g);
  float dl = 0.f;
  const float nl_own = p.nlse[tokq * CF::HEADS + grp * 2 + hl];
  {
    const bf16* orow = p.o + tokq * p.ldo2 + grp * CF::GC + hl * D;
    const bf16* drow = p.dout + tokq * p.ldd + grp * CF::GC + hl * D;
    if constexpr (D % 2 == 0) {
      uint32_t ov[D / 2], dv[D / 2];
#pragma unroll
      for (int i = 0; i < D / 2; ++i) { ov[i] = reinterpret_cast<const uint32_t*>(orow)[i]; dv[i] = reinterpret_cast<const uint32_t*>(drow)[i]; }
#pragma unroll
      for (int i = 0; i < D / 2; ++i) {
        dl = fmaf(bf16lo(dv[i]), bf16lo(ov[i]), dl);
        dl = fmaf(bf16hi(dv[i]), bf16hi(ov[i]), dl);
      }
    } else {
      uint16_t ov[D], dv[D];
#pragma unroll
      for (int i = 0; i < D; ++i) { ov[i] = reinterpret_cast<const uint16_t*>(orow)[i]; dv[i] = reinterpret_cast<const uint16_t*>(drow)[i]; }
#pragma unroll
      for (int i = 0; i < D; ++i) dl = fmaf(__uint_as_float((uint32_t)dv[i] << 16), __uint_as_float((uint32_t)ov[i] << 16), dl);
    }
  }
  constexpr float LOG2E = 1.4426950408889634f;
  const float rscale = 1.0f / p.scale;
  {  // the pair's bias tables / scale: reversed (pass 1) and natural (pass 2), each with a copy shifted by one float
    constexpr int NSRC = 2 * 961, NLD = (NSRC + 1023) / 1024;
    float tv[NLD];
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int j = tid + 1024 * k;
      const int jj = j < NSRC ? j : NSRC - 1;
      const int h2 = jj >= 961 ? 1 : 0, rel = jj - 961 * h2;
      tv[k] = p.table[rel * CF::HEADS + grp * 2 + h2];
    }
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int j = tid + 1024 * k;
      if (j < NSRC) {
        const int h2 = j >= 961 ? 1 : 0, rel = j - 961 * h2;
        const int ry = rel / 31, rx = rel - ry * 31;
        const float v = tv[k] * rscale;
        const int ir = (30 - ry) * CF::TROW + (30 - rx), in = ry * CF::TROW + rx;
        float* A = tabR + h2 * 2 * CF::TABF;
        A[ir] = v;
        if (ir >= 1) A[CF::TABF + ir - 1] = v;
        float* N = tabN + h2 * 2 * CF::TABF;
        N[in] = v;
        if (in >= 1) N[CF::TABF + in - 1] = v;
      }
    }
  }
  if (tid < 16) *reinterpret_cast<uint32_t*>(dOs + CF::SEC + 4 * tid) = 0u;   // guard behind the last dO row
  if constexpr (ldt > CF::PB) {   // zero the pad bytes of every staged row
    constexpr int padw = (ldt - CF::PB) / 4;
    for (int idx = tid; idx < 4 * 256 * padw; idx += 1024) {
      const int row = idx / padw, w = idx - row * padw;
      *reinterpret_cast<uint32_t*>(Qs + (size_t)row * ldt + CF::PB + 4 * w) = 0u;
    }
  }
  if (act) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int t = wv * 16 + i * RPI + lr;
      chunk_to_lds<CH>(smem + sec * CF::SEC + t * ldt + cw * GRAN, regs[i]);
    }
  }
#pragma unroll
  for (int i = 0; i < ND; ++i) {
    const int rr = i * RPD + dr0;
    if (dact0 && rr < 16) chunk_to_lds<CH>(dOs + (wv * 16 + rr) * ldt + dc * GRAN, dreg[i]);
  }
  if (h == 0) {
    stat[hl * 3 * 256 + qown] = nl_own;
    stat[hl * 3 * 256 + 256 + qown] = -dl;
  }
  __syncthreads();

  W16BCtx c;
  c.sc.h = h; c.sc.r = r;
  c.sc.thr = g.ws - g.shift;
  c.sc.mrow = g.shift > 0 && wr == g.nWh - 1;
  c.sc.mcol = g.shift > 0 && wc == g.nWw - 1;
  c.sc.masked = c.sc.mrow || c.sc.mcol;
  c.sc.cbits = (uint32_t)__builtin_bit_cast(uint16_t, (__bf16)(100.0f * rscale));
  c.sc.scale2 = p.scale * LOG2E;
  c.scale = p.scale;
  c.stat = (LDS_AS float*)stat;
  const int trofs = (4 * h + ((lane & 15) >> 2)) * ldt + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
  const int y0 = 2 * tl + (r >> 4), x0 = r & 15;
  const int u0 = (15 - y0) * CF::TROW + 15 - x0 + 4 * h;
  {  // pass 1: wave = (query tile tl, head hl)
    c.sc.qt = tl;
    c.sc.Qp = (lds_cp)(Qs + (tl * 32 + r) * ldt + h * 16);
    c.sc.Kp = (lds_cp)(Ks + r * ldt + h * 16);
    c.sc.tb = (const LDS_AS f32x2*)((u0 & 1) ? tabR + CF::TABF + (u0 - 1) : tabR + u0);
    c.dOp = (lds_cp)(dOs + (tl * 32 + r) * ldt + h * 16);
    c.Vrow = (lds_cp)(Vs + r * ldt + h * 16);
    c.Ktr = (lds_cp)(Ks + trofs);
    c.part = (LDS_AS float*)(part + hl * 16 * 16 * 32 + (2 * tl + (r >> 4)) * 16 * 32 + (r & 15));
  }
  f32x16 dq;
  if (hl == 0) w16_bwd_p1s<D, 0, bf16>(c, dq);
  else w16_bwd_p1s<D, 1, bf16>(c, dq);
  __syncthreads();
  {  // d(table) of both heads: one thread per entry adds its <= 16 slots in fixed order
    float* slab_row = p.slab + ((int64_t)win * CF::HEADS + grp * 2) * 961;
    for (int idx = tid; idx < 2 * 961; idx += 1024) {
      const int h2 = idx >= 961 ? 1 : 0, e = idx - 961 * h2;
      const int ry = e / 31, rx = e - ry * 31;
      const bf16* ph = part + h2 * 16 * 16 * 32;
      float sum = 0.f;
      for (int yi = 0; yi < 16; ++yi) {
        const int yj = yi + 15 - ry;
        if (yj >= 0 && yj < 16) sum += __bfloat162float(ph[(yi * 16 + yj) * 32 + rx]);
      }
      slab_row[idx] = sum;
    }
  }
  {  // pass 2: wave = (key tile tl, head hl)
    c.kt = tl;
    c.QA = (lds_cp)(Qs + r * ldt + h * 16);
    c.dOA = (lds_cp)(dOs + r * ldt + h * 16);
    c.Qtr = (lds_cp)(Qs + trofs);
    c.dOtr = (lds_cp)(dOs + trofs);
    c.Kown = (lds_cp)(Ks + (tl * 32 + r) * ldt + h * 16);
    c.Vown = (lds_cp)(Vs + (tl * 32 + r) * ldt + h * 16);
    c.Kst = (lds_cp)(Ks + (tl * 32 + r) * ldt);
    c.Vst = (lds_cp)(Vs + (tl * 32 + r) * ldt);
    c.tb2 = (const LDS_AS f32x2*)((u0 & 1) ? tabN + CF::TABF + (u0 - 1) : tabN + u0);
  }
  // pass 2 reads K / V of the wave's own key rows into registers at its start and stores dK / dV over them at its end.  The
  // other head's wave of the same tile reads the same rows, but only its own channels count (its packs are masked with
  // compile-time constants) and only this wave's channels are written: no ordering between the two is needed.
  if (hl == 0) w16_bwd_p2<D, 0>(c);
  else w16_bwd_p2<D, 1>(c);
  __syncthreads();   // every wave is done with Q as an operand: the wave's own (tile, head) piece of the query rows takes dQ
  {
    const lds_cp qrow = (lds_cp)(Qs + (tl * 32 + r) * ldt);
    if (hl == 0) store_tile_rows<0, 0, D>(qrow, dq, p.scale, h);
    else store_tile_rows<(D & ~3), D, 2 * D>(qrow, dq, p.scale, h);
  }
  __syncthreads();
  // dQ | dK | dV (in place of Q / K / V) -> global rows of this wave's 16 tokens
#pragma unroll 1
  for (int idx = lane; idx < 16 * CPR; idx += 64) {
    const int row = idx / CPR, k3 = idx - row * CPR;
    const int s3 = k3 / CPS, k = k3 - s3 * CPS;
    const int t = wv * 16 + row;
    const int64_t tok = win_token16(b, wr, wc, t, g);
    char* dst = reinterpret_cast<char*>(p.dqkv + tok * p.ldq) + s3 * (CF::C * 2) + grp * CF::PB + k * GRAN;
    const char* src = smem + s3 * CF::SEC + (size_t)t * ldt + k * GRAN;
    *reinterpret_cast<CH*>(dst) = chunk_from_lds<CH>(src);
  }
}


#ifndef W16_ABL1
#define W16_ABL1 0   // ablations of wattn16_bwd1_kernel (tools/abl_build.sh): 1 no pass 1, 2 no pass 2, 4 no table images / row-sum reduce, 8 no global stores
#endif
// ---- ONE head per workgroup, two workgroups per CU -----------------------------------------------------
// wattn16_bwd3_kernel loads, computes and stores in sequence and holds 119-150 KB of LDS: one workgroup per CU, nothing overlaps
// its load / stage / store skeleton (40 % of its time).  With a single head the staged pieces are 20 / 40 B
// per row (LDS rows of 32 / 48 B: one / two k-steps per product where the pair's masked packs take two / three; head dim 15: the
// dword-aligned 32-byte window around the head's 30 bytes — an odd head starts on a 2-byte boundary —, the neighbour's channel in it
// masked out of every product and never stored), the tables,
// statistics and d(table) row sums of one head: 67 / 75 KB, so TWO 8-wave workgroups share a CU and one's skeleton runs
// beside the other's passes.  wave = tile (queries in pass 1, keys in pass 2); the arithmetic is w16_bwd_p1s / w16_bwd_p2 of the
// pair kernel with the pair's head 0 as the only head.  The natural table (pass 2) is staged over the row sums (pass 1) after
// they have been reduced: its values wait in two registers per thread from the start.
// Measured (tools/w16_bench.py, B = 8 of 128 x 128, per call with the table reduce, one box): pair kernel 194 / 206 us -> 181 / 182
// (D = 10 / 20); with ONE of these workgroups per CU (40 KB of dummy LDS) 244 / 251: the second workgroup is worth 70 us.
// Ablations (W16_ABL1), D = 10 / 20: whole 181 / 182, no pass 1 118 / 118, no pass 2 124 / 126, neither 64 / 75, and without the
// table images / row-sum reduce 52 / 65, and without the global stores 33 / 40 — still additive: the passes (121 us) are
// bound by their vector instructions, so a second workgroup fills their stalls but cannot hide work; an offset between
// the two workgroups of a CU (s_sleep in the second one of the first round) changed nothing (179-187 us).
template <int D>
struct W16B1 {
  using CF = W16<D, 1>;
  static constexpr int SEC = CF::SEC;
  static constexpr int OFF_TABR = 4 * SEC + 64;
  static constexpr int OFF_STAT = OFF_TABR + 2 * CF::TABF * 4;      // [nlse | -delta | (unused)][256]
  static constexpr int OFF_PART = OFF_STAT + 3 * 256 * 4;           // [yi 16][yj 16][32] bf16 (pass 1), then the natural table (pass 2)
  static constexpr size_t SMEM = (size_t)OFF_PART + 16 * 16 * 32 * 2;
  static_assert(2 * CF::TABF * 4 <= 16 * 16 * 32 * 2, "the natural table lies over the row sums");
};

template <int D>
__global__ void __launch_bounds__(512, 4) wattn16_bwd1_kernel(const W16Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using CF = W16<D, 1>;
  using BF = W16B1<D>;
  using CH = typename Chunk<CF::GRAN>::type;
  constexpr int ldt = CF::LDT, GRAN = CF::GRAN, CPS = CF::CPS, CPR = 3 * CPS, RPI = 64 / CPR, NI = 32 / RPI;
  constexpr int RPD = 64 / CPS, ND = (32 + RPD - 1) / RPD;
  static_assert(32 % RPI == 0, "a wave stages the 32 token rows of its tile");
  const WinGeom g = p.g;
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int tl = __builtin_amdgcn_readfirstlane(tid >> 6);   // this wave's tile
  char* Qs = smem;
  char* Ks = Qs + CF::SEC;
  char* Vs = Ks + CF::SEC;
  char* dOs = Vs + CF::SEC;
  float* tabR = reinterpret_cast<float*>(smem + BF::OFF_TABR);
  float* stat = reinterpret_cast<float*>(smem + BF::OFF_STAT);
  bf16* part = reinterpret_cast<bf16*>(smem + BF::OFF_PART);
  float* tabN = reinterpret_cast<float*>(smem + BF::OFF_PART);

  const int nW = g.nWh * g.nWw, nwin = g.B * nW;
  int win, hd;   // the six heads of a window run on the same XCD (blockIdx round-robins over 8 XCDs)
  if ((nwin & 7) == 0) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    hd = slot % 6;
    win = (slot / 6) * 8 + xcd;
  } else {
    hd = blockIdx.x % 6;
    win = blockIdx.x / 6;
  }
  const int b = win / nW, wi = win - b * nW, wr = wi / g.nWw, wc = wi - wr * g.nWw;
  // byte offset of the head's piece inside a row section; head dim 15: the 32-byte window that starts 2 bytes early for odd heads
  const int par = (D & 1) ? (hd & 1) : 0;
  const int pofs = (D & 1) ? hd * (2 * D) - 2 * par : hd * CF::PB;

  // ---- this wave's 32 token rows of qkv and dO -> registers
  const int lr0 = lane / CPR, ch = lane - lr0 * CPR;
  const bool act = lr0 < RPI;
  const int lr = act ? lr0 : RPI - 1;
  const int sec = ch / CPS, cw = ch - sec * CPS;
  CH regs[NI], dreg[ND];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int t = tl * 32 + i * RPI + lr;
    const int64_t tok = win_token16(b, wr, wc, t, g);
    const char* src = reinterpret_cast<const char*>(p.qkv + tok * p.ld) + sec * (CF::C * 2) + pofs + cw * GRAN;
    regs[i] = *reinterpret_cast<const CH*>(src);
  }
  const int dr0 = lane / CPS, dc = lane - dr0 * CPS;
  const bool dact0 = dr0 < RPD;
#pragma unroll
  for (int i = 0; i < ND; ++i) {
    int rr = i * RPD + (dact0 ? dr0 : RPD - 1);
    rr = rr < 32 ? rr : 31;
    const int64_t tok = win_token16(b, wr, wc, tl * 32 + rr, g);
    const char* src = reinterpret_cast<const char*>(p.dout + tok * p.ldd) + pofs + dc * GRAN;
    dreg[i] = *reinterpret_cast<const CH*>(src);
  }
  // (nlse, -delta) of query tile tl: lane r = query, both lane halves compute the same
  const int qown = tl * 32 + r;
  const int64_t tokq = win_token16(b, wr, wc, qown, g);
  float dl = 0.f;
  const float nl_own = p.nlse[tokq * CF::HEADS + hd];
  {
    constexpr int NDW = (D & 1) ? 8 : D / 2;   // dwords of the head's piece (dim 15: the 16-channel window)
    const uint32_t* orow = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(p.o + tokq * p.ldo2) + pofs);
    const uint32_t* drow = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(p.dout + tokq * p.ldd) + pofs);
    uint32_t ov[NDW], dv[NDW];
#pragma unroll
    for (int i = 0; i < NDW; ++i) { ov[i] = orow[i]; dv[i] = drow[i]; }
#pragma unroll
    for (int i = 0; i < NDW; ++i) {
      dl = fmaf(bf16lo(dv[i]), bf16lo(ov[i]), dl);
      dl = fmaf(bf16hi(dv[i]), bf16hi(ov[i]), dl);
    }
    if constexpr (D & 1)   // the neighbour's channel in the window: the last one of an even head's, the first one of an odd head's
      dl -= par ? bf16lo(dv[0]) * bf16lo(ov[0]) : bf16hi(dv[NDW - 1]) * bf16hi(ov[NDW - 1]);
  }
  constexpr float LOG2E = 1.4426950408889634f;
  const float rscale = 1.0f / p.scale;
  // the head's bias table / scale: two entries per thread, the reversed image (pass 1, two copies shifted by one float) now,
  // the natural one after pass 1
  constexpr int NLD = (961 + 511) / 512;
  float tv[NLD];
#pragma unroll
  for (int k = 0; k < NLD; ++k) {
    const int j = tid + 512 * k;
    tv[k] = p.table[(j < 961 ? j : 960) * CF::HEADS + hd] * rscale;
  }
#pragma unroll
  for (int k = 0; k < NLD; ++k) {
    const int j = tid + 512 * k;
    if (j < 961 && !(W16_ABL1 & 4)) {
      const int ry = j / 31, rx = j - ry * 31;
      const int ir = (30 - ry) * CF::TROW + (30 - rx);
      tabR[ir] = tv[k];
      if (ir >= 1) tabR[CF::TABF + ir - 1] = tv[k];
    }
  }
  if (tid < 16) *reinterpret_cast<uint32_t*>(dOs + CF::SEC + 4 * tid) = 0u;   // guard behind the last dO row
  if constexpr (ldt > CF::PB) {   // zero the pad bytes of every staged row
    constexpr int padw = (ldt - CF::PB) / 4;
    for (int idx = tid; idx < 4 * 256 * padw; idx += 512) {
      const int row = idx / padw, w = idx - row * padw;
      *reinterpret_cast<uint32_t*>(Qs + (size_t)row * ldt + CF::PB + 4 * w) = 0u;
    }
  }
  if (act) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int t = tl * 32 + i * RPI + lr;
      chunk_to_lds<CH>(smem + sec * CF::SEC + t * ldt + cw * GRAN, regs[i]);
    }
  }
#pragma unroll
  for (int i = 0; i < ND; ++i) {
    const int rr = i * RPD + dr0;
    if (dact0 && rr < 32) chunk_to_lds<CH>(dOs + (tl * 32 + rr) * ldt + dc * GRAN, dreg[i]);
  }
  if (h == 0) {
    stat[qown] = nl_own;
    stat[256 + qown] = -dl;
  }
  __syncthreads();

  W16BCtx c;
  c.sc.h = h; c.sc.r = r;
  c.sc.thr = g.ws - g.shift;
  c.sc.mrow = g.shift > 0 && wr == g.nWh - 1;
  c.sc.mcol = g.shift > 0 && wc == g.nWw - 1;
  c.sc.masked = c.sc.mrow || c.sc.mcol;
  c.sc.cbits = (uint32_t)__builtin_bit_cast(uint16_t, (__bf16)(100.0f * rscale));
  c.sc.scale2 = p.scale * LOG2E;
  c.scale = p.scale;
  c.stat = (LDS_AS float*)stat;
  const int trofs = (4 * h + ((lane & 15) >> 2)) * ldt + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
  const int y0 = 2 * tl + (r >> 4), x0 = r & 15;
  const int u0 = (15 - y0) * CF::TROW + 15 - x0 + 4 * h;
  {  // pass 1: wave = query tile tl
    c.sc.qt = tl;
    c.sc.Qp = (lds_cp)(Qs + (tl * 32 + r) * ldt + h * 16);
    c.sc.Kp = (lds_cp)(Ks + r * ldt + h * 16);
    c.sc.tb = (const LDS_AS f32x2*)((u0 & 1) ? tabR + CF::TABF + (u0 - 1) : tabR + u0);
    c.dOp = (lds_cp)(dOs + (tl * 32 + r) * ldt + h * 16);
    c.Vrow = (lds_cp)(Vs + r * ldt + h * 16);
    c.Ktr = (lds_cp)(Ks + trofs);
    c.part = (LDS_AS float*)(part + (2 * tl + (r >> 4)) * 16 * 32 + (r & 15));
  }
  f32x16 dq;
  if (W16_ABL1 & 1) {
#pragma unroll
    for (int v = 0; v < 16; ++v) dq[v] = 0.f;
  } else {
    if (par) w16_bwd_p1s<D, (D & 1), bf16, 1>(c, dq);
    else w16_bwd_p1s<D, 0, bf16, 1>(c, dq);
  }
  __syncthreads();
  if (!(W16_ABL1 & 4)) {  // d(table): one thread per entry adds its <= 16 slots in fixed order
    float* slab_row = p.slab + ((int64_t)win * CF::HEADS + hd) * 961;
    float sums[NLD];
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int e = tid + 512 * k;
      const int ee = e < 961 ? e : 960;
      const int ry = ee / 31, rx = ee - ry * 31;
      float sum = 0.f;
      for (int yi = 0; yi < 16; ++yi) {
        const int yj = yi + 15 - ry;
        if (yj >= 0 && yj < 16) sum += __bfloat162float(part[(yi * 16 + yj) * 32 + rx]);
      }
      sums[k] = sum;
      if (e < 961) slab_row[e] = sum;
    }
    (void)sums;
  }
  __syncthreads();   // the row sums have been read: the natural table takes their place
#pragma unroll
  for (int k = 0; k < NLD; ++k) {
    const int j = tid + 512 * k;
    if (j < 961 && !(W16_ABL1 & 4)) {
      const int ry = j / 31, rx = j - ry * 31;
      const int in = ry * CF::TROW + rx;
      tabN[in] = tv[k];
      if (in >= 1) tabN[CF::TABF + in - 1] = tv[k];
    }
  }
  __syncthreads();
  {  // pass 2: wave = key tile tl
    c.kt = tl;
    c.QA = (lds_cp)(Qs + r * ldt + h * 16);
    c.dOA = (lds_cp)(dOs + r * ldt + h * 16);
    c.Qtr = (lds_cp)(Qs + trofs);
    c.dOtr = (lds_cp)(dOs + trofs);
    c.Kown = (lds_cp)(Ks + (tl * 32 + r) * ldt + h * 16);
    c.Vown = (lds_cp)(Vs + (tl * 32 + r) * ldt + h * 16);
    c.Kst = (lds_cp)(Ks + (tl * 32 + r) * ldt);
    c.Vst = (lds_cp)(Vs + (tl * 32 + r) * ldt);
    c.tb2 = (const LDS_AS f32x2*)((u0 & 1) ? tabN + CF::TABF + (u0 - 1) : tabN + u0);
  }
  if (!(W16_ABL1 & 2)) {
    if (par) w16_bwd_p2<D, (D & 1), 1>(c);
    else w16_bwd_p2<D, 0, 1>(c);
  }
  __syncthreads();   // every wave is done with Q as an operand: the wave's own tile of the query rows takes dQ
  if (par) store_tile_rows<0, (D & 1), (D & 1) + D>((lds_cp)(Qs + (tl * 32 + r) * ldt), dq, p.scale, h);
  else store_tile_rows<0, 0, D>((lds_cp)(Qs + (tl * 32 + r) * ldt), dq, p.scale, h);
  __syncthreads();
  // dQ | dK | dV (in place of Q / K / V) -> global rows of this wave's 32 tokens
#pragma unroll 1
  for (int idx = lane; idx < 32 * CPR; idx += 64) {
    const int row = idx / CPR, k3 = idx - row * CPR;
    const int s3 = k3 / CPS, k = k3 - s3 * CPS;
    const int t = tl * 32 + row;
    const int64_t tok = win_token16(b, wr, wc, t, g);
    char* dst = reinterpret_cast<char*>(p.dqkv + tok * p.ldq) + s3 * (CF::C * 2) + pofs + k * GRAN;
    const char* src = smem + s3 * CF::SEC + (size_t)t * ldt + k * GRAN;
    if (W16_ABL1 & 8) continue;
    if constexpr (D & 1) {   // the dword that holds the neighbour's channel leaves as the one 2-byte half that is this head's
      const uint32_t v = *reinterpret_cast<const uint32_t*>(src);
      if (k != (par ? 0 : CPS - 1)) *reinterpret_cast<uint32_t*>(dst) = v;
      else if (par) *reinterpret_cast<uint16_t*>(dst + 2) = (uint16_t)(v >> 16);
      else *reinterpret_cast<uint16_t*>(dst) = (uint16_t)v;
    } else {
      *reinterpret_cast<CH*>(dst) = chunk_from_lds<CH>(src);
    }
  }
}

#ifndef W16_BWD1
#define W16_BWD1 1   // with the forward's statistics: 1 = one head per workgroup, two workgroups per CU; 0 = the pair kernel below
#endif
#ifndef W16_BWD3
#define W16_BWD3 1   // with the forward's statistics: 1 = the 16-wave kernel (both heads of the pair side by side), 0 = the 8-wave kernel with
#endif               // the streaming first pass
template <int D>
int launch_bwd16(const W16Args& p, hipStream_t st) {
  {
    if (W16_BWD1 && p.nlse && p.o) {
      auto k1 = wattn16_bwd1_kernel<D>;
      constexpr size_t smem1 = W16B1<D>::SMEM;
      static_assert(smem1 <= 80 * 1024, "two workgroups per CU");
      (void)hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem1);
      const int64_t nwin1 = (int64_t)p.g.B * p.g.nWh * p.g.nWw;
      hipLaunchKernelGGL(k1, dim3((unsigned)(6 * nwin1)), dim3(512), smem1, st, p);
      return rdst_launch_status("wattn16_bwd1");
    }
  }
  if (W16_BWD3 && p.nlse && p.o) {
    auto k3 = wattn16_bwd3_kernel<D>;
    constexpr size_t smem3 = W16B3<D>::SMEM;
    static_assert(smem3 <= 160 * 1024, "LDS");
    (void)hipFuncSetAttribute((const void*)k3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem3);
    const int64_t nwin3 = (int64_t)p.g.B * p.g.nWh * p.g.nWw;
    hipLaunchKernelGGL(k3, dim3((unsigned)(3 * nwin3)), dim3(1024), smem3, st, p);
    return rdst_launch_status("wattn16_bwd3");
  }
  auto kern = (p.nlse && p.o) ? wattn16_bwd_kernel<D, true> : wattn16_bwd_kernel<D, false>;
  constexpr size_t smem = W16B<D>::SMEM;
  static_assert(smem <= 160 * 1024, "LDS");
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  const int64_t nwin = (int64_t)p.g.B * p.g.nWh * p.g.nWw;
  hipLaunchKernelGGL(kern, dim3((unsigned)(3 * nwin)), dim3(512), smem, st, p);
  return rdst_launch_status("wattn16_bwd");
}

bool al(const void* a, int64_t lda_bytes, int gsz) { return (uintptr_t)a % gsz == 0 && lda_bytes % gsz == 0; }

}  // namespace

// bf16, ws 16, 6 heads of dim 10 / 15 / 20, no explicit mask, scale > 0; RDST_ENOTSUP otherwise
int wattn16_fwd_mfma(const void* qkv, int64_t ld, const float* table, void* out, int64_t ldo, const WinGeom& g,
                     float scale, hipStream_t st, float* nlse) {
  if (g.ws != 16 || g.heads != 6 || g.mask || !(scale > 0.f) || g.C % 6) return RDST_ENOTSUP;
  if ((int64_t)g.B * g.nWh * g.nWw * 3 > 0x7fffffff) return RDST_ENOTSUP;
  W16Args p{};
  p.qkv = (const bf16*)qkv; p.ld = ld; p.table = table; p.out = (bf16*)out; p.ldo = ldo; p.g = g; p.scale = scale;
  p.nlse_out = nlse;
  const int d = g.C / 6;
  if (d == 10 && al(qkv, ld * 2, 8) && al(out, ldo * 2, 8)) return launch_fwd16<10>(p, st);
  if (d == 15 && al(qkv, ld * 2, 4) && al(out, ldo * 2, 4)) return launch_fwd16<15>(p, st);
  if (d == 20 && al(qkv, ld * 2, 16) && al(out, ldo * 2, 16)) return launch_fwd16<20>(p, st);
  return RDST_ENOTSUP;
}

// slab: [windows][6][961] partial d(table), one row per window (*nslab = windows)
int wattn16_bwd_mfma(const void* qkv, int64_t ld, const float* table, const void* dout, int64_t ldd, void* dqkv,
                     int64_t ldq, float* slab, int slab_rows, const WinGeom& g, float scale, int* nslab, hipStream_t st,
                     const void* o, int64_t ldo, const float* nlse) {
  if (g.ws != 16 || g.heads != 6 || g.mask || !(scale > 0.f) || g.C % 6) return RDST_ENOTSUP;
  const int64_t nwin = (int64_t)g.B * g.nWh * g.nWw;
  if (nwin * 6 > 0x7fffffff || slab_rows < nwin) return RDST_ENOTSUP;
  W16Args p{};
  p.qkv = (const bf16*)qkv; p.ld = ld; p.table = table; p.dout = (const bf16*)dout; p.ldd = ldd;
  p.dqkv = (bf16*)dqkv; p.ldq = ldq; p.slab = slab; p.g = g; p.scale = scale;
  p.o = (const bf16*)o; p.ldo2 = ldo; p.nlse = nlse;
  if (o && !al(o, ldo * 2, 4)) return RDST_ENOTSUP;   // (delta reads the head's piece of the output row with dword loads)
  if ((o == nullptr) != (nlse == nullptr)) return RDST_ENOTSUP;
  *nslab = (int)nwin;
  const int d = g.C / 6;
  const int a = d == 10 ? 8 : d == 15 ? 4 : 16;
  if (!(al(qkv, ld * 2, a) && al(dout, ldd * 2, a) && al(dqkv, ldq * 2, a))) return RDST_ENOTSUP;
  if (d == 10) return launch_bwd16<10>(p, st);
  if (d == 15) return launch_bwd16<15>(p, st);
  if (d == 20) return launch_bwd16<20>(p, st);
  return RDST_ENOTSUP;
}
