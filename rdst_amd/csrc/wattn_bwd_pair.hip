// K2 (window attention backward), round 4: bf16, 8x8 windows, 6 heads of dim D = 10 / 15 / 20, dense rows.
//
// Work item = (window, head PAIR): 40 / 60 / 80 bytes of every q / k / v / dOut row of the window.  A 4-wave workgroup
// (wave = (token tile t, head of the pair)) owns one item at a time and THREE workgroups share a CU (<= 52 KB of LDS,
// 168 registers), each at its own point of its item: one workgroup's row traffic overlaps the others' arithmetic.
//
// What the rounds 1-3 kernel (wattn_bwd_mfma_hd.hip, one 12-wave workgroup per window) could not do, and why this cut:
//   * Its no-compute skeleton took 46-62 us of its 73-82: a CU had ONE window of loads in flight, fetched into
//     registers (which the arithmetic phases need) less than a load latency before their use, and a wave's gradient
//     stores sat in front of its next loads on the in-order memory counter.
//     Here the next item's rows arrive by LDS-DMA (buffer_load ... lds) into a SECOND section set, issued a whole item
//     ahead and awaited with a COUNTED s_waitcnt vmcnt(N): the item's own gradient stores are younger than the DMA and
//     are never waited for.  No vector register ever holds an input row.
//   * The second section set has to fit: the P / dS images ([query][key] bf16 round trip through LDS that turned the
//     query-tile softmax into key-tile operands, 32 KB per head pair) are gone.  Pass T (wave = query tile: S^T, softmax,
//     dP^T, delta, dS^T -> dQ^T and the d(table) sums, exactly as before) leaves two floats per query in LDS — the row's
//     log-sum-exp and -delta — and pass N (wave = key tile) REBUILDS its logits in the other orientation,
//     S = Q.K^T + bias, P = exp2(S - lse), dS = P.(dP - delta) with -delta as the initial accumulator of dP = dO.V^T:
//     the accumulators already ARE the B operands of dV^T = dO^T.P and dK^T = Q^T.dS.  N is T's code with (Q, dO) and
//     (K, V) swapped and without the row reductions; one 4-wave barrier between the passes instead of two 12-wave ones
//     around an LDS round trip.
// Rows are staged as they lie (16-byte slots, row stride 48 / 80 / 80 bytes = an odd number of slots); a row piece that
// does not end on a slot boundary drags the first channels of the next head pair along: finite values that the
// compile-time channel masks of the Q / dO packs (T) and K / V packs (N) multiply by zero — which is why the rows must be
// DENSE (ld = 3C / C): the bytes behind a piece are then activations, never uninitialised memory.
#include "wattn_hd.h"
#include <stdlib.h>

namespace {
using namespace wahd;

#ifndef K2P_NH
#define K2P_NH 6   // heads per workgroup: 6 = one 12-wave workgroup per window (whole rows in, whole rows out); 2 = (window, head
                   // pair) items, three 4-wave workgroups per CU — measured round 4: its 40-80-byte row pieces are partial-line
                   // stores, 36 us per launch for the stores alone whatever the width
#endif
constexpr int HEADS_ALL = 6, NH = K2P_NH, NWV = 2 * NH, NTH = 64 * NWV;
constexpr int NG = HEADS_ALL / NH;   // head groups per window
constexpr int NLW = NH == 6 ? 4 : 2; // loader waves (0 .. NLW-1 issue every LDS-DMA piece, and nothing else that touches
                                     // vector memory); the other NWV - NLW waves issue every gradient store

struct PArgs {
  const bf16* qkv; const bf16* dout; bf16* dqkv;
  const float* table; float* slab;
  int64_t ld, ldd, ldq;          // elements
  uint32_t qkv_bytes, dout_bytes;
  int G;                         // window groups: workgroup (group, pair) walks windows group, group + G, ...
  WinGeom g;
  float scale;
};

template <int D>
struct PC {
  static constexpr int C2 = NH * D, ROWB = C2 * 2;          // channels / bytes of a pair's piece of a row
  static constexpr int SD = (ROWB + 15) / 16;               // 16-byte slots that hold data: 3 / 4 / 5
  static constexpr int TAILB = ROWB - 16 * (SD - 1);        // bytes of the piece inside its last slot: 8 / 12 / 16
  static constexpr int S = SD | 1;                          // slots per LDS row (odd: conflict-free ds_read_b128): 3 / 5 / 5
  static constexpr int LDT = 16 * S;
  static constexpr int RPP = 64 / S;                        // rows per DMA piece (RPP * S lanes active): 21 / 12 / 12
  static constexpr int PPS = (64 + RPP - 1) / RPP;          // pieces per section: 4 / 6 / 6
  static constexpr int NPIECE = 4 * PPS;                    // pieces per item
  static constexpr int PPW = NPIECE / NLW;                  // pieces per loader wave and item
  static constexpr int SECB = 64 * LDT, BUFB = 4 * SECB;    // section order inside a buffer: Q, K, V, dOut
  static constexpr int NBUF = (NH == 6 ? BUFB <= 40 * 1024 : BUFB <= 12 * 1024) ? 3 : 2;    // ring depth: NBUF - 1 items in flight per workgroup
  static constexpr int OFF_Q = 0, OFF_K = SECB, OFF_V = 2 * SECB, OFF_DO = 3 * SECB;
  static constexpr int OFF_TAB = NBUF * BUFB + 64;          // (64 zero bytes behind the last buffer: k-steps that run past a row)
  static constexpr int TABF = NH * 15 * TSX, TABC = TABF + 8;   // floats per staged table copy (natural order; + one shifted by a column)
  static constexpr int OFF_ID = OFF_TAB + 2 * TABC * 4;     // [64 lanes][2 x 16 B] 0/1 operand packs of the d(table) MFMA
  static constexpr int OFF_ST = OFF_ID + 64 * 32;           // lse2[NH][64], -delta[NH][64]
  static constexpr int SMEM_MAIN = OFF_ST + 2 * NH * 64 * 4;
  static constexpr int SMEM_EPI = NH * 64 * 65 * 4;
  static constexpr int SMEM = SMEM_MAIN > SMEM_EPI ? SMEM_MAIN : SMEM_EPI;
  static constexpr int CFULL = HEADS_ALL * D;
  // copy-out: the NWV - NLW storer waves share the 64 rows
  static constexpr int RPS = 64 / (NWV - NLW);              // rows per storer wave
  static constexpr int NCH = RPS * 3 * SD, NPASS = (NCH + 63) / 64;   // NH = 2: chunk = (row, section, data slot)
  static constexpr int ROW3B = 3 * ROWB;                    // NH = 6: a whole dqkv row per store instruction, lane = 16-byte chunk
  static_assert(NH != 6 || ROW3B <= 64 * 16, "a gradient row per wave instruction");
  static constexpr int og(int hd) { return (hd * D) & ~3; }
  static_assert(NPIECE % NLW == 0, "every loader wave issues the same number of pieces");
  static_assert(PPW * (NBUF - 1) < 64, "vmcnt is a 6-bit counter");
  static_assert(SMEM <= (NH == 6 ? 160 : 53) * 1024, "one 12-wave workgroup / three 4-wave workgroups per CU");
};

struct Ctx {
  lds_cp own;     // buffer + tok * ldt                      (the lane's own token row; + section offset)
  lds_cp rowA;    // buffer + r * ldt + 16 h                  (A-operand row packs of a 32-token tile; + section + tile)
  lds_cp tr;      // buffer + (4 h + q) * ldt + (16 (gq & 1) + 4 pp) * 2   (transposed-read position)
  lds_cp idp;
  const LDS_AS f32x2* tb;    // natural-order table: pass T position (pairs come back swapped), pass N position
  const LDS_AS f32x2* tn;
  const LDS_AS float* st;    // statistics of the wave's head
  int h;
  bool masked, mrow, mcol;
  int thr;
  float scale2, scale;
  int yi, xi;
  uint32_t cbits;
};

// shifted-window mask as one more k-step on both 32-row tiles of X (see wattn_mfma_hd.hip): +100/scale where the regions
// of the lane's token and of the tile rows' tokens AGREE; symmetric in (query, key), so both passes use it
__device__ __forceinline__ void mask_step(const Ctx& c, f32x16 (&X)[2]) {
  const int h = c.h;
  auto onehot = [&](int reg, uint32_t v) {
    Pack16 q;
    q.w[0] = h ? 0u : ((reg == 0 ? v : 0u) | (reg == 1 ? v << 16 : 0u));
    q.w[1] = h ? 0u : ((reg == 2 ? v : 0u) | (reg == 3 ? v << 16 : 0u));
    q.w[2] = 0u;
    q.w[3] = 0u;
    return q;
  };
  const int rx = (c.mcol && c.xi >= c.thr) ? 1 : 0;
  const Pack16 mB = onehot(2 * ((c.mrow && c.yi >= c.thr) ? 1 : 0) + rx, c.cbits);
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
    Mma<bf16>::mma(X[kt], onehot(2 * ((c.mrow && kt * 4 + (c.yi & 3) >= c.thr) ? 1 : 0) + rx, 0x3f80u), mB);
}

// pass T of one (query tile, head): dQ^T tile, d(table) sums, row statistics (lse2, -delta) -> LDS
template <int D, int HD>
__device__ __forceinline__ void pass_t(const Ctx& c, f32x16& dq, f32x16 (&Dsum)[2]) {
  using CF = PC<D>;
  constexpr int ldt = CF::LDT;
  constexpr int c_lo = HD * D, c_hi = c_lo + D;
  constexpr int t_lo = c_lo / 16, t_hi = (c_hi - 1) / 16;
  constexpr int OG = CF::og(HD);
  const int h = c.h;
  f32x16 X[2], Y[2];
  const LDS_AS f32x2* tbh = c.tb + HD * (15 * TSX / 2);
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int v = 0; v < 16; v += 2) {
      // bias of (query on the lane, key kt*4 + (v >> 2), 4 h + (v & 3)): natural column xi + 7 - 4 h - (v & 3) — the pair
      // (v, v + 1) lies at descending columns, so the aligned 8-byte read returns it swapped
      const f32x2 b2 = lds_read_f32x2(tbh + ((7 - (kt * 4 + (v >> 2))) * TSX + (2 - (v & 3))) / 2);
      X[kt][v] = b2.y;
      X[kt][v + 1] = b2.x;
      Y[kt][v] = 0.f;
      Y[kt][v + 1] = 0.f;
    }
#pragma unroll
  for (int t = t_lo; t <= t_hi; ++t) {
    Pack16 qb = lds_pack(c.own + CF::OFF_Q + h * 16 + t * 32), gb = lds_pack(c.own + CF::OFF_DO + h * 16 + t * 32);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t mA = qmask_bits(t * 16 + 2 * e, c_lo, c_hi), mB = qmask_bits(t * 16 + 8 + 2 * e, c_lo, c_hi);
      const uint32_t m = h ? mB : mA;
      qb.w[e] &= m;
      gb.w[e] &= m;
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      const Pack16 ka = lds_pack(c.rowA + CF::OFF_K + kt * 32 * ldt + t * 32);
      const Pack16 va = lds_pack(c.rowA + CF::OFF_V + kt * 32 * ldt + t * 32);
      Mma<bf16>::mma(X[kt], ka, qb);   // S'^T = K . Q^T  (+ bias / scale)
      Mma<bf16>::mma(Y[kt], va, gb);   // dP^T = V . dO^T
    }
  }
  if (c.masked) mask_step(c, X);
  float m = X[0][0];
#pragma unroll
  for (int v = 1; v < 16; ++v) m = __builtin_fmaxf(m, X[0][v]);
#pragma unroll
  for (int v = 0; v < 16; ++v) m = __builtin_fmaxf(m, X[1][v]);
  m = half_swap_max(m);
  const float nm = -c.scale2 * m;
  float l0 = 0.f, l1 = 0.f, d0 = 0.f, d1 = 0.f;
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int v = 0; v < 16; v += 2) {
      const float e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(X[kt][v], c.scale2, nm));
      const float e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(X[kt][v + 1], c.scale2, nm));
      l0 += e0;
      l1 += e1;
      d0 = __builtin_fmaf(e0, Y[kt][v], d0);          // l * delta = sum_j e dP
      d1 = __builtin_fmaf(e1, Y[kt][v + 1], d1);
      X[kt][v] = e0;
      X[kt][v + 1] = e1;
    }
  const float l = half_swap_sum(l0 + l1);
  const float inv = __builtin_amdgcn_rcpf(l);
  const float delta = half_swap_sum(d0 + d1) * inv;
  if (h == 0) {   // both lane halves hold the row's totals: P[q][k] = exp2(scale2 S' - lse2) in pass N
    const int tok = (c.yi * 8 + c.xi);
    const_cast<LDS_AS float*>(c.st)[tok] = __builtin_amdgcn_logf(l) - nm;
    const_cast<LDS_AS float*>(c.st)[NH * 64 + tok] = -delta;
  }
  // dS = P (dP - delta) = e ((dP - delta) / l), times `scale` (dQ = scale K^T dS^T: the factor rides on the two per-row
  // constants instead of on the 16 values of the dQ tile; the d(table) sums take it out again in the epilogue)
  const float invs = inv * c.scale;
  const float ndi = -delta * invs;
  Pack16 pdS[2][2];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int v0 = 8 * s + 2 * e;
        pdS[kt][s].w[e] = pack_bf16x2(X[kt][v0] * __builtin_fmaf(Y[kt][v0], invs, ndi), X[kt][v0 + 1] * __builtin_fmaf(Y[kt][v0 + 1], invs, ndi));
      }
  // dQ^T (rows = channels OG .. OG+31, cols = queries) = K^T . dS^T;  d(table) partial sums += I . dS^T
#pragma unroll
  for (int v = 0; v < 16; ++v) dq[v] = 0.f;
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const lds_cp kb = c.tr + CF::OFF_K + OG * 2 + (kt * 32 + 16 * s) * ldt;
      const Pack16 ka = lds_tr_pack(kb, kb + 8 * ldt);
      Mma<bf16>::mma(dq, ka, pdS[kt][s]);
      Mma<bf16>::mma(Dsum[kt], lds_pack(c.idp + 16 * s), pdS[kt][s]);
    }
}

// pass N of one (key tile, head): dV^T = dO^T . P and dK^T = Q^T . dS with the logits rebuilt key-on-the-lane
template <int D, int HD>
__device__ __forceinline__ void pass_n(const Ctx& c, f32x16& dv, f32x16& dk) {
  using CF = PC<D>;
  constexpr int ldt = CF::LDT;
  constexpr int c_lo = HD * D, c_hi = c_lo + D;
  constexpr int t_lo = c_lo / 16, t_hi = (c_hi - 1) / 16;
  constexpr int OG = CF::og(HD);
  const int h = c.h;
  f32x16 X[2], Y[2];
  const LDS_AS f32x2* tnh = c.tn + HD * (15 * TSX / 2);
  const LDS_AS f32x4* st4 = reinterpret_cast<const LDS_AS f32x4*>(c.st) + h;   // queries qt*32 + 8 g4 + 4 h + (0..3)
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const f32x4 nd = st4[(NH * 64) / 4 + qt * 8 + 2 * g4];
#pragma unroll
      for (int e = 0; e < 4; e += 2) {
        const f32x2 b2 = lds_read_f32x2(tnh + ((qt * 4 + g4) * TSX + e) / 2);
        X[qt][4 * g4 + e] = b2.x;
        X[qt][4 * g4 + e + 1] = b2.y;
      }
      Y[qt][4 * g4] = nd.x; Y[qt][4 * g4 + 1] = nd.y; Y[qt][4 * g4 + 2] = nd.z; Y[qt][4 * g4 + 3] = nd.w;
    }
#pragma unroll
  for (int t = t_lo; t <= t_hi; ++t) {
    Pack16 kb = lds_pack(c.own + CF::OFF_K + h * 16 + t * 32), vb = lds_pack(c.own + CF::OFF_V + h * 16 + t * 32);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t mA = qmask_bits(t * 16 + 2 * e, c_lo, c_hi), mB = qmask_bits(t * 16 + 8 + 2 * e, c_lo, c_hi);
      const uint32_t m = h ? mB : mA;
      kb.w[e] &= m;
      vb.w[e] &= m;
    }
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      const Pack16 qa = lds_pack(c.rowA + CF::OFF_Q + qt * 32 * ldt + t * 32);
      const Pack16 ga = lds_pack(c.rowA + CF::OFF_DO + qt * 32 * ldt + t * 32);
      Mma<bf16>::mma(X[qt], qa, kb);   // S'  = Q . K^T  (+ bias / scale)
      Mma<bf16>::mma(Y[qt], ga, vb);   // dP - delta = dO . V^T - delta
    }
  }
  if (c.masked) mask_step(c, X);
  Pack16 pP[2][2], pdS[2][2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const f32x4 ls = st4[qt * 8 + 2 * g4];
      const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(X[qt][4 * g4], c.scale2, -ls.x));
      const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(X[qt][4 * g4 + 1], c.scale2, -ls.y));
      const float p2 = __builtin_amdgcn_exp2f(__builtin_fmaf(X[qt][4 * g4 + 2], c.scale2, -ls.z));
      const float p3 = __builtin_amdgcn_exp2f(__builtin_fmaf(X[qt][4 * g4 + 3], c.scale2, -ls.w));
      const int s = g4 >> 1, e = 2 * (g4 & 1);
      pP[qt][s].w[e] = pack_bf16x2(p0, p1);
      pP[qt][s].w[e + 1] = pack_bf16x2(p2, p3);
      pdS[qt][s].w[e] = pack_bf16x2(p0 * Y[qt][4 * g4], p1 * Y[qt][4 * g4 + 1]);
      pdS[qt][s].w[e + 1] = pack_bf16x2(p2 * Y[qt][4 * g4 + 2], p3 * Y[qt][4 * g4 + 3]);
    }
#pragma unroll
  for (int v = 0; v < 16; ++v) {
    dv[v] = 0.f;
    dk[v] = 0.f;
  }
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const lds_cp gb = c.tr + CF::OFF_DO + OG * 2 + (qt * 32 + 16 * s) * ldt;
      const lds_cp qb = c.tr + CF::OFF_Q + OG * 2 + (qt * 32 + 16 * s) * ldt;
      Mma<bf16>::mma(dv, lds_tr_pack(gb, gb + 8 * ldt), pP[qt][s]);
      Mma<bf16>::mma(dk, lds_tr_pack(qb, qb + 8 * ldt), pdS[qt][s]);
    }
}

#ifndef K2P_ABL
#define K2P_ABL 0   // compile-time ablations (tools/abl_build.sh): 1 no pass T, 2 no pass N, 4 no tile stores to LDS, 8 no copy-out, 16 no DMA after the first ring fill
#endif

#define K2P_HEADS(CALL)                                                                         \
  if constexpr (NH == 2) { if (hd == 0) { CALL(0); } else { CALL(1); } }                          \
  else switch (hd) {                                                                             \
    case 0: CALL(0); break; case 1: CALL(1); break; case 2: CALL(2 % NH); break;                 \
    case 3: CALL(3 % NH); break; case 4: CALL(4 % NH); break; default: CALL(5 % NH); break;      \
  }

template <int D>
__global__ void __launch_bounds__(NTH, 3) wattn_bwd_pair_kernel(const PArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using CF = PC<D>;
  constexpr int ldt = CF::LDT, S = CF::S, SD = CF::SD, NBUF = CF::NBUF;
  const WinGeom g = p.g;
  int group, hg;
  {  // blockIdx -> (window group, head pair): the three pairs of a group on ONE XCD (blocks b and b + 8 share one) in
     // neighbouring dispatch slots, so the 128-byte lines their row pieces share meet in that L2; any mapping is correct
    const int b = blockIdx.x;
    if ((p.G & 7) == 0) {
      const int slot = b >> 3;
      hg = slot % NG;
      group = (slot / NG) * 8 + (b & 7);
    } else {
      hg = b % NG;
      group = b / NG;
    }
  }
  const int ch0 = hg * NH * D;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tl = wv & 1, hd = wv >> 1;
  const bool loader = wv < NLW;
  float* tabL = reinterpret_cast<float*>(smem + CF::OFF_TAB);
  constexpr float LOG2E = 1.4426950408889634f;
  const float rscale = 1.0f / p.scale;
  const int nW = g.nWh * g.nWw;
  const int nwin = g.B * nW;

  // ---- LDS-DMA plumbing (inline asm: the compiler must not see these loads, or it drains the queue at every LDS read) ----
  typedef uint32_t u32x4s_t __attribute__((ext_vector_type(4)));
  auto make_rsrc = [&](const void* ptr, uint32_t bytes) {
    u32x4s_t q;
    q.x = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)ptr);
    q.y = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)ptr >> 32) & 0xffffu);
    q.z = __builtin_amdgcn_readfirstlane(bytes);
    q.w = 0x00020000u;
    return q;
  };
  const u32x4s_t rs_qkv = make_rsrc(p.qkv, p.qkv_bytes), rs_do = make_rsrc(p.dout, p.dout_bytes);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(LDS_AS char*)smem;
  auto dma = [&](const u32x4s_t& rs, uint32_t ldst, uint32_t off) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(off), "s"(ldst), "s"(rs) : "memory");
  };
  struct WinPos { int b, wr, wc; };
  auto locate = [&](int win) {
    WinPos w;
    w.b = win / nW;
    const int wi = win - w.b * nW;
    w.wr = wi / g.nWw;
    w.wc = wi - w.wr * g.nWw;
    return w;
  };
  auto token = [&](const WinPos& w, int ri) {   // global token of window row ri (the cyclic shift is index arithmetic)
    int rr = w.wr * 8 + (ri >> 3) + g.shift;
    if (rr >= g.H) rr -= g.H;
    int cc = w.wc * 8 + (ri & 7) + g.shift;
    if (cc >= g.W) cc -= g.W;
    return (w.b * g.H + rr) * g.W + cc;     // < 2^31 tokens (checked by the caller)
  };
  const uint32_t ldb = (uint32_t)p.ld * 2u, lddb = (uint32_t)p.ldd * 2u;
  // Piece pc of an item: section pc / PPS, rows (pc % PPS) * RPP + lane / S, slot lane % S — the lanes of a row piece are
  // neighbours, so a wave instruction reads RPP contiguous pieces of 40 / 60 / 80 bytes; lanes past RPP * S (and past row
  // 63) are switched off, pad slots read out of range (zeros).  Loader wave w issues pieces w, w + NLW, ...
  auto issue = [&](const WinPos& w, int buf, int lane) {
    const int rl = lane / S, sl = lane - rl * S;
    if constexpr (CF::RPP == 4 && NH == HEADS_ALL) {
      // four rows per piece = half a window row: the row base is wave-uniform (scalar unit), a lane only adds its column —
      // ~4 vector instructions per piece instead of ~14 (the loader waves' issue code sits in front of pass T, on the path to
      // the first barrier of the window)
      int colv[2];
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        int cc = w.wc * 8 + 4 * hf + (rl & 3) + g.shift;
        if (cc >= g.W) cc -= g.W;
        colv[hf] = cc;
      }
      const bool on = rl < 4, data = sl < SD;
      const uint32_t sloff = (uint32_t)(sl * 16);
#pragma unroll
      for (int i = 0; i < CF::PPW; ++i) {
        const int pc = wv + NLW * i;                     // wave-uniform
        const int sec = pc / CF::PPS, pl = pc - sec * CF::PPS;
        const bool is_do = sec == 3;
        int rr = w.wr * 8 + (pl >> 1) + g.shift;         // window row pl / 2 (scalar)
        if (rr >= g.H) rr -= g.H;
        const int rb = (w.b * g.H + rr) * g.W;
        const uint32_t t = (uint32_t)(rb + colv[pl & 1]);
        const uint32_t off = data ? t * (is_do ? lddb : ldb) + (uint32_t)((is_do ? 0 : sec * CF::CFULL) * 2) + sloff : 0xffffffffu;
        if (on)
          dma(is_do ? rs_do : rs_qkv, __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(buf * CF::BUFB + sec * CF::SECB + pl * 4 * ldt)), off);
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < CF::PPW; ++i) {
      const int pc = wv + NLW * i;                     // wave-uniform
      const int sec = pc / CF::PPS, pl = pc - sec * CF::PPS;
      const int row = pl * CF::RPP + rl;
      const bool is_do = sec == 3;
      const uint32_t t = (uint32_t)token(w, row & 63);
      const uint32_t off = sl < SD ? t * (is_do ? lddb : ldb) + (uint32_t)(((is_do ? 0 : sec * CF::CFULL) + ch0) * 2 + sl * 16) : 0xffffffffu;
      if (rl < CF::RPP && row < 64)
        dma(is_do ? rs_do : rs_qkv, __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(buf * CF::BUFB + sec * CF::SECB + pl * CF::RPP * ldt)), off);
    }
  };

  int win = group;
  WinPos ahead[NBUF - 1];   // positions of the items in flight (item i + 1 .. i + NBUF - 1 while item i is computed)
  // ---- the first NBUF - 1 items go in flight ----
#pragma unroll
  for (int a = 0; a < NBUF - 1; ++a) {
    const int wa = win + a * p.G;
    ahead[a] = locate(wa < nwin ? wa : 0);
    if (loader && wa < nwin) issue(ahead[a], a, lane);
  }
  // ---- one-time LDS state ----
  {  // the last buffer of the ring and the tail behind it: finite before their first use (k-steps that run past a row read them)
    for (int i = tid * 16; i < CF::BUFB + 64; i += NTH * 16) *reinterpret_cast<float4*>(smem + (NBUF - 1) * CF::BUFB + i) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  {  // relative-position table / scale in natural order, two copies (the second shifted by a column) so that a lane's
     // pairs are aligned 8-byte reads whatever the parity of its window column
    constexpr int NT_SRC = 225 * NH, NLD = (NT_SRC + NTH - 1) / NTH;
    float tv[NLD];
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int j0 = tid + NTH * k, j = j0 < NT_SRC ? j0 : NT_SRC - 1;
      tv[k] = p.table[(j / NH) * HEADS_ALL + hg * NH + (j % NH)];
    }
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int j = tid + NTH * k;
      if (j < NT_SRC) {
        const int rel = j / NH, hh = j - rel * NH;
        const int dy = rel / 15, dxr = rel - dy * 15;
        const float v = tv[k] * rscale;
        tabL[(hh * 15 + dy) * TSX + dxr] = v;
        if (dxr >= 1) tabL[CF::TABC + (hh * 15 + dy) * TSX + dxr - 1] = v;
      }
    }
  }
  if (tid < 64) {  // 0/1 operand of the d(table) MFMA: A[m][8h + jj] = 1 where m is the key of pack element jj
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int j0 = 2 * e, j1 = 2 * e + 1;
        const int k0 = 16 * s + 8 * (j0 >> 2) + 4 * h + (j0 & 3), k1 = 16 * s + 8 * (j1 >> 2) + 4 * h + (j1 & 3);
        *reinterpret_cast<uint32_t*>(smem + CF::OFF_ID + lane * 32 + 16 * s + 4 * e) =
            (r == k0 ? 0x00003f80u : 0u) | (r == k1 ? 0x3f800000u : 0u);
      }
  }
  Ctx c;
  c.thr = g.ws - g.shift;
  c.cbits = (uint32_t)__builtin_bit_cast(uint16_t, (__bf16)(100.0f * rscale));
  c.scale2 = p.scale * LOG2E;
  c.scale = p.scale;
  // the lane-dependent LDS positions are re-derived per item from an opaque copy of the thread index (built once in
  // front of the loop they are a dozen registers live across it)
  auto lane_ctx = [&](int tidx, int buf) {
    const int lane = tidx & 63, r = lane & 31, h = lane >> 5;
    const int yi = tl * 4 + (r >> 3), xi = r & 7;
    const int gq = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const int tok = tl * 32 + r;
    lds_cp base = (lds_cp)(smem + buf * CF::BUFB);
    c.h = h;
    c.own = base + tok * ldt;
    c.rowA = base + r * ldt + h * 16;
    c.tr = base + (4 * h + q) * ldt + (16 * (gq & 1) + 4 * pp) * 2;
    c.idp = (lds_cp)(smem + CF::OFF_ID + lane * 32);
    {
      const int a0 = xi + 4 - 4 * h;     // pass T: lowest natural column of the lane's four (xi + 7 - 4 h - 3), key row 0
      const float* tb = (a0 & 1) ? tabL + CF::TABC + yi * TSX + (a0 - 1) : tabL + yi * TSX + a0;
      c.tb = (const LDS_AS f32x2*)tb;
      const int n0 = 4 * h - xi + 7;     // pass N: natural column for query column 4 h of key column xi
      const float* tn = (n0 & 1) ? tabL + CF::TABC + (7 - yi) * TSX + (n0 - 1) : tabL + (7 - yi) * TSX + n0;
      c.tn = (const LDS_AS f32x2*)tn;
    }
    c.st = (const LDS_AS float*)(smem + CF::OFF_ST) + hd * 64;
    c.yi = yi;
    c.xi = xi;
  };
  f32x16 Dsum[2];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int v = 0; v < 16; ++v) Dsum[kt][v] = 0.f;

  // every wave: the table loads are done; loader waves: item 0 has landed, items 1 .. NBUF - 2 stay in flight
  if (NBUF > 2 && loader) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CF::PPW * (NBUF - 2)) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  int buf = 0;
  for (; win < nwin; win += p.G) {
    __syncthreads();  // b0: this item's rows have landed (the loader waves waited for them); the buffer of the previous item is free
    const WinPos w = ahead[0];
#pragma unroll
    for (int a = 0; a + 1 < NBUF - 1; ++a) ahead[a] = ahead[a + 1];
    int tidw = tid;
    asm volatile("" : "+v"(tidw));
    {
      const int nxt = win + (NBUF - 1) * p.G;
      const int nb = buf == 0 ? NBUF - 1 : buf - 1;           // the buffer item i - 1 just left
      ahead[NBUF - 2] = locate(nxt < nwin ? nxt : 0);
      if (loader && nxt < nwin && !(K2P_ABL & 16)) issue(ahead[NBUF - 2], nb, tidw & 63);   // in flight for NBUF - 1 items
    }
    lane_ctx(tidw, buf);
    c.mrow = g.shift > 0 && w.wr == g.nWh - 1;
    c.mcol = g.shift > 0 && w.wc == g.nWw - 1;
    c.masked = __builtin_amdgcn_readfirstlane((int)(c.mrow || c.mcol)) != 0;
    f32x16 dq, dv, dk;
#if K2P_ABL & 1
#pragma unroll
    for (int v = 0; v < 16; ++v) dq[v] = 0.f;
#else
#define K2P_T(HD) pass_t<D, HD>(c, dq, Dsum)
    K2P_HEADS(K2P_T)
#undef K2P_T
#endif
    __syncthreads();  // b1: the statistics of both query tiles are in LDS
#if K2P_ABL & 2
#pragma unroll
    for (int v = 0; v < 16; ++v) { dv[v] = 0.f; dk[v] = 0.f; }
#else
#define K2P_N(HD) pass_n<D, HD>(c, dv, dk)
    K2P_HEADS(K2P_N)
#undef K2P_N
#endif
    __syncthreads();  // b2: nobody reads the sections any more: the gradient tiles go where Q / K / V were
#if !(K2P_ABL & 4)
#define K2P_ST(HD)                                                                               \
    {                                                                                            \
      store_tile_rows<CF::og(HD), HD * D, HD * D + D>(c.own + CF::OFF_Q, dq, 1.0f, c.h);         \
      store_tile_rows<CF::og(HD), HD * D, HD * D + D>(c.own + CF::OFF_K, dk, c.scale, c.h);      \
      store_tile_rows<CF::og(HD), HD * D, HD * D + D>(c.own + CF::OFF_V, dv, 1.0f, c.h);         \
    }
    K2P_HEADS(K2P_ST)
#undef K2P_ST
#endif
    __syncthreads();  // b3
    if (!loader) {
#if !(K2P_ABL & 8)
      const char* sb = smem + buf * CF::BUFB;
      int lane2 = tid;
      asm volatile("" : "+v"(lane2));
      lane2 &= 63;
      if constexpr (NH == 6) {
        // the storer waves, whole rows: a store instruction writes ONE dqkv row (360 / 540 / 720 contiguous bytes), lane =
        // 16-byte chunk of it; the chunk's four dwords come from the Q / K / V sections (a third of 120 / 180 bytes does not
        // end on a chunk boundary).  Plain stores (the compiler keeps the store-data hazard distance); nothing waits for them.
        const int b0 = lane2 * 16;
        int lo[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int b = b0 + 4 * i, sec = (b >= CF::ROWB ? 1 : 0) + (b >= 2 * CF::ROWB ? 1 : 0);
          lo[i] = sec * CF::SECB + (b - sec * CF::ROWB);
        }
        const int nb = CF::ROW3B - b0;          // bytes of the row from this chunk on: >= 16 full chunk, 8 / 12 tail, <= 0 idle
#pragma unroll
        for (int rr = 0; rr < CF::RPS; ++rr) {
          const int row = (wv - NLW) * CF::RPS + rr;                                 // wave-uniform
          const uint32_t t = (uint32_t)token(w, row);
          char* dst = reinterpret_cast<char*>(p.dqkv) + (size_t)t * ((size_t)p.ldq * 2) + (size_t)b0;
          const char* src = sb + row * ldt;
          u32x4_t v;
          if constexpr (CF::ROWB % 16 == 0) {
            v = *reinterpret_cast<const u32x4_t*>(src + lo[0]);
          } else {
            v.x = *reinterpret_cast<const uint32_t*>(src + lo[0]);
            v.y = *reinterpret_cast<const uint32_t*>(src + lo[1]);
            v.z = *reinterpret_cast<const uint32_t*>(src + lo[2]);
            v.w = *reinterpret_cast<const uint32_t*>(src + lo[3]);
          }
          if (nb >= 16) {
            *reinterpret_cast<u32x4_t*>(dst) = v;
          } else if (nb == 8) {
            u32x2_t v2;
            v2.x = v.x; v2.y = v.y;
            *reinterpret_cast<u32x2_t*>(dst) = v2;
          } else if (nb == 12) {
            u32x3_a4 v3;
            v3.x = v.x; v3.y = v.y; v3.z = v.z;
            *reinterpret_cast<u32x3_a4*>(dst) = v3;
          }
        }
      } else {
      // the storer waves: rows RPS (wv - NLW) .. of the three gradient sections, chunk cid = (row, section, slot) with the
      // slots of a row piece on neighbouring lanes.  Plain stores (not inline asm: the compiler has to see them to keep
      // the store-data hazard distance); nothing ever waits for them.
#pragma unroll
      for (int k = 0; k < CF::NPASS; ++k) {
        const int cid = 64 * k + lane2;
        const int rowl = cid / (3 * SD), rem = cid - rowl * (3 * SD), sec = rem / SD, sl = rem - sec * SD;
        const int row = (wv - NLW) * CF::RPS + rowl;
        const bool act = cid < CF::NCH;
        const uint32_t t = (uint32_t)token(w, act ? row : 0);
        char* dst = reinterpret_cast<char*>(p.dqkv) + (size_t)t * ((size_t)p.ldq * 2) + (size_t)((sec * CF::CFULL + ch0) * 2 + sl * 16);
        const char* src = sb + sec * CF::SECB + row * ldt + sl * 16;
        if (act) {
          if (CF::TAILB == 16 || sl < SD - 1) {
            *reinterpret_cast<u32x4_t*>(dst) = *reinterpret_cast<const u32x4_t*>(src);
          } else if (CF::TAILB == 8) {
            *reinterpret_cast<u32x2_t*>(dst) = *reinterpret_cast<const u32x2_t*>(src);
          } else {
            const u32x4_t v4 = *reinterpret_cast<const u32x4_t*>(src);
            u32x3_a4 v;
            v.x = v4.x; v.y = v4.y; v.z = v4.z;
            *reinterpret_cast<u32x3_a4*>(dst) = v;
          }
        }
      }
      }
#endif
    } else {
      // the loader waves: the next item's rows have landed; the NBUF - 2 younger items stay in flight.  These waves have
      // no store outstanding (completion is reported in issue order: a load behind a store would wait for the store)
      if (NBUF > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CF::PPW * (NBUF - 2)) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    buf = buf == NBUF - 1 ? 0 : buf + 1;
  }
  // d(table): the per-lane partial sums of all items of this workgroup -> slab row [group][6][225], heads of this pair.
  // Laid out as a dense [head][query][key] matrix in LDS; each of the 15 x 15 relative positions adds up its diagonal
  // in a fixed order (deterministic).
  __syncthreads();
  float* ds = reinterpret_cast<float*>(smem);
  {
    const int r = lane & 31, h = lane >> 5;
    const int yi = tl * 4 + (r >> 3), xi = r & 7;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int yj = kt * 4 + (v >> 2), xj = (v & 3) + 4 * h;
        ds[(hd * 64 + yi * 8 + xi) * 65 + yj * 8 + xj] = Dsum[kt][v] * rscale;   // (pass T accumulated scale * dS)
      }
  }
  __syncthreads();
  float* my = p.slab + ((int64_t)group * HEADS_ALL + hg * NH) * 225;
  for (int e = tid; e < NH * 225; e += NTH) {
    const int hd2 = e / 225, rem = e - hd2 * 225, dy = rem / 15 - 7, dx = rem - (rem / 15) * 15 - 7;
    const float* base = ds + hd2 * 64 * 65;
    float a = 0.f;
#pragma unroll
    for (int qy = 0; qy < 8; ++qy) {
      float t[8];
#pragma unroll
      for (int qx = 0; qx < 8; ++qx) {
        const int ky = qy - dy, kx = qx - dx;
        const bool ok = (unsigned)ky < 8u && (unsigned)kx < 8u;
        t[qx] = base[ok ? (qy * 8 + qx) * 65 + ky * 8 + kx : 0];
        t[qx] = ok ? t[qx] : 0.f;
      }
#pragma unroll
      for (int qx = 0; qx < 8; ++qx) a += t[qx];
    }
    my[e] = a;
  }
}

template <int D>
int launch_pair(const PArgs& p0, int slab_rows, int* nslab, hipStream_t st) {
  using CF = PC<D>;
  auto kern = wattn_bwd_pair_kernel<D>;
  const int64_t nwin = (int64_t)p0.g.B * p0.g.nWh * p0.g.nWw;
  int64_t G = 256;      // window groups x 3 pairs = 768 workgroups = three per CU
  if (G > nwin) G = nwin;
  if (G > slab_rows) G = slab_rows;
  *nslab = (int)G;
  PArgs p = p0;
  p.G = (int)G;
  if (CF::SMEM > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CF::SMEM);
  hipLaunchKernelGGL(kern, dim3((unsigned)(G * NG)), dim3(NTH), CF::SMEM, st, p);
  return rdst_launch_status("wattn_bwd_pair");
}

}  // namespace

// bf16, ws 8, 6 heads of dim 10 / 15 / 20, no explicit mask, scale > 0, DENSE rows (ld_qkv = ld_dqkv = 3C, ld_dout = C) of
// at most 2^31 bytes; RDST_ENOTSUP otherwise (the caller falls back to wattn_bwd_mfma_hd.hip).
// slab: [slab_rows >= 1][6][225] floats; *nslab = rows written (summed by the caller)
int wattn_bwd_pair(const void* qkv, int64_t ld, const float* table, const void* dout, int64_t ldd, void* dqkv, int64_t ldq,
                   float* slab, int slab_rows, const WinGeom& g, float scale, int* nslab, hipStream_t st) {
  if (g.ws != 8 || g.heads != 6 || g.mask || !(scale > 0.f) || g.C % 6 || slab_rows < 1) return RDST_ENOTSUP;
  if (ld != 3 * g.C || ldq != 3 * g.C || ldd != g.C) return RDST_ENOTSUP;
  const int64_t ntok = (int64_t)g.B * g.H * g.W;
  if (ntok * 3 * g.C * 2 >= (1ll << 31)) return RDST_ENOTSUP;
  static int off = -1;
  if (off < 0) {
    const char* e = rdst_dbg_getenv("RDST_K2_PAIR");
    off = (e && e[0] == '0') ? 1 : 0;
  }
  if (off) return RDST_ENOTSUP;
  const int d = g.C / 6;
  const int gran = d == 10 ? 8 : d == 15 ? 4 : 16;
  if ((uintptr_t)qkv % gran || (uintptr_t)dout % gran || (uintptr_t)dqkv % gran) return RDST_ENOTSUP;
  PArgs p{};
  p.qkv = (const bf16*)qkv; p.dout = (const bf16*)dout; p.dqkv = (bf16*)dqkv; p.table = table; p.slab = slab;
  p.ld = ld; p.ldd = ldd; p.ldq = ldq; p.g = g; p.scale = scale;
  p.qkv_bytes = (uint32_t)(ntok * 3 * g.C * 2);
  p.dout_bytes = (uint32_t)(ntok * g.C * 2);
  if (d == 10) return launch_pair<10>(p, slab_rows, nslab, st);
  if (d == 15) return launch_pair<15>(p, slab_rows, nslab, st);
  if (d == 20) return launch_pair<20>(p, slab_rows, nslab, st);
  return RDST_ENOTSUP;
}
