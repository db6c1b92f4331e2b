// K2 (window attention backward) specialised at compile time like the forward (wattn_mfma_hd.hip):
// bf16, 8x8 windows, HEADS = 6 heads of dim D (C = 60 / 90 / 120).
//
// Work item = (window, group of NH heads); a workgroup of 2 NH waves, wave w = (tile t = w & 1, local head hd = w >> 1),
// each wave's head a compile-time constant.  NH = 2 (round 4): a 4-wave workgroup owns one head PAIR of a window — 40 / 60 /
// 80 bytes of every q / k / v / dOut row — needs 49-53 KB of LDS, and THREE of them share a CU.  They run independently:
// while one waits for its rows or drains its gradient rows, the other two compute.  (NH = 6, rounds 1-3: one 12-wave
// workgroup per CU did a whole window behind six barriers, the no-compute skeleton of that kernel alone took 46-62 us of
// its 73-82: one window of loads in flight per CU, issued less than a load latency ahead.)  The three pair workgroups of
// a window walk the same windows in the same order on the same XCD (blockIdx -> (group, pair) below), so the 128-byte
// lines their row pieces share meet in that XCD's L2.
//   phase A  (t = query tile; keys in the accumulator registers, query on the lane, as in the forward)
//     S^T = K.Q^T (+ bias/scale as initial accumulator) -> softmax ONCE -> P^T;  dP^T = V.dO^T;
//     delta = rowsum(P dP);  dS^T = P^T (dP^T - delta)
//     dQ^T  = K^T . dS^T            (accumulator-as-operand, K read transposed)
//     d(table) += dS^T  as an MFMA with a constant 0/1 operand (I . dS^T): the 32 per-lane partial sums
//       per head never touch the vector ALUs and live in accumulator registers for the whole kernel
//   P and dS go to LDS once (bf16, [query][key], 8-B chunks XOR-swizzled by the row), which is the
//   transposition the other two products need — the softmax is NOT recomputed in a second orientation:
//   phase B  (t = key tile; key on the lane)
//     dV^T = dO^T . P,   dK^T = Q^T . dS   (both operands read transposed from LDS)
// The P/dS images overlay the K and V sections, which are dead once phase A is done; dQ/dK/dV tiles are
// written over the Q/K/V sections after phase B and leave as full dqkv rows.  The next window's rows
// (qkv + dOut) are prefetched into registers while the current window is computed.
// HBM traffic = the algorithmic 7*C*elt bytes per token (+ table, + one d(table) slab row per workgroup).
#include "wattn_hd.h"
#include <stdlib.h>

namespace {
using namespace wahd;

constexpr int HEADS_ALL = 6;     // heads of the layer (the kernel covers 6 heads of dim 10 / 15 / 20)
constexpr int MAXR2 = 6;         // row chunks staged per lane (12 rows per pass of the workgroup, 64 rows)

struct BwArgs {
  const bf16* qkv; int64_t ld;
  const float* table;
  const bf16* dout; int64_t ldd;
  bf16* dqkv; int64_t ldq;
  float* slab;
  int G;                       // window groups: workgroup (group, head group) walks windows group, group + G, ...
  WinGeom g;
  float scale;
  unsigned long long* stamps;  // RDST_K2_STAMPS=n: [grid][16] s_memtime stamps of thread 0 (debug only), else NULL
};

template <int D, int HEADS>
struct HdB {
  static constexpr int C = D * HEADS;
  static constexpr int SEC = C * 2;
  static constexpr int LDT0 = ((SEC + 31) / 32) * 32;
  // odd number of 16-B slots (b128 row reads); a 256-B row gets 80 more, not 16: at 272 the four rows of a transposed
  // read (ds_read_b64_tr_b16: 4 rows x 64 B per 16 lanes) would sit 16 B apart on the same banks (4-way conflicts)
  static constexpr int LDT = LDT0 % 256 == 0 ? LDT0 + 80 : (LDT0 / 16) % 2 == 0 ? LDT0 + 16 : LDT0;
  static constexpr int PROW = 128;                                     // bytes per P / dS row (64 keys, swizzled)
  static constexpr int PMAT = 64 * PROW;
  static constexpr int OFF_Q = 0, OFF_DO = 64 * LDT, OFF_R = 2 * 64 * LDT;
  static constexpr int OFF_K = OFF_R, OFF_V = OFF_R + 64 * LDT;        // dead after phase A ...
  static constexpr int RSIZE = 2 * 64 * LDT > HEADS * 2 * PMAT ? 2 * 64 * LDT : HEADS * 2 * PMAT;  // ... P/dS overlay them
  static constexpr int OFF_TAB = OFF_R + RSIZE;
  static constexpr int TABF = HEADS * 15 * TSX, TABB = TABF + 8;
  static constexpr int OFF_ID = OFF_TAB + (TABB + TABF + 8) * 4;       // [64 lanes][2 x 16 B] 0/1 operand packs
  // the d(table) epilogue lays all HEADS x 64 x 64 partial sums out in LDS (row stride 65 floats: conflict-free)
  static constexpr size_t SMEM_EPI = (size_t)HEADS * 64 * 65 * 4;
  static constexpr size_t SMEM = (size_t)OFF_ID + 64 * 32 > SMEM_EPI ? (size_t)OFF_ID + 64 * 32 : SMEM_EPI;
  static constexpr int og(int hd) { return (hd * D) & ~3; }             // first channel row of a head's output tile
};

struct BwCtx {
  // per-lane LDS bases; everything else is a compile-time offset from one of these
  lds_cp rowT;    // Q-section row of the lane's token (query in phase A, key in phase B), no lane-half offset
  lds_cp rowK;    // K-section row (lane & 31) + 16 B * lane half: A-operand packs of K (and V at +64*ldt)
  lds_cp trK;     // K section, transposed-read position for key-major k-steps (4h + q, column quad)
  lds_cp trQ;     // Q section, transposed-read position for query-major k-steps (8h + q, column quad)
  lds_cp PWp, PRp0, PRp1;   // swizzled P/dS image: the lane's write row, its two read rows
  lds_cp idp;     // the lane's 0/1 operand packs of the d(table) MFMA (2 x 16 B)
  const LDS_AS f32x2* tb;
  int h;
  bool masked, mrow, mcol;
  int thr;
  float scale2, scale;
  int yi, xi;        // the lane's query row / column inside the window (key row of tile kt: 4 kt + (yi & 3))
  uint32_t cbits;    // bf16(100 / scale)
};

// phase A of one (query tile, head): leaves P^T / dS^T packs in pP / pdS and the dQ^T tile in dq
template <int D, int HEADS, int HD>
__device__ __forceinline__ void bw_phase_a(const BwCtx& c, Pack16 (&pP)[2][2], Pack16 (&pdS)[2][2], f32x16& dq,
                                           f32x16 (&Dsum)[2]) {
  using CF = HdB<D, HEADS>;
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
      const f32x2 b2 = lds_read_f32x2(tbh + ((7 - (kt * 4 + (v >> 2))) * TSX + (v & 3)) / 2);
      X[kt][v] = b2.x;
      X[kt][v + 1] = b2.y;
      Y[kt][v] = 0.f;
      Y[kt][v + 1] = 0.f;
    }
#pragma unroll
  for (int t = t_lo; t <= t_hi; ++t) {
    Pack16 qb = lds_pack(c.rowT + h * 16 + t * 32), gb = lds_pack(c.rowT + h * 16 + (CF::OFF_DO - CF::OFF_Q) + t * 32);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t mA = qmask_bits(t * 16 + 2 * e, c_lo, c_hi), mB = qmask_bits(t * 16 + 8 + 2 * e, c_lo, c_hi);
      const uint32_t m = h ? mB : mA;
      qb.w[e] &= m;
      gb.w[e] &= m;
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      const Pack16 ka = lds_pack(c.rowK + kt * 32 * ldt + t * 32);
      const Pack16 va = lds_pack(c.rowK + (CF::OFF_V - CF::OFF_K) + kt * 32 * ldt + t * 32);
      Mma<bf16>::mma(X[kt], ka, qb);   // S'^T  = K . Q^T   (+ bias / scale)
      Mma<bf16>::mma(Y[kt], va, gb);   // dP^T  = V . dO^T
    }
  }
  if (c.masked) {
    // shifted-window mask as one more k-step (see wattn_mfma_hd.hip): +100/scale where the regions of query and
    // key AGREE is the reference's -100 where they differ up to a per-row constant, which softmax ignores
    // (the one-hot packs are rebuilt here, a dozen vector instructions, instead of living in registers)
    auto onehot = [&](int reg, uint32_t v) {   // elements 0..3 of lane half 0 (k = 0..3 of the k-step)
      Pack16 q;
      q.w[0] = h ? 0u : ((reg == 0 ? v : 0u) | (reg == 1 ? v << 16 : 0u));
      q.w[1] = h ? 0u : ((reg == 2 ? v : 0u) | (reg == 3 ? v << 16 : 0u));
      q.w[2] = 0u;
      q.w[3] = 0u;
      return q;
    };
    const int rx = (c.mcol && c.xi >= c.thr) ? 1 : 0;   // xi = lane & 7 is also the key column of row r of a key tile
    const Pack16 mQ = onehot(2 * ((c.mrow && c.yi >= c.thr) ? 1 : 0) + rx, c.cbits);
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
      Mma<bf16>::mma(X[kt], onehot(2 * ((c.mrow && kt * 4 + (c.yi & 3) >= c.thr) ? 1 : 0) + rx, 0x3f80u), mQ);
  }
#if defined(K2_ABL) && (K2_ABL & 16)
  // ablation (r05 verdict item 2a, upper bound): what phase A would cost with the forward's log-sum-exp and delta = rowsum(dO o O) handed
  // in per (token, head) — no row maximum, no row sum, no normalisation, no delta reduction (numerically meaningless: timing only)
  {
    const float nl = -c.scale2 * X[0][0];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(X[kt][v], c.scale2, nl));
        Y[kt][v] = pv * (Y[kt][v] - 0.125f);
        X[kt][v] = pv;
      }
  }
#else
  float m = X[0][0];
#pragma unroll
  for (int v = 1; v < 16; ++v) m = __builtin_fmaxf(m, X[0][v]);
#pragma unroll
  for (int v = 0; v < 16; ++v) m = __builtin_fmaxf(m, X[1][v]);
  m = half_swap_max(m);
  const float nm = -c.scale2 * m;
  float l0 = 0.f, l1 = 0.f;
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int v = 0; v < 16; v += 2) {
      const float e0 = __builtin_amdgcn_exp2f(__builtin_fmaf(X[kt][v], c.scale2, nm));
      const float e1 = __builtin_amdgcn_exp2f(__builtin_fmaf(X[kt][v + 1], c.scale2, nm));
      l0 += e0;
      l1 += e1;
      X[kt][v] = e0;
      X[kt][v + 1] = e1;
    }
  const float inv = __builtin_amdgcn_rcpf(half_swap_sum(l0 + l1));
  float d0 = 0.f, d1 = 0.f;
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int v = 0; v < 16; v += 2) {
      X[kt][v] *= inv;                                  // P
      X[kt][v + 1] *= inv;
      d0 = __builtin_fmaf(X[kt][v], Y[kt][v], d0);      // delta = sum_j P dP
      d1 = __builtin_fmaf(X[kt][v + 1], Y[kt][v + 1], d1);
    }
  const float delta = half_swap_sum(d0 + d1);
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int v = 0; v < 16; ++v) Y[kt][v] = X[kt][v] * (Y[kt][v] - delta);   // dS (gradient of the logits)
#endif
  // element jj of lane half h of pack (kt, s) is key kt*32 + 16s + 8(jj>>2) + 4h + (jj&3)
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        pP[kt][s].w[e] = pack_bf16x2(X[kt][8 * s + 2 * e], X[kt][8 * s + 2 * e + 1]);
        pdS[kt][s].w[e] = pack_bf16x2(Y[kt][8 * s + 2 * e], Y[kt][8 * s + 2 * e + 1]);
      }
  // dQ^T (rows = channels OG .. OG+31, cols = queries) = K^T . dS^T;  d(table) partial sums += I . dS^T
#pragma unroll
  for (int v = 0; v < 16; ++v) dq[v] = 0.f;
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const lds_cp kb = c.trK + OG * 2 + (kt * 32 + 16 * s) * ldt;
      const Pack16 ka = lds_tr_pack(kb, kb + 8 * ldt);
      Mma<bf16>::mma(dq, ka, pdS[kt][s]);
      Mma<bf16>::mma(Dsum[kt], lds_pack(c.idp + 16 * s), pdS[kt][s]);
    }
}

// P^T / dS^T packs -> the head's [query][key] bf16 images (8-B chunks XOR-swizzled by the row)
template <int D, int HEADS>
__device__ __forceinline__ void bw_store_p(const BwCtx& c, const Pack16 (&pP)[2][2], const Pack16 (&pdS)[2][2]) {
  using CF = HdB<D, HEADS>;
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int g2 = 0; g2 < 2; ++g2) {
        const int A8 = (kt * 8 + 4 * s + 2 * g2) * 8;   // chunk index (without the lane-half bit) * 8 bytes
        const lds_cp dst = (lds_cp)((uint32_t)(uintptr_t)c.PWp ^ (uint32_t)A8);
        u32x2_t w;
        w.x = pP[kt][s].w[2 * g2]; w.y = pP[kt][s].w[2 * g2 + 1];
        *reinterpret_cast<LDS_AS u32x2_t*>(dst) = w;
        w.x = pdS[kt][s].w[2 * g2]; w.y = pdS[kt][s].w[2 * g2 + 1];
        *reinterpret_cast<LDS_AS u32x2_t*>(dst + CF::PMAT) = w;
      }
}

// phase B of one (key tile, head): dV^T = dO^T . P,  dK^T = Q^T . dS   (rows = channels OG .. OG+31, cols = keys)
template <int D, int HEADS, int HD>
__device__ __forceinline__ void bw_phase_b(const BwCtx& c, f32x16& dv, f32x16& dk) {
  using CF = HdB<D, HEADS>;
  constexpr int ldt = CF::LDT;
  constexpr int OG = CF::og(HD);
#pragma unroll
  for (int v = 0; v < 16; ++v) {
    dv[v] = 0.f;
    dk[v] = 0.f;
  }
#pragma unroll
  for (int s = 0; s < 4; ++s) {  // k-steps over the 64 queries: rows 16s + 8h + (0..7)
    const Pack16 pb = lds_tr_pack(c.PRp0 + s * 16 * CF::PROW, c.PRp1 + s * 16 * CF::PROW);
    const Pack16 sb = lds_tr_pack(c.PRp0 + CF::PMAT + s * 16 * CF::PROW, c.PRp1 + CF::PMAT + s * 16 * CF::PROW);
    const lds_cp ga = c.trQ + (CF::OFF_DO - CF::OFF_Q) + OG * 2 + s * 16 * ldt;
    const lds_cp qa = c.trQ + OG * 2 + s * 16 * ldt;
    const Pack16 gA = lds_tr_pack(ga, ga + 4 * ldt);
    const Pack16 qA = lds_tr_pack(qa, qa + 4 * ldt);
    Mma<bf16>::mma(dv, gA, pb);
    Mma<bf16>::mma(dk, qA, sb);
  }
}

template <int D, int HEADS, int GRAN>
__global__ void __launch_bounds__(128 * HEADS, 3) wattn_bwd_hd_kernel(const BwArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using CF = HdB<D, HEADS>;
  using CH = typename Chunk<GRAN>::type;
  constexpr int NW2 = 2 * HEADS, NT2 = 64 * NW2;      // waves / threads per workgroup
  constexpr int NG = HEADS_ALL / HEADS;               // head groups per window
  constexpr int CFULL = HEADS_ALL * D;                // channels of a q / k / v third of a qkv row
  constexpr int ldt = CF::LDT, secb = CF::SEC;
  constexpr int cps = secb / GRAN;  // chunks per section
  constexpr int LPR = 4 * cps;      // lanes that move one token row (q, k, v, dOut pieces)
  constexpr int RPI = 64 / LPR;     // token rows per wave instruction
  static_assert(LPR <= 64 && NW2 * RPI * MAXR2 >= 64, "copy plan");
  const WinGeom g = p.g;
  // blockIdx -> (window group, head group): the NG workgroups of a group sit on ONE XCD (blocks b and b + 8 share one)
  // in neighbouring dispatch slots; placement is for L2 locality only, any mapping is correct
  int group, hg;
  {
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
  const int ch0 = hg * HEADS * D;   // first channel of this workgroup's heads inside each third / inside a dOut row
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tl = wv & 1, hd = wv >> 1;   // tile (queries in phase A, keys in phase B) and head of this wave
  float* tabL = reinterpret_cast<float*>(smem + CF::OFF_TAB);
  constexpr float LOG2E = 1.4426950408889634f;
  const float rscale = 1.0f / p.scale;
  const int nW = g.nWh * g.nWw;
  const int nwin = g.B * nW;
  const int yi = tl * 4 + (r >> 3), xi = r & 7;
  const int thr = g.ws - g.shift;
  const int gq = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  const int tok = tl * 32 + r;   // the lane's query (phase A) / key (phase B) row

  int nst = 0;
  auto stamp = [&]() {
    if (RDST_DBGV(p.stamps) && tid == 0 && nst < 14) p.stamps[(size_t)blockIdx.x * 16 + nst++] = __builtin_readcyclecounter();
  };
  stamp();  // 0
  BwCtx c;
  // the lane-dependent LDS positions of the context, re-derived per window from an opaque copy of the thread index: built
  // once in front of the window loop they are a dozen registers live across it, and at 168 registers per wave D = 15 / 20
  // spilled four (none now: 163 / 168 / 161 registers) — which is what kept D = 15 from fetching the next window early
  auto lane_ctx = [&](int tidx) {
    const int lane = tidx & 63, r = lane & 31, h = lane >> 5;
    const int yi = tl * 4 + (r >> 3), xi = r & 7;
    const int gq = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const int tok = tl * 32 + r;
  c.h = h;
    c.rowT = (lds_cp)(smem + CF::OFF_Q + tok * ldt);
    c.rowK = (lds_cp)(smem + CF::OFF_K + r * ldt + h * 16);
    c.trK = (lds_cp)(smem + CF::OFF_K + (4 * h + q) * ldt + (16 * (gq & 1) + 4 * pp) * 2);
    c.trQ = (lds_cp)(smem + CF::OFF_Q + (8 * h + q) * ldt + (16 * (gq & 1) + 4 * pp) * 2);
    {
      const int pbase = CF::OFF_R + hd * 2 * CF::PMAT;
      c.PWp = (lds_cp)(smem + ((pbase + tok * CF::PROW) ^ ((((r & 15) ^ h)) * 8)));
      const int row0 = 8 * h + q, row1 = row0 + 4, chunk = tl * 8 + 4 * (gq & 1) + pp;
      c.PRp0 = (lds_cp)(smem + ((pbase + row0 * CF::PROW) ^ ((chunk ^ (row0 & 15)) * 8)));
      c.PRp1 = (lds_cp)(smem + ((pbase + row1 * CF::PROW) ^ ((chunk ^ (row1 & 15)) * 8)));
    }
    c.idp = (lds_cp)(smem + CF::OFF_ID + lane * 32);
    {
      const int u0 = 4 * h - xi + 7;
      const float* tb = (u0 & 1) ? tabL + CF::TABB + yi * TSX + (u0 - 1) : tabL + yi * TSX + u0;
      c.tb = (const LDS_AS f32x2*)tb;
    }
    c.yi = yi;
    c.xi = xi;
  };
  lane_ctx(tid);
  c.thr = thr;
  c.cbits = (uint32_t)__builtin_bit_cast(uint16_t, (__bf16)(100.0f * rscale));
  c.scale2 = p.scale * LOG2E;
  c.scale = p.scale;
  if (tid < 64) {  // 0/1 operand of the d(table) MFMA: A[m][8h + jj] = 1 where m is the key of pack element jj
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
  // copy plan: a wave instruction moves RPI token rows (LPR lanes each: 3 cps lanes the q / k / v chunks, cps lanes the
  // dOut chunks); pass k of the workgroup covers rows NW2 * RPI * k + wave * RPI + rsub
  const int rsub = lane / LPR, l2 = lane - rsub * LPR;
  const int sec = (l2 >= cps) + (l2 >= 2 * cps) + (l2 >= 3 * cps);   // 0 Q, 1 K, 2 V, 3 dOut
  const int chk = l2 - sec * cps;
  const int sec_off = sec == 0 ? CF::OFF_Q : sec == 1 ? CF::OFF_K : sec == 2 ? CF::OFF_V : CF::OFF_DO;
  char* my_lds = smem + sec_off + chk * GRAN;
  const bool ld_act = rsub < RPI, st_act = ld_act && sec < 3;
  const int row0 = wv * RPI + (ld_act ? rsub : 0);
  // byte offset of the lane's chunk inside a qkv / dOut / dqkv row
  const int goff = ((sec < 3 ? sec * CFULL : 0) + ch0) * 2 + (ld_act ? chk : 0) * GRAN;

  struct WinPos { int b, wr, wc; };
  auto locate = [&](int win) {
    WinPos w;
    w.b = win / nW;
    const int wi = win - w.b * nW;
    w.wr = wi / g.nWw;
    w.wc = wi - w.wr * g.nWw;
    return w;
  };
  auto token = [&](const WinPos& w, int ri) {  // global token row of window row ri
    int rr = w.wr * 8 + (ri >> 3) + g.shift;
    if (rr >= g.H) rr -= g.H;
    int cc = w.wc * 8 + (ri & 7) + g.shift;
    if (cc >= g.W) cc -= g.W;
    return ((int64_t)w.b * g.H + rr) * g.W + cc;
  };
  // Every staging register is (re)defined by every fetch — rows past 63 and the fetch after the last
  // window re-read a valid row — so none of them is live across the register-hungry phases: a spilled
  // prefetch register makes the wave WAIT for its load right after issuing it.
  CH regs[MAXR2];
  auto fetch = [&](const WinPos& w) {
#pragma unroll
    for (int k = 0; k < MAXR2; ++k) {
      const int ri = row0 + NW2 * RPI * k < 64 ? row0 + NW2 * RPI * k : row0;
      const int64_t t = token(w, ri);
      const char* src = (sec < 3 ? reinterpret_cast<const char*>(p.qkv + t * p.ld) : reinterpret_cast<const char*>(p.dout + t * p.ldd)) + goff;
      regs[k] = *reinterpret_cast<const CH*>(src);
    }
  };
  f32x16 Dsum[2];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int v = 0; v < 16; ++v) Dsum[kt][v] = 0.f;

  int win = group;
  WinPos cur = locate(win < nwin ? win : 0);
  fetch(cur);   // in flight while the table is staged
  {  // relative-position table / scale, x-reversed, two copies (see wattn_mfma_hd.hip)
    constexpr int NT_SRC = 225 * HEADS, NLD = (NT_SRC + NT2 - 1) / NT2;
    float tv[NLD];
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int j0 = tid + NT2 * k, j = j0 < NT_SRC ? j0 : NT_SRC - 1;
      tv[k] = p.table[(j / HEADS) * HEADS_ALL + hg * HEADS + (j % HEADS)];
    }
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int j = tid + NT2 * k;
      if (j < NT_SRC) {
        const int rel = j / HEADS, hh = j - rel * HEADS;
        const int dy = rel / 15, u = 14 - (rel - dy * 15);
        const float v = tv[k] * rscale;
        tabL[(hh * 15 + dy) * TSX + u] = v;
        if (u >= 1) tabL[CF::TABB + (hh * 15 + dy) * TSX + u - 1] = v;
      }
    }
  }
  for (; win < nwin; win += p.G) {
    __syncthreads();  // b0: the previous window's rows have left the sections
#pragma unroll
    for (int k = 0; k < MAXR2; ++k) {
      const int ri = row0 + NW2 * RPI * k;
      if (ri < 64 && ld_act) chunk_to_lds<CH>(my_lds + ri * ldt, regs[k]);
    }
    if (tid < 4 * 64) {  // pad columns [SEC, ldt) of the four sections: padded k-steps must read finite zeros
      constexpr int padw = (ldt - secb) / 4;
      const int s4 = tid >> 6, row = tid & 63;
      const int so = s4 == 0 ? CF::OFF_Q : s4 == 1 ? CF::OFF_K : s4 == 2 ? CF::OFF_V : CF::OFF_DO;
#pragma unroll
      for (int w4 = 0; w4 < padw; ++w4) *reinterpret_cast<uint32_t*>(smem + so + row * ldt + secb + 4 * w4) = 0u;
    }
    __syncthreads();  // b1
    stamp();  // 1 + 6k: staged
    const WinPos w = cur;
    {
      int tidw = tid;
      asm volatile("" : "+v"(tidw));
      lane_ctx(tidw);
    }
    c.mrow = g.shift > 0 && w.wr == g.nWh - 1;
    c.mcol = g.shift > 0 && w.wc == g.nWw - 1;
    c.masked = __builtin_amdgcn_readfirstlane((int)(c.mrow || c.mcol)) != 0;

    Pack16 pP[2][2], pdS[2][2];
    f32x16 dq, dv, dk;
#define RDST_BW_HEADS(CALL)                                                                    \
    if constexpr (HEADS == 2) { if (hd == 0) { CALL(0); } else { CALL(1); } }                   \
    else switch (hd) {                                                                          \
      case 0: CALL(0); break; case 1: CALL(1); break; case 2: CALL(2 % HEADS); break;           \
      case 3: CALL(3 % HEADS); break; case 4: CALL(4 % HEADS); break; default: CALL(5 % HEADS); break; \
    }
#define RDST_BW_A(HD) bw_phase_a<D, HEADS, HD>(c, pP, pdS, dq, Dsum)
#ifndef K2_ABL
#define K2_ABL 0   // compile-time ablations (tools/abl_build.sh): 1 no phase A, 2 no phase B, 4 no P / dS images, 8 no copy-out
#endif
#if K2_ABL & 1
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
        for (int e = 0; e < 4; ++e) { pP[a][b2].w[e] = 0u; pdS[a][b2].w[e] = 0u; }
#pragma unroll
    for (int v = 0; v < 16; ++v) dq[v] = 0.f;
#else
    RDST_BW_HEADS(RDST_BW_A)
#endif
    __syncthreads();  // b2: nobody reads K / V any more: the P / dS images may overlay them
    stamp();  // 2 + 6k: phase A done
    if (!(K2_ABL & 4)) bw_store_p<D, HEADS>(c, pP, pdS);
    // The next window's rows are fetched as early as the registers allow: in flight during phase B, the gradient
    // stores and the copy-out (never across phase A, the register-hungry one: a spilled prefetch register makes the
    // wave WAIT for its load).  Measured cold, per launch: D = 10: 76.5 -> 73.1 us, D = 20: 88.3 -> 81.6 us; D = 15
    // (12-byte chunks) spilled that way (80 -> 91 us) until the context was rebuilt per window: now 80.4 -> 78.0 us.
    constexpr bool EARLY_FETCH = true;
    if constexpr (EARLY_FETCH) {
      const int nxt = win + p.G;
      cur = locate(nxt < nwin ? nxt : win);
      fetch(cur);
    }
    __syncthreads();  // b3
    stamp();  // 3 + 6k: P/dS stored
#define RDST_BW_B(HD) bw_phase_b<D, HEADS, HD>(c, dv, dk)
#if K2_ABL & 2
#pragma unroll
    for (int v = 0; v < 16; ++v) { dv[v] = 0.f; dk[v] = 0.f; }
#else
    RDST_BW_HEADS(RDST_BW_B)
#endif
    __syncthreads();  // b4: Q, dOut, P, dS are dead: the gradient tiles go where Q / K / V were
    stamp();  // 4 + 6k: phase B done
    if constexpr (!EARLY_FETCH) {
      const int nxt = win + p.G;
      cur = locate(nxt < nwin ? nxt : win);
      fetch(cur);
    }
#define RDST_BW_ST(HD)                                                                        \
    {                                                                                         \
      store_tile_rows<CF::og(HD), HD * D, HD * D + D>(c.rowT, dq, c.scale, h);                 \
      store_tile_rows<CF::og(HD), HD * D, HD * D + D>(c.rowT + (CF::OFF_K - CF::OFF_Q), dk, c.scale, h);                 \
      store_tile_rows<CF::og(HD), HD * D, HD * D + D>(c.rowT + (CF::OFF_V - CF::OFF_Q), dv, 1.0f, h);                    \
    }
    RDST_BW_HEADS(RDST_BW_ST)
#undef RDST_BW_ST
#undef RDST_BW_B
#undef RDST_BW_A
#undef RDST_BW_HEADS
    __syncthreads();  // b5
    stamp();  // 5 + 6k: tiles stored
#pragma unroll
    for (int k = 0; k < MAXR2; ++k) {
      const int ri = row0 + NW2 * RPI * k;
      if (ri < 64 && st_act && !(K2_ABL & 8)) {
        const int64_t t = token(w, ri);
        char* dst = reinterpret_cast<char*>(p.dqkv + t * p.ldq) + goff;
        *reinterpret_cast<CH*>(dst) = chunk_from_lds<CH>(my_lds + ri * ldt);
      }
    }
    stamp();  // 6 + 6k: copy-out issued
  }
  // d(table): the per-lane partial sums of all windows of this workgroup -> one slab row [HEADS][225]
  __syncthreads();
  if (RDST_DBGV(p.stamps) && tid == 0) p.stamps[(size_t)blockIdx.x * 16 + 14] = __builtin_readcyclecounter();   // slot 14: loop done
  // Every lane holds 32 sums over this workgroup's windows: dS summed per (query, key) pair of its (tile, head).
  // They are laid out as a dense [head][query][key] matrix in LDS (plain stores) and each of the 15 x 15 relative
  // positions then adds up its diagonal in a fixed order.  (LDS float atomics on the 225 entries — up to 64 lanes of
  // a wave on one address — took 75 thousand cycles here: a third of the kernel, measured.)
  float* ds = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int yj = kt * 4 + (v >> 2), xj = (v & 3) + 4 * h;
      ds[(hd * 64 + yi * 8 + xi) * 65 + yj * 8 + xj] = Dsum[kt][v];
    }
  __syncthreads();
  float* my = p.slab + ((int64_t)group * HEADS_ALL + hg * HEADS) * 225;   // slab row [group][6][225]
  for (int e = tid; e < HEADS * 225; e += NT2) {
    const int hd2 = e / 225, rem = e - hd2 * 225, dy = rem / 15 - 7, dx = rem - (rem / 15) * 15 - 7;
    const float* base = ds + hd2 * 64 * 65;
    // all 64 (query row, query column) candidates with fixed trip counts: the reads are independent and go out
    // together; pairs whose key falls outside the window read entry 0 and add nothing
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
  if (RDST_DBGV(p.stamps) && tid == 0) p.stamps[(size_t)blockIdx.x * 16 + 15] = __builtin_readcyclecounter();   // slot 15: kernel end
}

template <int D, int HEADS, int GRAN>
int launch_bw(const BwArgs& p, int slab_rows, int* nslab, hipStream_t st) {
  using CF = HdB<D, HEADS>;
  auto kern = wattn_bwd_hd_kernel<D, HEADS, GRAN>;
  if (CF::SMEM > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CF::SMEM);
  constexpr int NT2 = 128 * HEADS, NG = HEADS_ALL / HEADS;
  const int64_t nwin = (int64_t)p.g.B * p.g.nWh * p.g.nWw;
  int64_t G = 256;      // window groups: NG workgroups each, 256 x NG workgroups = 12 waves per CU
  if (G > nwin) G = nwin;
  if (G > slab_rows) G = slab_rows;
  *nslab = (int)G;
  BwArgs pg = p;
  pg.G = (int)G;
  const int64_t grid = G * NG;
  static int want_stamps = -1;
  if (want_stamps < 0) {
    const char* e = rdst_dbg_getenv("RDST_K2_STAMPS");
    want_stamps = e ? atoi(e) : 0;
  }
  if (want_stamps > 0) {  // debug: in-kernel phase stamps of every workgroup, summarised on stderr
    BwArgs q = pg;
    const size_t n = (size_t)grid * 16;
    (void)hipMalloc((void**)&q.stamps, n * 8);
    (void)hipMemsetAsync(q.stamps, 0, n * 8, st);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NT2), CF::SMEM, st, q);
    (void)hipStreamSynchronize(st);
    unsigned long long* hst = (unsigned long long*)malloc(n * 8);
    (void)hipMemcpy(hst, q.stamps, n * 8, hipMemcpyDeviceToHost);
    (void)hipFree(q.stamps);
    if (--want_stamps == 0) {
      double sum[16] = {0};
      int cnt[16] = {0};
      for (int64_t w = 0; w < grid; ++w)
        for (int k = 1; k < 16; ++k) {
          if (!hst[w * 16 + k]) continue;
          sum[k] += (double)(hst[w * 16 + k] - hst[w * 16 + k - 1]);
          cnt[k]++;
        }
      fprintf(stderr, "[K2 stamps D=%d grid=%lld] mean ticks between consecutive stamps\n", D, (long long)grid);
      for (int k = 1; k < 16; ++k)
        if (cnt[k]) fprintf(stderr, "  %2d: %9.0f\n", k, sum[k] / cnt[k]);
    }
    free(hst);
    return rdst_launch_status("wattn_bwd_hd");
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NT2), CF::SMEM, st, pg);
  return rdst_launch_status("wattn_bwd_hd");
}

bool aligned_to(const void* a, const void* b, const void* cc, int64_t la, int64_t lb, int64_t lc, int gsz) {
  return (uintptr_t)a % gsz == 0 && (uintptr_t)b % gsz == 0 && (uintptr_t)cc % gsz == 0 && la % gsz == 0 && lb % gsz == 0 &&
         lc % gsz == 0;
}

}  // namespace

// bf16, ws 8, 6 heads of dim 10 / 15 / 20, no explicit mask, scale > 0; RDST_ENOTSUP otherwise.
// slab: [slab_rows >= 1][6][225] floats; *nslab = rows written (summed by the caller, dtable_reduce)
int wattn_bwd_mfma_hd(const void* qkv, int64_t ld, const float* table, const void* dout, int64_t ldd, void* dqkv,
                      int64_t ldq, float* slab, int slab_rows, const WinGeom& g, float scale, int* nslab, hipStream_t st) {
  if (g.ws != 8 || g.heads != 6 || g.mask || !(scale > 0.f) || g.C % 6 || slab_rows < 1) return RDST_ENOTSUP;
  static int v1 = -1;
  if (v1 < 0) {
    const char* e = rdst_dbg_getenv("RDST_K2_V1");
    v1 = (e && e[0] == '1') ? 1 : 0;
  }
  if (v1) return RDST_ENOTSUP;
  BwArgs p{};
  p.qkv = (const bf16*)qkv; p.ld = ld; p.table = table; p.dout = (const bf16*)dout; p.ldd = ldd; p.dqkv = (bf16*)dqkv;
  p.ldq = ldq; p.slab = slab; p.g = g; p.scale = scale;
  const int d = g.C / 6;
  const int64_t a = ld * 2, b = ldd * 2, cc = ldq * 2;
#ifndef K2_NH
#define K2_NH 6    // heads per workgroup: 6 = one 12-wave workgroup per window (rounds 1-3; now the fallback for strided rows); 2 = (window, head pair) items with register-staged rows, three workgroups per CU (measured round 4: no faster, see wattn_bwd_pair.hip)
#endif
  if (d == 10 && aligned_to(qkv, dout, dqkv, a, b, cc, 8)) return launch_bw<10, K2_NH, 8>(p, slab_rows, nslab, st);
  if (d == 15 && aligned_to(qkv, dout, dqkv, a, b, cc, 4)) return launch_bw<15, K2_NH, 12>(p, slab_rows, nslab, st);
  if (d == 20 && aligned_to(qkv, dout, dqkv, a, b, cc, 16)) return launch_bw<20, K2_NH, 16>(p, slab_rows, nslab, st);
  return RDST_ENOTSUP;
}
