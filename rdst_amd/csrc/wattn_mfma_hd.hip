// K1 (window attention forward) specialised at compile time for the shapes of the RDST-E1 family:
// bf16, 8x8 windows, HEADS heads of dim D (C = HEADS*D = 60 / 90 / 120).  Same data flow as the
// shape-generic matrix-core kernel (wattn_mfma.hip: one 4-wave workgroup per window, Q/K/V sections
// in LDS, S^T = K.Q^T so the softmax is in-register), re-cut to minimise VECTOR instructions — the
// generic kernel is vector-issue bound (~430 VALU per head and wave against 6 MFMAs):
//   * head ranges are compile-time, so the Q-pack masks are constants (<= 4 v_and per k-step), every
//     LDS address is one per-lane base + an immediate, and the head loop is fully unrolled;
//   * q is NOT pre-scaled at copy-in.  With X' = q.k + bias/scale (bias/scale = initial accumulator),
//     p = exp2(scale2*X' - scale2*max X'): the scale rides on the fma that subtracts the max;
//   * the relative-position table is staged x-reversed in two copies (the second shifted by one
//     float) so each lane fetches its 32 bias values per head as 16 aligned ds_read_b64;
//   * wave (qt, hg) owns the CONTIGUOUS heads hg*HEADS/2 .. and computes O^T = V^T . P^T (operands of
//     the generic kernel swapped): channel rows land in the accumulator registers, query on the lane.
//     P stays un-normalised; 1/l is applied while the head's rows are merged into the wave's output
//     tile(s), which go to LDS once per window as 8-B stores (4 channels of one query);
//   * cross-half reductions use v_permlane32_swap instead of ds_bpermute.
#include "wattn_hd.h"
#include <stdlib.h>

namespace {
using namespace wahd;

struct HdArgs {
  const bf16* qkv; int64_t ld;
  const float* table;
  bf16* out; int64_t ldo;
  WinGeom g;
  float scale;
  int dbg;  // RDST_K1_DEBUG ablation switches: 1 skip compute, 2 skip HBM loads, 4 skip HBM stores
  unsigned long long* stamps;  // RDST_K1_STAMPS=1: [grid][16] s_memtime stamps of wave 0 (debug only), else NULL
};

template <int D, int HEADS>
struct Hd {
  static constexpr int C = D * HEADS;
  static constexpr int SEC = C * 2;                                  // bytes of one Q/K/V section of a token row
  static constexpr int LDT0 = ((SEC + 31) / 32) * 32;
  // odd number of 16-B slots (b128 row reads); a 256-B row gets 80 more, not 16: at 272 the four rows of a transposed
  // read (ds_read_b64_tr_b16: 4 rows x 64 B per 16 lanes) would sit 16 B apart on the same banks (4-way conflicts)
  static constexpr int LDT = LDT0 % 256 == 0 ? LDT0 + 80 : (LDT0 / 16) % 2 == 0 ? LDT0 + 16 : LDT0;
  static constexpr int HW = HEADS / 2;                               // heads per wave group
  static constexpr int TABF = HEADS * 15 * TSX;                      // floats per table copy
  static constexpr int TABB = TABF + 8;                              // float offset of the shifted copy
  static constexpr size_t SMEM = (size_t)3 * 64 * LDT + (size_t)(TABB + TABF + 8) * 4;
  static constexpr int lo(int hg) { return hg * HW * D; }            // channels owned by wave group hg
  static constexpr int hi(int hg) { return (hg + 1) * HW * D; }
  static constexpr int og(int hg) { return lo(hg) & ~3; }            // first channel row of its output tiles
  static constexpr int ntiles(int hg) { return (hi(hg) - og(hg) + 31) / 32; }
  static constexpr int NT = ntiles(0) > ntiles(1) ? ntiles(0) : ntiles(1);
};

template <int D, int HEADS> struct TotT { f32x16 t[Hd<D, HEADS>::NT]; };

struct HdCtx {
  lds_cp Qp, Kp, Vp, Op;         // per-lane bases into the Q / K / V sections (O overwrites Q)
  const LDS_AS f32x2* tb;        // per-lane base into the staged table (copy chosen by parity)
  int h;                         // lane half
  bool masked, mrow, mcol;
  int thr;
  float scale2;
  Pack16 mK[2], mQ;   // one-hot region operands of the mask k-step (rebuilt per masked window)
};

// One head of one wave: 32 queries (lane & 31) x 64 keys.
template <int D, int HEADS, int HD>
__device__ __forceinline__ void hd_head(TotT<D, HEADS>& tot, const HdCtx& c) {
  using CF = Hd<D, HEADS>;
  constexpr int ldt = CF::LDT;
  constexpr int HG = HD / CF::HW;
  constexpr int c_lo = HD * D, c_hi = c_lo + D;
  constexpr int t_lo = c_lo / 16, t_hi = (c_hi - 1) / 16;
  constexpr int OG = CF::og(HG), LO = CF::lo(HG), HI = CF::hi(HG);
  const int h = c.h;

  // X[kt][v]: key j = kt*32 + acc_row(v, h) = (yj = kt*4 + (v>>2), xj = 4h + (v&3)), query = the lane's
  f32x16 X[2];
  const LDS_AS f32x2* tbh = c.tb + HD * (15 * TSX / 2);
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int v = 0; v < 16; v += 2) {
      const f32x2 b2 = lds_read_f32x2(tbh + ((7 - (kt * 4 + (v >> 2))) * TSX + (v & 3)) / 2);
      X[kt][v] = b2.x;
      X[kt][v + 1] = b2.y;
    }
#pragma unroll
  for (int t = t_lo; t <= t_hi; ++t) {
    Pack16 qb = lds_pack(c.Qp + t * 32);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t mA = qmask_bits(t * 16 + 2 * e, c_lo, c_hi), mB = qmask_bits(t * 16 + 8 + 2 * e, c_lo, c_hi);
      qb.w[e] &= h ? mB : mA;
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      const Pack16 ka = lds_pack(c.Kp + kt * 32 * ldt + t * 32);
      Mma<bf16>::mma(X[kt], ka, qb);
    }
  }
  if (c.masked) {
    // Shifted-window mask (wave-uniform: only the last window row / column of a shifted block) as ONE more
    // k-step: the reference adds -100 where the regions of query and key differ; a softmax row is invariant to a
    // constant, so +100 where they AGREE is the same thing, and [region_j == region_i] is a rank-4 product of
    // one-hot vectors — no vector-ALU work per logit at all.
    Mma<bf16>::mma(X[0], c.mK[0], c.mQ);
    Mma<bf16>::mma(X[1], c.mK[1], c.mQ);
  }
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

  // un-normalised P^T as the B operand: element jj of lane half h is key 16s + 8(jj>>2) + 4h + (jj&3)
  Pack16 pb[2][2];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int e = 0; e < 4; ++e) pb[kt][s].w[e] = pack_bf16x2(X[kt][8 * s + 2 * e], X[kt][8 * s + 2 * e + 1]);

  const float inv = __builtin_amdgcn_rcpf(half_swap_sum(l0 + l1));
  typedef LDS_AS s16x4_t* lds_tr_p;
#pragma unroll
  for (int tt = 0; tt < CF::NT; ++tt) {
    const int r_lo = OG + 32 * tt;            // channel rows of this output tile: [r_lo, r_lo + 32)
    if (r_lo + 32 <= c_lo || r_lo >= c_hi) continue;
    f32x16 acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const lds_cp vb = c.Vp + r_lo * 2 + (kt * 32 + 16 * s) * ldt;
        const s16x4_t b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_p)(vb));
        const s16x4_t b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_p)(vb + 8 * ldt));
        const u32x2_t u0 = __builtin_bit_cast(u32x2_t, b0), u1 = __builtin_bit_cast(u32x2_t, b1);
        Pack16 va;
        va.w[0] = u0.x; va.w[1] = u0.y; va.w[2] = u1.x; va.w[3] = u1.y;
        Mma<bf16>::mma(acc, va, pb[kt][s]);   // rows = channels (V^T), cols = queries
      }
    // merge the head's rows into the wave's output tile, normalising on the way
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int r0 = r_lo + (v & 3) + 8 * (v >> 2), r1 = r0 + 4;   // rows of lane half 0 / 1
      const bool own0 = r0 >= LO && r0 < HI, own1 = r1 >= LO && r1 < HI;
      const bool in0 = own0 && r0 >= c_lo && r0 < c_hi, in1 = own1 && r1 >= c_lo && r1 < c_hi;
      const int first = own0 ? r0 / D : r1 / D;   // the wave's first head that touches this register
      if (in0 && in1) {
        tot.t[tt][v] = acc[v] * inv;
      } else if (in0 || in1) {
        const bool sel = in0 ? (h == 0) : (h != 0);
        const float keep = (first == HD) ? 0.f : tot.t[tt][v];
        tot.t[tt][v] = sel ? acc[v] * inv : keep;
      }
    }
  }
}

template <int D, int HEADS, int HD, int END>
struct HeadLoop {
  static __device__ __forceinline__ void run(TotT<D, HEADS>& tot, const HdCtx& c) {
    hd_head<D, HEADS, HD>(tot, c);
    HeadLoop<D, HEADS, HD + 1, END>::run(tot, c);
  }
};
template <int D, int HEADS, int END>
struct HeadLoop<D, HEADS, END, END> {
  static __device__ __forceinline__ void run(TotT<D, HEADS>&, const HdCtx&) {}
};

// All heads of wave group HG, then its output tile(s) -> the (dead) Q channels of the wave's query rows.
template <int D, int HEADS, int HG>
__device__ __forceinline__ void hd_group(const HdCtx& c) {
  using CF = Hd<D, HEADS>;
  constexpr int OG = CF::og(HG), LO = CF::lo(HG), HI = CF::hi(HG);
  TotT<D, HEADS> tot;
#pragma unroll
  for (int tt = 0; tt < CF::NT; ++tt)
#pragma unroll
    for (int v = 0; v < 16; ++v) tot.t[tt][v] = 0.f;
  HeadLoop<D, HEADS, HG * CF::HW, (HG + 1) * CF::HW>::run(tot, c);
  const int h = c.h;
#pragma unroll
  for (int tt = 0; tt < CF::NT; ++tt)
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int cb = OG + 32 * tt + 8 * g4;       // lane half 0: channels cb..cb+3, half 1: cb+4..cb+7
      const bool full0 = cb >= LO && cb + 3 < HI, full1 = cb + 4 >= LO && cb + 7 < HI;
      const bool any0 = cb + 3 >= LO && cb < HI, any1 = cb + 7 >= LO && cb + 4 < HI;
      if (!any0 && !any1) continue;
      if (full0 && full1) {
        u32x2_t w;
        w.x = pack_bf16x2(tot.t[tt][4 * g4], tot.t[tt][4 * g4 + 1]);
        w.y = pack_bf16x2(tot.t[tt][4 * g4 + 2], tot.t[tt][4 * g4 + 3]);
        *reinterpret_cast<LDS_AS u32x2_t*>(c.Op + cb * 2) = w;   // Op already carries the +8h
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool w0 = cb + e >= LO && cb + e < HI, w1 = cb + 4 + e >= LO && cb + 4 + e < HI;
          const bool wr = h ? w1 : w0;
          if (wr) *reinterpret_cast<LDS_AS uint16_t*>(c.Op + (cb + e) * 2) = __builtin_bit_cast(uint16_t, (__bf16)tot.t[tt][4 * g4 + e]);
        }
      }
    }
}

template <int D, int HEADS, int GRAN, int ITERS, int MINB>
__global__ void __launch_bounds__(256, MINB) wattn_fwd_hd_kernel(const HdArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using CF = Hd<D, HEADS>;
  using CH = typename Chunk<GRAN>::type;
  constexpr int C = CF::C, ldt = CF::LDT, secb = CF::SEC;
  const WinGeom g = p.g;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  char* Qs = smem;
  char* Ks = Qs + 64 * ldt;
  char* Vs = Ks + 64 * ldt;
  float* tabL = reinterpret_cast<float*>(Vs + 64 * ldt);  // copy A [HEADS][15][TSX], copy B at +TABB

  int nst = 0;
  auto stamp = [&]() {
    if (RDST_DBGV(p.stamps) && tid == 0 && nst < 16) p.stamps[(size_t)blockIdx.x * 16 + nst++] = __builtin_readcyclecounter();
  };
  stamp();  // 0: kernel start
  constexpr float LOG2E = 1.4426950408889634f;
  const float rscale = 1.0f / p.scale;
  const int nW = g.nWh * g.nWw;
  const int nwin = g.B * nW;

  constexpr int cps = secb / GRAN;  // chunks per section
  constexpr int per_row = 3 * cps;
  const int wv = __builtin_amdgcn_readfirstlane(wave);
  const int qt = wv & 1, hg = wv >> 1;
  const int yi = qt * 4 + (r >> 3), xi = r & 7;
  const int thr = g.ws - g.shift;

  HdCtx c;
  c.h = h;
  c.Qp = (lds_cp)(Qs + (qt * 32 + r) * ldt + h * 16);
  c.Kp = (lds_cp)(Ks + r * ldt + h * 16);
  {
    const int gq = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    c.Vp = (lds_cp)(Vs + (4 * h + q) * ldt + (16 * (gq & 1) + 4 * pp) * 2);
  }
  c.Op = (lds_cp)(Qs + (qt * 32 + r) * ldt + 8 * h);
  {
    const int u0 = 4 * h - xi + 7;  // table column of the lane's first key column (xj = 4h)
    const float* tb = (u0 & 1) ? tabL + CF::TABB + yi * TSX + (u0 - 1) : tabL + yi * TSX + u0;
    c.tb = (const LDS_AS f32x2*)tb;
  }
  c.thr = thr;
  const uint32_t cbits = (uint32_t)__builtin_bit_cast(uint16_t, (__bf16)(100.0f * rscale));   // bf16(100 / scale)
  c.scale2 = p.scale * LOG2E;

  // Software pipeline over the workgroup's windows: the NEXT window's token rows are in flight (in
  // registers) while the current one is computed, so every workgroup keeps HBM requests outstanding
  // all the time instead of only during its copy-in phase.
  struct WinPos { int wr, wc, c0; int64_t rbase[2]; };
  auto locate = [&](int win) {
    WinPos w;
    const int b = win / nW, wi = win - b * nW;
    w.wr = wi / g.nWw;
    w.wc = wi - w.wr * g.nWw;
    w.c0 = w.wc * 8 + g.shift;
#pragma unroll
    for (int yy = 0; yy < 2; ++yy) {  // token rows of this wave: window rows y = 2*wv, 2*wv+1 (8 tokens each)
      int rr = w.wr * 8 + wv * 2 + yy + g.shift;
      if (rr >= g.H) rr -= g.H;
      w.rbase[yy] = ((int64_t)b * g.H + rr) * g.W;
    }
    return w;
  };
  CH regs[16][ITERS];
  auto fetch = [&](const WinPos& w) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      int col = w.c0 + (i & 7);
      if (col >= g.W) col -= g.W;
      const int64_t tok = (RDST_DBGV(p.dbg) & 2) ? 0 : w.rbase[i >> 3] + col;
      const char* src = reinterpret_cast<const char*>(p.qkv + tok * p.ld);
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        int ch = lane + 64 * it;
        ch = ch < per_row ? ch : per_row - 1;  // clamp, not predicate: keeps the staging registers SROA-able
        regs[i][it] = *reinterpret_cast<const CH*>(src + (size_t)ch * GRAN);
      }
    }
  };
  int win = blockIdx.x;
  WinPos cur = locate(win < nwin ? win : 0);
  fetch(cur);   // in flight while the table is staged
  // table, x-reversed: A[u] = T(dy, 14-u) so that u = xj - xi + 7 ascends with the key column;
  // B[u] = A[u+1] serves the lanes whose first u is odd with the same aligned 8-B reads.
  // Coalesced loads, all issued before the first store (one latency, not one per element).
  {
    constexpr int NT_SRC = 225 * HEADS, NLD = (NT_SRC + 255) / 256;
    float tv[NLD];
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int j = tid + 256 * k;
      tv[k] = p.table[j < NT_SRC ? j : NT_SRC - 1];
    }
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int j = tid + 256 * k;
      if (j < NT_SRC) {
        const int rel = j / HEADS, hd = j - rel * HEADS;
        const int dy = rel / 15, u = 14 - (rel - dy * 15);
        const float v = tv[k] * rscale;
        tabL[(hd * 15 + dy) * TSX + u] = v;
        if (u >= 1) tabL[CF::TABB + (hd * 15 + dy) * TSX + u - 1] = v;
      }
    }
  }
  {  // zero the pad columns [SEC, ldt) of every section once: padded k-steps must read zeros
    constexpr int padw = (ldt - secb) / 4;
    for (int idx = tid; idx < 3 * 64 * padw; idx += 256) {
      const int row = idx / padw, w = idx - row * padw;
      *reinterpret_cast<uint32_t*>(Qs + (size_t)row * ldt + secb + 4 * w) = 0u;
    }
  }
  stamp();  // 1: prologue done (first fetch issued, table staged)
  for (; win < nwin; win += gridDim.x) {
    __syncthreads();  // the previous window's O has been copied out: the tile may be overwritten
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = wv * 16 + i;
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        const int ch = lane + 64 * it;
        if (ch < per_row) {
          const int off = ch * GRAN;
          const int sec = (off >= secb) + (off >= 2 * secb);
          chunk_to_lds<CH>(smem + sec * (64 * ldt - secb) + row * ldt + off, regs[i][it]);
        }
      }
    }
    __syncthreads();
    stamp();  // 2 + 3k: window k staged in LDS
    const WinPos w = cur;
    // every fetch defines every staging register (past the last window it re-reads this one): a conditionally
    // defined register stays live around the loop and costs registers / invites spills
    const int nxt = win + gridDim.x;
    cur = locate(nxt < nwin ? nxt : win);
    fetch(cur);

    c.mrow = g.shift > 0 && w.wr == g.nWh - 1;
    c.mcol = g.shift > 0 && w.wc == g.nWw - 1;
    c.masked = __builtin_amdgcn_readfirstlane((int)(c.mrow || c.mcol)) != 0;
    if (c.masked) {   // region id = 2 [row >= thr] + [col >= thr] (a flag only counts on the block's last window row / column)
      auto onehot = [&](int reg, uint32_t v, Pack16& q) {   // elements 0..3 of lane half 0 (k = 0..3 of the k-step)
        q.w[0] = h ? 0u : ((reg == 0 ? v : 0u) | (reg == 1 ? v << 16 : 0u));
        q.w[1] = h ? 0u : ((reg == 2 ? v : 0u) | (reg == 3 ? v << 16 : 0u));
        q.w[2] = 0u;
        q.w[3] = 0u;
      };
      const int rx = (c.mcol && xi >= thr) ? 1 : 0;          // xi = r & 7 is also the key column of row r of a key tile
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) onehot(2 * ((c.mrow && kt * 4 + (r >> 3) >= thr) ? 1 : 0) + rx, 0x3f80u, c.mK[kt]);
      onehot(2 * ((c.mrow && yi >= thr) ? 1 : 0) + rx, cbits, c.mQ);
    }
    if (!(RDST_DBGV(p.dbg) & 1)) {
      if (hg == 0) hd_group<D, HEADS, 0>(c);
      else hd_group<D, HEADS, 1>(c);
    }
    __syncthreads();
    stamp();  // 3 + 3k: computed
    if (!(RDST_DBGV(p.dbg) & 4)) {  // LDS (Q section now holds O) -> global rows
      if constexpr (cps <= 16) {  // 4 token rows per store instruction: lane -> (row = lane >> 4, chunk = lane & 15)
        const int rsub = lane >> 4, chk = lane & 15;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int i = 4 * j + rsub;
          int col = w.c0 + (i & 7);
          if (col >= g.W) col -= g.W;
          const int64_t tok = (j < 2 ? w.rbase[0] : w.rbase[1]) + col;
          char* dst = reinterpret_cast<char*>(p.out + tok * p.ldo);
          const char* src = Qs + (size_t)(wv * 16 + i) * ldt;
          if (chk < cps) *reinterpret_cast<CH*>(dst + (size_t)chk * GRAN) = chunk_from_lds<CH>(src + (size_t)chk * GRAN);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          int col = w.c0 + (i & 7);
          if (col >= g.W) col -= g.W;
          const int64_t tok = w.rbase[i >> 3] + col;
          char* dst = reinterpret_cast<char*>(p.out + tok * p.ldo);
          const char* src = Qs + (size_t)(wv * 16 + i) * ldt;
          if (lane < cps) *reinterpret_cast<CH*>(dst + (size_t)lane * GRAN) = chunk_from_lds<CH>(src + (size_t)lane * GRAN);
        }
      }
    }
    stamp();  // 4 + 3k: copy-out issued
  }
}

template <int D, int HEADS, int GRAN, int ITERS, int MINB>
int launch_hd(const HdArgs& p, hipStream_t st) {
  using CF = Hd<D, HEADS>;
  auto kern = wattn_fwd_hd_kernel<D, HEADS, GRAN, ITERS, MINB>;
  if (CF::SMEM > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CF::SMEM);
  const int64_t nwin = (int64_t)p.g.B * p.g.nWh * p.g.nWw;
  int wg_per_cu = (int)((160 * 1024) / CF::SMEM);
  if (wg_per_cu > MINB) wg_per_cu = MINB;
  {
    static int env_wgs = -1;
    if (env_wgs < 0) {
      const char* e = rdst_dbg_getenv("RDST_K1_WGS");
      env_wgs = e ? atoi(e) : 0;
    }
    if (env_wgs > 0 && env_wgs < wg_per_cu) wg_per_cu = env_wgs;
  }
  int64_t grid = 256 * (int64_t)wg_per_cu;
  if (grid > nwin) grid = nwin;
  static int want_stamps = -1;
  if (want_stamps < 0) {
    const char* e = rdst_dbg_getenv("RDST_K1_STAMPS");
    want_stamps = e ? atoi(e) : 0;
  }
  if (want_stamps > 0) {  // debug: in-kernel phase stamps of every workgroup, summarised on stderr
    HdArgs q = p;
    const size_t n = (size_t)grid * 16;
    (void)hipMalloc((void**)&q.stamps, n * 8);
    (void)hipMemsetAsync(q.stamps, 0, n * 8, st);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), CF::SMEM, st, q);
    (void)hipStreamSynchronize(st);
    unsigned long long* hst = (unsigned long long*)malloc(n * 8);
    (void)hipMemcpy(hst, q.stamps, n * 8, hipMemcpyDeviceToHost);
    (void)hipFree(q.stamps);
    unsigned long long t0 = ~0ull;
    for (int64_t w = 0; w < grid; ++w) if (hst[w * 16] && hst[w * 16] < t0) t0 = hst[w * 16];
    if (--want_stamps == 0) {  // print for the last requested launch only
      double sum[16] = {0}, mx[16] = {0}, mn[16];
      int cnt[16] = {0};
      for (int k = 0; k < 16; ++k) mn[k] = 1e30;
      for (int64_t w = 0; w < grid; ++w)
        for (int k = 0; k < 16; ++k) {
          const unsigned long long t = hst[w * 16 + k];
          if (!t) continue;
          const double d = (double)(t - hst[w * 16]);  // relative to the workgroup's own start (clocks differ per XCD)
          sum[k] += d; cnt[k]++;
          if (d > mx[k]) mx[k] = d;
          if (d < mn[k]) mn[k] = d;
        }
      fprintf(stderr, "[K1 stamps D=%d grid=%lld] ticks since the workgroup's own start (k: n mean min max)\n", D, (long long)grid);
      for (int k = 0; k < 16; ++k)
        if (cnt[k]) fprintf(stderr, "  %2d: %5d %9.0f %9.0f %9.0f\n", k, cnt[k], sum[k] / cnt[k], mn[k], mx[k]);
    }
    free(hst);
    return rdst_launch_status("wattn_fwd_hd");
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), CF::SMEM, st, p);
  return rdst_launch_status("wattn_fwd_hd");
}

bool aligned_to(const void* a, const void* b, int64_t lda_bytes, int64_t ldb_bytes, int gsz) {
  return (uintptr_t)a % gsz == 0 && (uintptr_t)b % gsz == 0 && lda_bytes % gsz == 0 && ldb_bytes % gsz == 0;
}

}  // namespace

// bf16, ws 8, 6 heads of dim 10 / 15 / 20, no explicit mask, scale > 0; RDST_ENOTSUP otherwise
int wattn_fwd_mfma_hd(const void* qkv, int64_t ld, const float* table, void* out, int64_t ldo, const WinGeom& g,
                      float scale, hipStream_t st) {
  if (g.ws != 8 || g.heads != 6 || g.mask || !(scale > 0.f) || g.C % 6) return RDST_ENOTSUP;
  static int v1 = -1;
  if (v1 < 0) {
    const char* e = rdst_dbg_getenv("RDST_K1_V1");
    v1 = (e && e[0] == '1') ? 1 : 0;
  }
  if (v1) return RDST_ENOTSUP;
  HdArgs p{};
  p.qkv = (const bf16*)qkv; p.ld = ld; p.table = table; p.out = (bf16*)out; p.ldo = ldo; p.g = g; p.scale = scale;
  { const char* e = rdst_dbg_getenv("RDST_K1_DEBUG"); p.dbg = e ? atoi(e) : 0; }
  const int d = g.C / 6;
  const int64_t lb = ld * 2, lob = ldo * 2;
  static int minb = -1;
  if (minb < 0) {
    const char* e = rdst_dbg_getenv("RDST_K1_MINB");
    minb = e ? atoi(e) : 0;
  }
  if (d == 10 && aligned_to(qkv, out, lb, lob, 8)) return minb == 4 ? launch_hd<10, 6, 8, 1, 4>(p, st) : launch_hd<10, 6, 8, 1, 3>(p, st);
  if (d == 15 && aligned_to(qkv, out, lb, lob, 4)) return launch_hd<15, 6, 12, 1, 3>(p, st);
  if (d == 20 && aligned_to(qkv, out, lb, lob, 16)) return launch_hd<20, 6, 16, 1, 2>(p, st);
  return RDST_ENOTSUP;
}
