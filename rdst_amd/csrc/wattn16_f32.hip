// Window attention for 16x16 windows in EXACT fp32 on the matrix cores (v_mfma_f32_32x32x2_f32: an fmaf chain), forward and
// backward: the reference-precision path of BASELINE.json configs[3] (reference: networks/swin_transformer_sr.py:110-141 with
// N = 256, the shift mask of :211-232; the reference computes in fp32 only).  6 heads of dim 10 / 15 / 20.
//
// Same geometry as the bf16 kernels of wattn16_mfma.hip (a wave owns a 32-query tile, all 256 keys of the head live in its
// accumulator registers, the relative-position table is staged reversed in two copies, d(table) by lane-shift diagonal sums),
// with what fp32 changes:
//   * a workgroup owns (window, ONE head): 256 rows x 4 sections x (D padded to 16 / 24 floats) is 80 - 112 KB of LDS;
//     the six heads of a window run on the same XCD;
//   * the arithmetic is the reference's: q * scale at copy-in (:117), S = q k^T + bias (:118-126; the bias is the initial
//     accumulator), + (-100) where the shift regions differ (:128-131; one k-step of 0/1 region vectors against a
//     {0, -100} vector: exact), softmax as expf(x - max) * (1 / sum), then P V;
//   * "accumulator as operand": an fp32 accumulator register IS the B operand of a 32x32x2 MFMA (k = lane half), so
//     P V, dQ = dS K, dV = P^T dO and dK = dS^T Q run one MFMA per accumulator register with the A operand (one float of
//     V^T / K^T / dO^T / Q^T per lane) read from LDS: no repacking, no transposed tiles.
#include "wattn_hd.h"
#include "wattn16.h"

namespace {
using namespace wahd;
using namespace w16c;

struct F16Args {
  const float* qkv; int64_t ld;
  const float* table;
  float* out; int64_t ldo;
  const float* dout; int64_t ldd;     // backward only
  float* dqkv; int64_t ldq;
  float* slab;                        // backward: [window][heads][961] partial d(table)
  WinGeom g;
  float scale;
};

template <int D>
struct F16 {
  static constexpr int HEADS = 6, C = HEADS * D;
  static constexpr int NKS = (D + 7) / 8, DP = 8 * NKS;   // k-steps of 8 channels; padded head width
  static constexpr int PB = D * 4;                        // bytes of the head's piece of a row section: 40 / 60 / 80
  static constexpr int CPS = 5, GRAN = PB / CPS;          // five chunks of 8 / 12 / 16 bytes
  static constexpr int LDT = DP * 4 + 16;                 // LDS row stride: 80 / 80 / 112, an odd number of 16-B slots
  static constexpr int SEC = 256 * LDT;
  static constexpr int TROW = 32, TABF = 31 * TROW + 16;  // as in wattn16_mfma.hip: the two copies 16 banks apart
  static constexpr int PADW = LDT / 4 - D;                // floats behind the head's channels in every row
  // backward
  static constexpr int OFF_TABR = 4 * SEC;                     // reversed table, two copies (pass 1)
  static constexpr int OFF_STAT = OFF_TABR + 2 * TABF * 4;     // [max | 1 / sum | -delta][256]
  static constexpr int OFF_PART = OFF_STAT + 3 * 256 * 4;      // d(table) row sums [yi 16][yj 16][32]; pass 2: natural table
  static constexpr size_t SMEM_BWD = (size_t)OFF_PART + 16 * 16 * 32 * 4;
  static constexpr size_t SMEM_FWD = (size_t)3 * SEC + 2 * TABF * 4 + 128;
};

// workgroup -> (window, head): the six heads of a window run on the same XCD (blockIdx round-robins over 8 XCDs)
__device__ __forceinline__ void f16_locate(int nwin, int& win, int& hd) {
  const int b = blockIdx.x;
  if ((nwin & 7) == 0) {
    const int xcd = b & 7, slot = b >> 3;
    hd = slot % 6;
    win = (slot / 6) * 8 + xcd;
  } else {
    hd = b % 6;
    win = b / 6;
  }
}

__device__ __forceinline__ void scale_chunk(u32x2_t& v, float s) {
  v.x = __float_as_uint(__uint_as_float(v.x) * s); v.y = __float_as_uint(__uint_as_float(v.y) * s);
}
__device__ __forceinline__ void scale_chunk(u32x3_a4& v, float s) {
  v.x = __float_as_uint(__uint_as_float(v.x) * s); v.y = __float_as_uint(__uint_as_float(v.y) * s);
  v.z = __float_as_uint(__uint_as_float(v.z) * s);
}
__device__ __forceinline__ void scale_chunk(u32x4_t& v, float s) {
  v.x = __float_as_uint(__uint_as_float(v.x) * s); v.y = __float_as_uint(__uint_as_float(v.y) * s);
  v.z = __float_as_uint(__uint_as_float(v.z) * s); v.w = __float_as_uint(__uint_as_float(v.w) * s);
}

__device__ __forceinline__ f32x16 mfma1(float a, float b, const f32x16& acc) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ float lds_f32(lds_cp p) { return *reinterpret_cast<const LDS_AS float*>(p); }

struct F16Ctx {
  lds_cp Qp, Kp;            // the lane's query row / key row r of tile 0, + 16 B x lane half
  const LDS_AS f32x2* tb;   // lane base into the reversed table copy of its parity
  int h, r, qt, thr;
  bool masked, mrow, mcol;
};

// region code of window position (y, x): 2 (last window row of a shifted block and y >= ws - shift) + (the same for x)
__device__ __forceinline__ int f16_region(const F16Ctx& c, int y, int x) {
  return 2 * ((c.mrow && y >= c.thr) ? 1 : 0) + ((c.mcol && x >= c.thr) ? 1 : 0);
}
__device__ __forceinline__ void region_pack(int reg, bool equal, float val, int h, Pack16& q) {   // k = 0..3 on lane half 0
  const uint32_t bits = __float_as_uint(val);
#pragma unroll
  for (int e = 0; e < 4; ++e) q.w[e] = (h == 0 && ((e == reg) == equal)) ? bits : 0u;
}

// S^T tiles of one query tile: X[kt][v] = (q scale) . k + bias (+ -100 where the regions differ); key j = 32 kt + acc_row(v, h)
// = (yj = 2 kt + (v >> 3), xj = 8 ((v >> 2) & 1) + 4 h + (v & 3)), query = the lane's.
template <int D>
__device__ __forceinline__ void f16_scores(f32x16 (&X)[8], const F16Ctx& c) {
  using CF = F16<D>;
  constexpr int ldt = CF::LDT;
#pragma unroll
  for (int kt = 0; kt < 8; ++kt)
#pragma unroll
    for (int v = 0; v < 16; v += 2) {
      const f32x2 b2 = lds_read_f32x2(c.tb + ((2 * kt + (v >> 3)) * CF::TROW + 8 * ((v >> 2) & 1) + (v & 3)) / 2);
      X[kt][v] = b2.x;
      X[kt][v + 1] = b2.y;
    }
#pragma unroll
  for (int t = 0; t < CF::NKS; ++t) {
    const Pack16 qb = lds_pack(c.Qp + t * 32);
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) Mma<float>::mma(X[kt], lds_pack(c.Kp + kt * 32 * ldt + t * 32), qb);
  }
  if (c.masked) {   // wave-uniform
    Pack16 mQ;
    region_pack(f16_region(c, 2 * c.qt + (c.r >> 4), c.r & 15), false, -100.0f, c.h, mQ);   // -100 at every OTHER region
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
      Pack16 mK;
      region_pack(f16_region(c, 2 * kt + (c.r >> 4), c.r & 15), true, 1.0f, c.h, mK);       // one-hot region of key row r
      Mma<float>::mma(X[kt], mK, mQ);
    }
  }
}

// softmax over the 256 keys of the lane's query, in registers: X <- P^T; returns (max, 1 / sum)
__device__ __forceinline__ void f16_softmax(f32x16 (&X)[8], float& m, float& inv) {
  m = X[0][0];
#pragma unroll
  for (int kt = 0; kt < 8; ++kt)
#pragma unroll
    for (int v = 0; v < 16; ++v) m = __builtin_fmaxf(m, X[kt][v]);
  m = half_swap_max(m);
  float l0 = 0.f, l1 = 0.f;
#pragma unroll
  for (int kt = 0; kt < 8; ++kt)
#pragma unroll
    for (int v = 0; v < 16; v += 2) {
      const float e0 = expf(X[kt][v] - m), e1 = expf(X[kt][v + 1] - m);
      l0 += e0;
      l1 += e1;
      X[kt][v] = e0;
      X[kt][v + 1] = e1;
    }
  inv = 1.0f / half_swap_sum(l0 + l1);
#pragma unroll
  for (int kt = 0; kt < 8; ++kt)
#pragma unroll
    for (int v = 0; v < 16; ++v) X[kt][v] *= inv;
}

// rows (registers) = channels 8 g4 + 4 h + e, lane = token: the 16-byte groups that start inside the head go to the token's
// LDS row (a group may run into the row's padding: nobody reads it as an operand afterwards)
template <int D>
__device__ __forceinline__ void f16_store_tile(lds_cp rowp, const f32x16& t, float mul, int h) {
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) {
    if (8 * g4 >= D) continue;
    if (8 * g4 + 4 * h < D) {
      f32x4 w;
      w.x = t[4 * g4] * mul; w.y = t[4 * g4 + 1] * mul; w.z = t[4 * g4 + 2] * mul; w.w = t[4 * g4 + 3] * mul;
      *reinterpret_cast<LDS_AS f32x4*>(rowp + (8 * g4 + 4 * h) * 4) = w;
    }
  }
}

// reversed table of head hd, two copies: A[dy'][u'] = T[30 - dy'][30 - u'], B[k] = A[k + 1]  (natural: A[dy][u] = T[dy][u])
template <int D, bool REV>
__device__ __forceinline__ void f16_stage_table(float* tab, const float* table, int hd, int tid) {
  using CF = F16<D>;
  float tv[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int j = tid + 512 * k;
    tv[k] = table[(j < 961 ? j : 960) * CF::HEADS + hd];
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int j = tid + 512 * k;
    if (j < 961) {
      const int ry = j / 31, rx = j - ry * 31;
      const int idx = REV ? (30 - ry) * CF::TROW + (30 - rx) : ry * CF::TROW + rx;
      tab[idx] = tv[k];
      if (idx >= 1) tab[CF::TABF + idx - 1] = tv[k];
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Forward: 8 waves, wave = query tile.
template <int D>
__global__ void __launch_bounds__(512, 1) wattn16_f32_fwd_kernel(const F16Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using CF = F16<D>;
  using CH = typename Chunk<CF::GRAN>::type;
  constexpr int ldt = CF::LDT, GRAN = CF::GRAN, CPS = CF::CPS;
  const WinGeom g = p.g;
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* Qs = smem;
  char* Ks = Qs + CF::SEC;
  char* Vs = Ks + CF::SEC;
  float* tabL = reinterpret_cast<float*>(Vs + CF::SEC);

  const int nW = g.nWh * g.nWw, nwin = g.B * nW;
  int win, hd;
  f16_locate(nwin, win, hd);
  const int b = win / nW, wi = win - b * nW, wr = wi / g.nWw, wc = wi - wr * g.nWw;

  // ---- the head's pieces of the window's 256 rows (q | k | v: 15 chunks per row) -> registers, all in flight
  constexpr int NCH = 256 * 3 * CPS, NI = (NCH + 511) / 512;
  CH regs[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    int idx = tid + 512 * i;
    idx = idx < NCH ? idx : NCH - 1;
    const int row = idx / (3 * CPS), k3 = idx - row * (3 * CPS), sec = k3 / CPS, cw = k3 - sec * CPS;
    const int64_t tok = win_token16(b, wr, wc, row, g);
    regs[i] = *reinterpret_cast<const CH*>(reinterpret_cast<const char*>(p.qkv + tok * p.ld + sec * CF::C + hd * D) + cw * GRAN);
  }
  f16_stage_table<D, true>(tabL, p.table, hd, tid);
  for (int idx = tid; idx < 3 * 256 * CF::PADW; idx += 512) {   // zeros behind the head's channels: padded k-steps read them
    const int row = idx / CF::PADW, w = idx - row * CF::PADW;
    *reinterpret_cast<float*>(Qs + (size_t)row * ldt + (D + w) * 4) = 0.f;
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int idx = tid + 512 * i;
    if (idx < NCH) {
      const int row = idx / (3 * CPS), k3 = idx - row * (3 * CPS), sec = k3 / CPS, cw = k3 - sec * CPS;
      CH v = regs[i];
      if (sec == 0) scale_chunk(v, p.scale);   // q = q * self.scale
      chunk_to_lds<CH>(smem + sec * CF::SEC + row * ldt + cw * GRAN, v);
    }
  }
  __syncthreads();

  F16Ctx c;
  c.h = h; c.r = r; c.qt = wv;
  c.thr = g.ws - g.shift;
  c.mrow = g.shift > 0 && wr == g.nWh - 1;
  c.mcol = g.shift > 0 && wc == g.nWw - 1;
  c.masked = c.mrow || c.mcol;
  c.Qp = (lds_cp)(Qs + (wv * 32 + r) * ldt + h * 16);
  c.Kp = (lds_cp)(Ks + r * ldt + h * 16);
  {
    const int yi = 2 * wv + (r >> 4), xi = r & 15;
    const int u0 = (15 - yi) * CF::TROW + 15 - xi + 4 * h;
    c.tb = (const LDS_AS f32x2*)((u0 & 1) ? tabL + CF::TABF + (u0 - 1) : tabL + u0);
  }
  {
    f32x16 X[8];
    f16_scores<D>(X, c);
    float m, inv;
    f16_softmax(X, m, inv);
    f32x16 acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
    const lds_cp vel = (lds_cp)(Vs + 4 * h * ldt + r * 4);   // V^T[channel r][key]: one float of the key's row
#pragma unroll
    for (int kt = 0; kt < 8; ++kt)
#pragma unroll
      for (int v = 0; v < 16; ++v)
        acc = mfma1(lds_f32(vel + (kt * 32 + 8 * (v >> 2) + (v & 3)) * ldt), X[kt][v], acc);   // O^T += V^T . P^T
    f16_store_tile<D>((lds_cp)(Qs + (wv * 32 + r) * ldt), acc, 1.0f, h);   // over the wave's own (dead) query rows
  }
  __syncthreads();
  // O (in the Q section) -> global rows
  for (int idx = tid; idx < 256 * CPS; idx += 512) {
    const int row = idx / CPS, k = idx - row * CPS;
    const int64_t tok = win_token16(b, wr, wc, row, g);
    char* dst = reinterpret_cast<char*>(p.out + tok * p.ldo + hd * D) + k * GRAN;
    *reinterpret_cast<CH*>(dst) = chunk_from_lds<CH>(Qs + (size_t)row * ldt + k * GRAN);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Backward: 8 waves, two passes as in wattn16_mfma.hip (pass 1: wave = query tile -> dQ, d(table), row statistics;
// pass 2: wave = key tile -> dK, dV).
struct F16BCtx {
  F16Ctx sc;
  lds_cp dOp, Vrow, Kel;         // pass 1: dO^T pack base, V rows as A operand, K^T element base
  LDS_AS float* part;            // pass 1: the lane's slot row [yi][.][lane & 15] of the d(table) row sums
  LDS_AS float* stat;
  lds_cp QA, dOA, Qel, dOel, Kown, Vown;   // pass 2
  const LDS_AS f32x2* tb2;       // pass 2: lane base into the natural table copy of its parity
  int kt;
};

template <int D>
__device__ __forceinline__ void f16_bwd_p1(const F16BCtx& c, f32x16& dq) {
  using CF = F16<D>;
  constexpr int ldt = CF::LDT;
  const int h = c.sc.h;
  f32x16 X[8];
  f16_scores<D>(X, c.sc);
  float m, inv;
  f16_softmax(X, m, inv);
  Pack16 dob[CF::NKS];   // dO^T of the lane's query
#pragma unroll
  for (int t = 0; t < CF::NKS; ++t) dob[t] = lds_pack(c.dOp + t * 32);
  auto dp_tile = [&](int kt, f32x16& dp, float init) {
#pragma unroll
    for (int v = 0; v < 16; ++v) dp[v] = init;
#pragma unroll
    for (int t = 0; t < CF::NKS; ++t) Mma<float>::mma(dp, lds_pack(c.Vrow + kt * 32 * ldt + t * 32), dob[t]);
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
    LDS_AS float* st = c.stat + c.sc.qt * 32 + c.sc.r;
    st[0] = m;
    st[256] = inv;
    st[512] = -delta;   // pass 2: the initial accumulator of dP, so that dS = P . acc
  }
#pragma unroll
  for (int v = 0; v < 16; ++v) dq[v] = 0.f;
#pragma unroll
  for (int kt = 0; kt < 8; ++kt) {
    f32x16 dp;
    dp_tile(kt, dp, -delta);
    // register v: key (yj = 2 kt + (v >> 3), xj = XL + 4 h) with XL = 8 ((v >> 2) & 1) + (v & 3); column c' = xi - XL + 15
    // of the key row's sums goes to lane c' of `lo` (c' < 16) / lane c' - 16 of `hi`; true column = c' - 4 h
    float lo[2] = {0.f, 0.f}, hi[2] = {0.f, 0.f};
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const float ds = X[kt][v] * dp[v];
      dp[v] = ds;
      const int yl = v >> 3;
      switch (8 * ((v >> 2) & 1) + (v & 3)) {
#define RDST_F16_DIAG(XL) case XL: lo[yl] += dpp_row_shr<15 - XL>(ds); hi[yl] += dpp_row_shl<XL + 1>(ds); break;
        RDST_F16_DIAG(0) RDST_F16_DIAG(1) RDST_F16_DIAG(2) RDST_F16_DIAG(3)
        RDST_F16_DIAG(8) RDST_F16_DIAG(9) RDST_F16_DIAG(10) RDST_F16_DIAG(11)
#undef RDST_F16_DIAG
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
    for (int v = 0; v < 16; ++v)
      dq = mfma1(lds_f32(c.Kel + (kt * 32 + 8 * (v >> 2) + (v & 3)) * ldt), dp[v], dq);   // dQ^T += K^T . dS^T
  }
}

template <int D>
__device__ __forceinline__ void f16_bwd_p2(const F16BCtx& c, lds_cp Kst, lds_cp Vst) {
  using CF = F16<D>;
  constexpr int ldt = CF::LDT;
  const int h = c.sc.h, r = c.sc.r;
  Pack16 kb[CF::NKS], vb[CF::NKS];   // K^T / V^T of the lane's key
#pragma unroll
  for (int t = 0; t < CF::NKS; ++t) {
    kb[t] = lds_pack(c.Kown + t * 32);
    vb[t] = lds_pack(c.Vown + t * 32);
  }
  Pack16 mKc;   // -100 at every region but the lane's key's
  if (c.sc.masked) region_pack(f16_region(c.sc, 2 * c.kt + (r >> 4), r & 15), false, -100.0f, h, mKc);
  f32x16 dk, dv;
#pragma unroll
  for (int v = 0; v < 16; ++v) { dk[v] = 0.f; dv[v] = 0.f; }
  const LDS_AS char* stb = (const LDS_AS char*)c.stat + h * 16;
#pragma unroll 1
  for (int qt = 0; qt < 8; ++qt) {
    // S tile: rows = queries of tile qt (yi = 2 qt + (v >> 3), xi = 8 ((v >> 2) & 1) + 4 h + (v & 3)), cols = the lane's key
    f32x16 X, dp;
#pragma unroll
    for (int v = 0; v < 16; v += 2) {
      const f32x2 b2 = lds_read_f32x2(c.tb2 + ((2 * qt + (v >> 3)) * CF::TROW + 8 * ((v >> 2) & 1) + (v & 3)) / 2);
      X[v] = b2.x;
      X[v + 1] = b2.y;
    }
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {   // dP starts at -delta of its query row
      const f32x4 nd = *reinterpret_cast<const LDS_AS f32x4*>(stb + (512 + qt * 32 + 8 * g4) * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) dp[4 * g4 + e] = nd[e];
    }
#pragma unroll
    for (int t = 0; t < CF::NKS; ++t) {
      Mma<float>::mma(X, lds_pack(c.QA + qt * 32 * ldt + t * 32), kb[t]);
      Mma<float>::mma(dp, lds_pack(c.dOA + qt * 32 * ldt + t * 32), vb[t]);
    }
    if (c.sc.masked) {
      Pack16 mQr;   // rows = queries of tile qt: row r is query (2 qt + (r >> 4), r & 15)
      region_pack(f16_region(c.sc, 2 * qt + (r >> 4), r & 15), true, 1.0f, h, mQr);
      Mma<float>::mma(X, mQr, mKc);
    }
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const f32x4 mm = *reinterpret_cast<const LDS_AS f32x4*>(stb + (qt * 32 + 8 * g4) * 4);
      const f32x4 iv = *reinterpret_cast<const LDS_AS f32x4*>(stb + (256 + qt * 32 + 8 * g4) * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int v = 4 * g4 + e;
        const float pv = expf(X[v] - mm[e]) * iv[e];
        X[v] = pv;            // P
        dp[v] = pv * dp[v];   // dS
      }
    }
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int ro = (qt * 32 + 8 * (v >> 2) + (v & 3)) * ldt;
      dv = mfma1(lds_f32(c.dOel + ro), X[v], dv);    // dV^T += dO^T . P
      dk = mfma1(lds_f32(c.Qel + ro), dp[v], dk);    // dK^T += (q scale)^T . dS
    }
  }
  f16_store_tile<D>(Kst, dk, 1.0f, h);
  f16_store_tile<D>(Vst, dv, 1.0f, h);
}

template <int D>
__global__ void __launch_bounds__(512, 1) wattn16_f32_bwd_kernel(const F16Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using CF = F16<D>;
  using CH = typename Chunk<CF::GRAN>::type;
  constexpr int ldt = CF::LDT, GRAN = CF::GRAN, CPS = CF::CPS;
  const WinGeom g = p.g;
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* Qs = smem;
  char* Ks = Qs + CF::SEC;
  char* Vs = Ks + CF::SEC;
  char* dOs = Vs + CF::SEC;
  float* tabR = reinterpret_cast<float*>(smem + CF::OFF_TABR);
  float* stat = reinterpret_cast<float*>(smem + CF::OFF_STAT);
  float* part = reinterpret_cast<float*>(smem + CF::OFF_PART);
  float* tabN = part;   // pass 2: the natural table takes the place of the (then summed) row sums

  const int nW = g.nWh * g.nWw, nwin = g.B * nW;
  int win, hd;
  f16_locate(nwin, win, hd);
  const int b = win / nW, wi = win - b * nW, wr = wi / g.nWw, wc = wi - wr * g.nWw;

  // ---- q | k | v | dO pieces of the window's 256 rows: 20 chunks per row, 10 per thread
  constexpr int NCH = 256 * 4 * CPS, NI = NCH / 512;
  static_assert(NCH % 512 == 0, "copy plan");
  CH regs[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int idx = tid + 512 * i;
    const int row = idx / (4 * CPS), k4 = idx - row * (4 * CPS), sec = k4 / CPS, cw = k4 - sec * CPS;
    const int64_t tok = win_token16(b, wr, wc, row, g);
    const float* src = sec < 3 ? p.qkv + tok * p.ld + sec * CF::C + hd * D : p.dout + tok * p.ldd + hd * D;
    regs[i] = *reinterpret_cast<const CH*>(reinterpret_cast<const char*>(src) + cw * GRAN);
  }
  f16_stage_table<D, true>(tabR, p.table, hd, tid);
  for (int idx = tid; idx < 4 * 256 * CF::PADW; idx += 512) {
    const int row = idx / CF::PADW, w = idx - row * CF::PADW;
    *reinterpret_cast<float*>(Qs + (size_t)row * ldt + (D + w) * 4) = 0.f;
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int idx = tid + 512 * i;
    const int row = idx / (4 * CPS), k4 = idx - row * (4 * CPS), sec = k4 / CPS, cw = k4 - sec * CPS;
    CH v = regs[i];
    if (sec == 0) scale_chunk(v, p.scale);
    chunk_to_lds<CH>(smem + sec * CF::SEC + row * ldt + cw * GRAN, v);
  }
  __syncthreads();

  F16BCtx c;
  c.sc.h = h; c.sc.r = r; c.sc.qt = wv;
  c.sc.thr = g.ws - g.shift;
  c.sc.mrow = g.shift > 0 && wr == g.nWh - 1;
  c.sc.mcol = g.shift > 0 && wc == g.nWw - 1;
  c.sc.masked = c.sc.mrow || c.sc.mcol;
  c.stat = (LDS_AS float*)stat;
  const int y0 = 2 * wv + (r >> 4), x0 = r & 15;   // the lane's query (pass 1) / key (pass 2)
  const int u0 = (15 - y0) * CF::TROW + 15 - x0 + 4 * h;
  c.sc.Qp = (lds_cp)(Qs + (wv * 32 + r) * ldt + h * 16);
  c.sc.Kp = (lds_cp)(Ks + r * ldt + h * 16);
  c.sc.tb = (const LDS_AS f32x2*)((u0 & 1) ? tabR + CF::TABF + (u0 - 1) : tabR + u0);
  c.dOp = (lds_cp)(dOs + (wv * 32 + r) * ldt + h * 16);
  c.Vrow = (lds_cp)(Vs + r * ldt + h * 16);
  c.Kel = (lds_cp)(Ks + 4 * h * ldt + r * 4);
  c.part = (LDS_AS float*)(part + (2 * wv + (r >> 4)) * 16 * 32 + (r & 15));
  f32x16 dq;
  f16_bwd_p1<D>(c, dq);
  __syncthreads();
  w16_dtable_out(part, p.slab + ((int64_t)win * CF::HEADS + hd) * 961, tid);
  __syncthreads();
  f16_stage_table<D, false>(tabN, p.table, hd, tid);
  __syncthreads();
  {  // pass 2: wave = key tile wv
    c.kt = wv;
    c.QA = (lds_cp)(Qs + r * ldt + h * 16);
    c.dOA = (lds_cp)(dOs + r * ldt + h * 16);
    c.Qel = (lds_cp)(Qs + 4 * h * ldt + r * 4);
    c.dOel = (lds_cp)(dOs + 4 * h * ldt + r * 4);
    c.Kown = (lds_cp)(Ks + (wv * 32 + r) * ldt + h * 16);
    c.Vown = (lds_cp)(Vs + (wv * 32 + r) * ldt + h * 16);
    c.tb2 = (const LDS_AS f32x2*)((u0 & 1) ? tabN + CF::TABF + (u0 - 1) : tabN + u0);
    f16_bwd_p2<D>(c, (lds_cp)(Ks + (wv * 32 + r) * ldt), (lds_cp)(Vs + (wv * 32 + r) * ldt));
  }
  __syncthreads();   // every wave is done with Q as an operand: the wave's own query rows take dQ
  f16_store_tile<D>((lds_cp)(Qs + (wv * 32 + r) * ldt), dq, p.scale, h);   // d(q) = scale . d(q scale)
  __syncthreads();
  // dQ | dK | dV (in place of Q / K / V) -> global rows
  for (int idx = tid; idx < 256 * 3 * CPS; idx += 512) {
    const int row = idx / (3 * CPS), k3 = idx - row * (3 * CPS), sec = k3 / CPS, k = k3 - sec * CPS;
    const int64_t tok = win_token16(b, wr, wc, row, g);
    char* dst = reinterpret_cast<char*>(p.dqkv + tok * p.ldq + sec * CF::C + hd * D) + k * GRAN;
    *reinterpret_cast<CH*>(dst) = chunk_from_lds<CH>(smem + sec * CF::SEC + (size_t)row * ldt + k * GRAN);
  }
}

template <int D>
int launch_f16_fwd(const F16Args& p, hipStream_t st) {
  auto kern = wattn16_f32_fwd_kernel<D>;
  constexpr size_t smem = F16<D>::SMEM_FWD;
  static_assert(smem <= 160 * 1024, "LDS");
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  const int64_t nwin = (int64_t)p.g.B * p.g.nWh * p.g.nWw;
  hipLaunchKernelGGL(kern, dim3((unsigned)(6 * nwin)), dim3(512), smem, st, p);
  return rdst_launch_status("wattn16_f32_fwd");
}

template <int D>
int launch_f16_bwd(const F16Args& p, hipStream_t st) {
  auto kern = wattn16_f32_bwd_kernel<D>;
  constexpr size_t smem = F16<D>::SMEM_BWD;
  static_assert(smem <= 160 * 1024, "LDS");
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  const int64_t nwin = (int64_t)p.g.B * p.g.nWh * p.g.nWw;
  hipLaunchKernelGGL(kern, dim3((unsigned)(6 * nwin)), dim3(512), smem, st, p);
  return rdst_launch_status("wattn16_f32_bwd");
}

bool al(const void* a, int64_t lda_bytes, int gsz) { return (uintptr_t)a % gsz == 0 && lda_bytes % gsz == 0; }
int f16_align(int d) { return d == 10 ? 8 : d == 15 ? 4 : 16; }

}  // namespace

// fp32, ws 16, 6 heads of dim 10 / 15 / 20, no explicit mask, scale > 0; RDST_ENOTSUP otherwise
int wattn16_fwd_f32(const float* qkv, int64_t ld, const float* table, float* out, int64_t ldo, const WinGeom& g, float scale,
                    hipStream_t st) {
  if (g.ws != 16 || g.heads != 6 || g.mask || !(scale > 0.f) || g.C % 6) return RDST_ENOTSUP;
  if ((int64_t)g.B * g.nWh * g.nWw * 6 > 0x7fffffff) return RDST_ENOTSUP;
  const int d = g.C / 6;
  if (d != 10 && d != 15 && d != 20) return RDST_ENOTSUP;
  const int a = f16_align(d);
  if (!(al(qkv, ld * 4, a) && al(out, ldo * 4, a))) return RDST_ENOTSUP;
  F16Args p{};
  p.qkv = qkv; p.ld = ld; p.table = table; p.out = out; p.ldo = ldo; p.g = g; p.scale = scale;
  return d == 10 ? launch_f16_fwd<10>(p, st) : d == 15 ? launch_f16_fwd<15>(p, st) : launch_f16_fwd<20>(p, st);
}

// slab: [windows][6][961] partial d(table), one row per window (*nslab = windows)
int wattn16_bwd_f32(const float* qkv, int64_t ld, const float* table, const float* dout, int64_t ldd, float* dqkv, int64_t ldq,
                    float* slab, int slab_rows, const WinGeom& g, float scale, int* nslab, hipStream_t st) {
  if (g.ws != 16 || g.heads != 6 || g.mask || !(scale > 0.f) || g.C % 6) return RDST_ENOTSUP;
  const int64_t nwin = (int64_t)g.B * g.nWh * g.nWw;
  if (nwin * 6 > 0x7fffffff || slab_rows < nwin) return RDST_ENOTSUP;
  const int d = g.C / 6;
  if (d != 10 && d != 15 && d != 20) return RDST_ENOTSUP;
  const int a = f16_align(d);
  if (!(al(qkv, ld * 4, a) && al(dout, ldd * 4, a) && al(dqkv, ldq * 4, a))) return RDST_ENOTSUP;
  F16Args p{};
  p.qkv = qkv; p.ld = ld; p.table = table; p.dout = dout; p.ldd = ldd; p.dqkv = dqkv; p.ldq = ldq; p.slab = slab; p.g = g;
  p.scale = scale;
  *nslab = (int)nwin;
  return d == 10 ? launch_f16_bwd<10>(p, st) : d == 15 ? launch_f16_bwd<15>(p, st) : launch_f16_bwd<20>(p, st);
}
