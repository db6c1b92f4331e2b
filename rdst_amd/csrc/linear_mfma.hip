// K3 on the gfx950 matrix cores: Linear forward / dgrad / wgrad for the skinny GEMMs of the Swin
// blocks (M = B*H*W tokens, K and N in 30..360).
//
// Design (MI355X-first, not a tiled-GEMM port):
//  * forward / dgrad ("NT": both operands k-contiguous): PERSISTENT 8-wave workgroups, one per CU.
//    The whole weight matrix (<= 104 KB in bf16) is converted fp32 -> compute type once per workgroup
//    and stays resident in LDS; every wave then streams 32-row slabs of tokens: the slab is loaded
//    straight from HBM in MFMA-fragment shape (16 B per lane per k-step, rows only dword aligned),
//    LayerNorm (statistics via one cross-half shuffle) or GELU is applied in registers, the A
//    fragments stay in registers across all column tiles, and the 32x32 accumulators are stored with
//    bias / scale / residual (fwd) or activation-gradient / accumulate (dgrad) fused in.  HBM sees
//    each activation byte once; no LayerNorm output, GELU output or transposed copy ever exists.
//  * wgrad ("TN": contraction over tokens): each workgroup owns a contiguous range of tokens and all
//    of dW (<= 48 accumulator tiles over 8 waves); token stripes of 32 rows are staged row-major in
//    LDS and read TRANSPOSED by ds_read_b64_tr_b16 (bf16) or element-wise (fp32, 32x32x2 MFMA takes
//    single k elements).  A ones-column appended to f(X) makes d(bias) fall out of the same MFMAs.
//    Per-workgroup partials go to a slab and are summed in fixed order (deterministic).
#include "linear.h"
#include "mfma.h"
#include "reduce_batch.h"
#include <stdlib.h>

namespace {

bool mfma_disabled() {
  static int v = -1;
  if (v < 0) {
    const char* e = rdst_dbg_getenv("RDST_DISABLE_MFMA");
    v = (e && e[0] == '1') ? 1 : 0;
  }
  return v == 1;
}

constexpr int MODE_FWD = 0, MODE_DGRAD = 1;
constexpr float kLnEps = 1e-5f;

template <typename T>
struct LinArgs {
  const T* A; int64_t lda;            // rows being contracted: X (fwd) / dY (dgrad)
  const float* lnw; const float* lnb; int in_act;
  const float* Wt; int wK;            // nn.Linear weight (N_lin, K_lin), wK = K_lin
  const float* bias;
  const T* R; int64_t ldr;
  T* Y; int64_t ldy;                  // fwd: Y, dgrad (no LN): dX
  float* stats;                       // fwd + LN: (M,2) {mean, rstd}
  float* dA;                          // dgrad + LN: fp32 (M, Nout)
  const T* Xa; int64_t ldxa;          // dgrad: pre-activation X for act'
  const T* Acc; int64_t ldacc;        // dgrad: + dX_add
  int64_t M; int Kc; int Nout; float s;
  int Tn; int ldw; int nch; int aoff;
  int dbg;   // RDST_LIN_DEBUG ablation switches: 1 skip stores, 2 skip fused-add operand loads, 4 skip the column tiles
  unsigned long long* stamps;   // RDST_LIN_STAMPS=n (debug): [grid][16] s_memtime stamps of thread 0
};

// The accumulators are kept TRANSPOSED (D^T = W . A^T: output column n in the registers, token on the lane):
// a lane then owns 4 consecutive output columns of ITS OWN token row per register group, one
// v_permlane32_swap per register turns two groups into 8 consecutive columns, and the row goes out
// as 16-B stores straight from the registers — no LDS bounce, no second pass over the tile.  The
// residual / activation-gradient / accumulate operands are read with the same 16-B row chunks and
// added in fp32 before the single rounding.  The bias is the initial accumulator.  The next slab's
// fragments are prefetched while the current one is multiplied.
template <typename T, int TMAX, int MODE, bool SP = false>
__global__ void __launch_bounds__(512) lin_mfma_kernel(const LinArgs<T> p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using MM = Mma<T, SP>;   // SP: the split arithmetic of RDST_F32X3 (mfma.h) — the weights are split as they are staged, a slab's fragments once
  constexpr int KP = MM::KP, HP = MM::HP;
  constexpr bool BF = sizeof(T) == 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  const int Tn = p.Tn;
  char* Ws = smem;
  float* gam = reinterpret_cast<float*>(smem + (size_t)p.nch * p.ldw);
  float* bet = gam + Tn * KP;
  float* biasL = bet + Tn * KP;  // [npad] (forward)
  // wave-private staging tile for the coalesced slab loads: 32 rows x (128 B of the row + 16 B pad)
  constexpr int ABUF_LD = 144;
  char* abuf = smem + p.aoff + wave * (32 * ABUF_LD);
  const bool has_ln = (MODE == MODE_FWD) && p.lnw != nullptr;
  int nst = 0;
  auto stamp = [&]() {
    if (RDST_DBGV(p.stamps) && tid == 0 && nst < 16) p.stamps[(size_t)blockIdx.x * 16 + nst++] = __builtin_readcyclecounter();
  };
  stamp();  // 0: start
  const int npad = ((p.Nout + 31) / 32) * 32;
  // The small parameter vectors are LOADED here (one value per thread: Tn*KP <= 512, npad <= 512) but written to
  // LDS only after the weight staging has issued its loads: every dependent global-load wait in the prologue
  // costs ~2 us of a 25-50 us kernel, so all prologue loads go out before the first wait.
  float pre_g = 0.f, pre_b = 0.f, pre_bias = 0.f;
  if (has_ln && tid < p.Kc) {
    pre_g = p.lnw[tid];
    pre_b = p.lnb[tid];
  }
  if (MODE == MODE_FWD && p.bias && tid < p.Nout) pre_bias = p.bias[tid];
  const int64_t nslabs = (p.M + 31) / 32;
  const float invK = 1.0f / (float)p.Kc;
  // 16-B row chunks need dword-aligned rows: decided once per kernel, not per store
  auto rows_vec = [](const void* base, int64_t ld) { return (reinterpret_cast<uintptr_t>(base) & 3) == 0 && (ld * (int64_t)sizeof(T)) % 4 == 0; };
  const bool y_vec = rows_vec(p.Y, p.ldy), r_vec = rows_vec(p.R, p.ldr), xa_vec = rows_vec(p.Xa, p.ldxa), ac_vec = rows_vec(p.Acc, p.ldacc);

  // Slab loads.  A lane reading 16 B of ITS OWN row (fragment shape) makes every wave instruction touch
  // 32 different rows for 32 B each: measured 0.8 TB/s.  Instead rows are read COALESCED — 8 lanes x
  // 16 B = 128 B of one row, 8 rows per instruction — in 128-B column chunks, bounced through the
  // wave-private LDS tile and re-read as fragments (ds_read_b128, odd 16-B-slot stride).  The raw chunks
  // of the NEXT slab are prefetched while the current one is multiplied.  A row's last, shorter chunk is
  // read as the row's last 16 B (overlapping), so nothing outside the row is touched.
  constexpr int NKC = (TMAX + 3) / 4;                    // 128-B column chunks (4 k-steps each)
  const int rowbytes = p.Kc * (int)sizeof(T);
  const int crow = lane >> 3, cchk = lane & 7;           // row within a group of 8, 16-B chunk within the 128 B
  // elements past the row's end inside the last k-step must be zero (they meet zero weight columns, but may be NaN bits)
  auto zero_tail = [&](Pack16 (&a)[TMAX]) {
    const int klast = (Tn - 1) * KP + h * HP;
    if (klast + HP > p.Kc) {
#pragma unroll
      for (int t = 0; t < TMAX; ++t)
        if (t == Tn - 1) {
          float f[HP];
          MM::unpack(a[t], f);
#pragma unroll
          for (int e = 0; e < HP; ++e) f[e] = (klast + e < p.Kc) ? f[e] : 0.f;
          a[t] = MM::pack(f);
        }
    }
  };
  auto issue_raw = [&](Pack16 (&raw)[NKC][4], int64_t slab) {
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc) {
      int off = kc * 128 + cchk * 16;
      if (off + 16 > rowbytes) off = rowbytes - 16;      // the row's last chunk, or a lane past the row's end (re-reads it)
      if (off < 0) off = 0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int64_t row = slab * 32 + 8 * i + crow;
        row = row < p.M ? row : p.M - 1;
        const u32x4_a4 v = *reinterpret_cast<const u32x4_a4*>(reinterpret_cast<const char*>(p.A + row * p.lda) + off);
        raw[kc][i].w[0] = v.x; raw[kc][i].w[1] = v.y; raw[kc][i].w[2] = v.z; raw[kc][i].w[3] = v.w;
      }
    }
  };
  auto raw_to_frags = [&](const Pack16 (&raw)[NKC][4], Pack16 (&a)[TMAX]) {
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc) {
      if (kc * 128 < rowbytes) {
        int off = kc * 128 + cchk * 16;
        const bool act = off < rowbytes;
        if (off + 16 > rowbytes) off = rowbytes - 16;
        const int loc = off - kc * 128;                  // >= 0: a chunk's tail is at least 16 B (see `coal`)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (act) *reinterpret_cast<Pack16*>(abuf + (8 * i + crow) * ABUF_LD + loc) = raw[kc][i];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
          const int t = 4 * kc + tt;
          if (t < TMAX && t < Tn) a[t] = *reinterpret_cast<const Pack16*>(abuf + r * ABUF_LD + tt * 32 + h * 16);
        }
      }
    }
    zero_tail(a);
  };
  auto load_chunks = [&](Pack16 (&a)[TMAX], int64_t slab) {   // no prefetch: one 128-B column chunk at a time
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc) {
      if (kc * 128 < rowbytes) {
        int off = kc * 128 + cchk * 16;
        const bool act = off < rowbytes;
        if (off + 16 > rowbytes) off = rowbytes - 16;
        const int loc = off - kc * 128;
        Pack16 rw[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          int64_t row = slab * 32 + 8 * i + crow;
          row = row < p.M ? row : p.M - 1;
          const u32x4_a4 v = *reinterpret_cast<const u32x4_a4*>(reinterpret_cast<const char*>(p.A + row * p.lda) + off);
          rw[i].w[0] = v.x; rw[i].w[1] = v.y; rw[i].w[2] = v.z; rw[i].w[3] = v.w;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (act) *reinterpret_cast<Pack16*>(abuf + (8 * i + crow) * ABUF_LD + loc) = rw[i];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
          const int t = 4 * kc + tt;
          if (t < TMAX && t < Tn) a[t] = *reinterpret_cast<const Pack16*>(abuf + r * ABUF_LD + tt * 32 + h * 16);
        }
      }
    }
    zero_tail(a);
  };

  constexpr bool PFETCH = TMAX <= 8;   // the next slab's raw chunks in registers; beyond that they do not fit
  Pack16 a[TMAX], raw[PFETCH ? NKC : 1][4];
  const int64_t slab0 = (int64_t)blockIdx.x * 8 + wave, sstep = (int64_t)gridDim.x * 8;
  if constexpr (PFETCH) {
    if (slab0 < nslabs) issue_raw(raw, slab0);   // in flight while the weights are staged
  }
  for (int n0 = 0; n0 < p.Nout; n0 += p.nch) {
    __syncthreads();
    const int nc = (p.Nout - n0 < p.nch) ? p.Nout - n0 : p.nch;
    const int ncp = ((nc + 31) / 32) * 32;
    if (RDST_DBGV(p.dbg) & 16) {
    } else if (MODE == MODE_FWD) {
      // rows n of W (N,K), k contiguous -> packs of the LDS image.  Lean on purpose: thread -> (row, pack) by
      // shift / mask, 4 items of a thread in flight, 16-B loads (a 12-deep batch with integer
      // divisions and a scalar path per slot cost 5 us of instruction issue per launch, measured)
      int shp = 0;
      while ((1 << shp) < 2 * Tn) ++shp;
      const int total = ncp << shp;
      constexpr int U = 4;    // deeper batches do not help: the staging is bound by L2 bandwidth (256 workgroups each read all of W)
      for (int base = tid; base < total; base += 512 * U) {
        u32x4_a4 v[U][HP / 4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int idx = base + 512 * u;
          const int n = idx >> shp, ph = idx & ((1 << shp) - 1);
          const bool ok = idx < total && n < nc && ph < 2 * Tn;
          const float* src = p.Wt + (int64_t)(n0 + (ok ? n : 0)) * p.wK;
#pragma unroll
          for (int q = 0; q < HP / 4; ++q) {
            const int kq = ph * HP + 4 * q;
            if (ok && kq + 4 <= p.Kc) {
              v[u][q] = *reinterpret_cast<const u32x4_a4*>(src + kq);
            } else {
              v[u][q].x = (ok && kq < p.Kc) ? __float_as_uint(src[kq]) : 0u;
              v[u][q].y = (ok && kq + 1 < p.Kc) ? __float_as_uint(src[kq + 1]) : 0u;
              v[u][q].z = (ok && kq + 2 < p.Kc) ? __float_as_uint(src[kq + 2]) : 0u;
              v[u][q].w = 0u;
            }
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int idx = base + 512 * u;
          const int n = idx >> shp, ph = idx & ((1 << shp) - 1);
          // (a chunk that is not a whole number of column tiles is the "tight" single chunk of launch_lin: the rows past
          // it lie over gamma / beta / bias and the staging tiles and must not be written)
          if (idx < total && ph < 2 * Tn && (n < nc || (p.nch & 31) == 0)) {
            float f[HP];
#pragma unroll
            for (int q = 0; q < HP / 4; ++q) {
              f[4 * q] = __uint_as_float(v[u][q].x); f[4 * q + 1] = __uint_as_float(v[u][q].y);
              f[4 * q + 2] = __uint_as_float(v[u][q].z); f[4 * q + 3] = __uint_as_float(v[u][q].w);
            }
            *reinterpret_cast<Pack16*>(Ws + (size_t)n * p.ldw + ph * 16) = MM::pack_op(f);
          }
        }
      }
    } else {
      // dgrad: output column n = K_lin index, contraction k = N_lin index: W (N_lin, K_lin) read row-wise
      // (coalesced), scattered transposed into the LDS image [n][k]
      lds_zero16(Ws, ((p.nch & 31) ? nc : ncp) * p.ldw, tid, 512);   // (a tight chunk ends with its last row: see launch_lin)
      __syncthreads();
      stage_scatter<T>(p.Wt + n0, p.Kc, nc, (int64_t)p.wK, tid, 512, Ws, [&](int k, int n) { return n * p.ldw + k * (int)sizeof(T); });
      if constexpr (MM::SPLIT) {   // the scattered image holds floats: split it in place, pack by pack
        __syncthreads();
        const int rows = (p.nch & 31) ? nc : ncp, ppr = 2 * Tn;
        for (int i = tid; i < rows * ppr; i += 512) {
          const int n = i / ppr, ph = i - n * ppr;
          Pack16* q = reinterpret_cast<Pack16*>(Ws + (size_t)n * p.ldw + ph * 16);
          *q = MM::op(*q);
        }
      }
    }
    if (n0 == 0) {
      if (has_ln && tid < Tn * KP) { gam[tid] = pre_g; bet[tid] = pre_b; }
      if (MODE == MODE_FWD && tid < npad) biasL[tid] = pre_bias;
    }
    __syncthreads();
    stamp();  // 1: weights staged
    const int nct = ncp / 32;

    if constexpr (PFETCH) {
      if (n0 != 0 && slab0 < nslabs) issue_raw(raw, slab0);
    }
    for (int64_t slab = slab0; slab < ((RDST_DBGV(p.dbg) & 8) ? 0 : nslabs); slab += sstep) {
      if constexpr (PFETCH) {
        raw_to_frags(raw, a);
        // every iteration defines the whole prefetch set (past the end it re-reads this slab)
        issue_raw(raw, slab + sstep < nslabs ? slab + sstep : slab);
      } else {
        load_chunks(a, slab);
      }
      stamp();  // 2+: slab's fragments ready
      const int64_t row = slab * 32 + r;
      const bool valid = row < p.M;
      if (MODE == MODE_FWD) {
        if (has_ln) {
          // LayerNorm on the fragments: the row is spread over the two lane halves.  Elements past Kc are zero in
          // `a` (zero_tail), so the sums need no masks; the (x - mean)^2 sum is corrected for them analytically.
          const int klast = (Tn - 1) * KP + h * HP;
          const int npadl = klast + HP > p.Kc ? (klast + HP - p.Kc < HP ? klast + HP - p.Kc : HP) : 0;
          if constexpr (TMAX <= 8) {
            float f[TMAX][HP];   // unpacked once, kept across the three passes
            float sum = 0.f;
#pragma unroll
            for (int t = 0; t < TMAX; ++t)
              if (t < Tn) {
                MM::unpack(a[t], f[t]);
#pragma unroll
                for (int e = 0; e < HP; ++e) sum += f[t][e];
              }
            sum += __shfl_xor(sum, 32, 64);
            const float mean = sum * invK;
            float var = 0.f;
#pragma unroll
            for (int t = 0; t < TMAX; ++t)
              if (t < Tn) {
#pragma unroll
                for (int e = 0; e < HP; ++e) {
                  f[t][e] -= mean;
                  var = fmaf(f[t][e], f[t][e], var);
                }
              }
            var -= (float)npadl * mean * mean;
            var += __shfl_xor(var, 32, 64);
            const float rstd = 1.0f / sqrtf(var * invK + kLnEps);
            if (n0 == 0 && h == 0 && valid && p.stats) {
              p.stats[2 * row] = mean;
              p.stats[2 * row + 1] = rstd;
            }
#pragma unroll
            for (int t = 0; t < TMAX; ++t)
              if (t < Tn) {
                const int k0 = t * KP + h * HP;
                float gq[HP], bq[HP];
#pragma unroll
                for (int q = 0; q < HP / 4; ++q) {
                  const float4 g4 = *reinterpret_cast<const float4*>(gam + k0 + 4 * q);
                  const float4 b4 = *reinterpret_cast<const float4*>(bet + k0 + 4 * q);
                  gq[4 * q] = g4.x; gq[4 * q + 1] = g4.y; gq[4 * q + 2] = g4.z; gq[4 * q + 3] = g4.w;
                  bq[4 * q] = b4.x; bq[4 * q + 1] = b4.y; bq[4 * q + 2] = b4.z; bq[4 * q + 3] = b4.w;
                }
#pragma unroll
                for (int e = 0; e < HP; ++e) f[t][e] = fmaf(f[t][e] * rstd, gq[e], bq[e]);   // gamma = beta = 0 past Kc
                a[t] = MM::pack_op(f[t]);
              }
          } else {
            float sum = 0.f;
#pragma unroll
            for (int t = 0; t < TMAX; ++t)
              if (t < Tn) {
                float f[HP];
                MM::unpack(a[t], f);
#pragma unroll
                for (int e = 0; e < HP; ++e) sum += f[e];
              }
            sum += __shfl_xor(sum, 32, 64);
            const float mean = sum * invK;
            float var = 0.f;
#pragma unroll
            for (int t = 0; t < TMAX; ++t)
              if (t < Tn) {
                float f[HP];
                MM::unpack(a[t], f);
#pragma unroll
                for (int e = 0; e < HP; ++e) {
                  const float d = f[e] - mean;
                  var = fmaf(d, d, var);
                }
              }
            var -= (float)npadl * mean * mean;
            var += __shfl_xor(var, 32, 64);
            const float rstd = 1.0f / sqrtf(var * invK + kLnEps);
            if (n0 == 0 && h == 0 && valid && p.stats) {
              p.stats[2 * row] = mean;
              p.stats[2 * row + 1] = rstd;
            }
#pragma unroll
            for (int t = 0; t < TMAX; ++t)
              if (t < Tn) {
                float f[HP];
                MM::unpack(a[t], f);
                const int k0 = t * KP + h * HP;
#pragma unroll
                for (int e = 0; e < HP; ++e) f[e] = fmaf((f[e] - mean) * rstd, gam[k0 + e], bet[k0 + e]);
                a[t] = MM::pack_op(f);
              }
          }
        } else if (p.in_act) {
#pragma unroll
          for (int t = 0; t < TMAX; ++t)
            if (t < Tn) {
              float f[HP];
              MM::unpack(a[t], f);
#pragma unroll
              for (int e = 0; e < HP; ++e) f[e] = apply_act<BF || SP>(f[e], p.in_act);   // (split mode: the fast GELU, |err| 1.5e-7)
              a[t] = MM::pack_op(f);
            }
        }
      }
      if constexpr (MM::SPLIT) {   // (the LayerNorm / activation passes above leave operands already)
        if (!(MODE == MODE_FWD && (has_ln || p.in_act))) {
#pragma unroll
          for (int t = 0; t < TMAX; ++t)
            if (t < Tn) a[t] = MM::op(a[t]);
        }
      }
      for (int ct = 0; ct < ((RDST_DBGV(p.dbg) & 4) ? 0 : nct); ++ct) {
        f32x16 acc;
        if (MODE == MODE_FWD) {   // bias = initial accumulator: register group g holds columns 8g + 4h .. +3
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const float4 bq = *reinterpret_cast<const float4*>(biasL + n0 + ct * 32 + 8 * g4 + 4 * h);
            acc[4 * g4] = bq.x; acc[4 * g4 + 1] = bq.y; acc[4 * g4 + 2] = bq.z; acc[4 * g4 + 3] = bq.w;
          }
        } else {
#pragma unroll
          for (int v = 0; v < 16; ++v) acc[v] = 0.f;
        }
        const char* wrow = Ws + (size_t)(ct * 32 + r) * p.ldw + h * 16;
#pragma unroll
        for (int t = 0; t < TMAX; ++t)
          if (t < Tn) {
            const Pack16 wa = *reinterpret_cast<const Pack16*>(wrow + t * 32);
            if constexpr (TMAX > 16) MM::mma_da(acc, wa, a[t]);   // (split mode: the duplicated halves of 32 fragments would not fit)
            else MM::mma(acc, wa, a[t]);   // rows = output columns, cols = tokens
          }
        // epilogue: two pairs of register groups -> two runs of 8 consecutive columns of the lane's row
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
          float c8[8];
          int cb;   // first of the lane's 8 columns (bf16) / of each 4-column run (fp32)
          if (BF) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[8 * gp + e]),
                                                               __float_as_uint(acc[8 * gp + 4 + e]), false, false);
              c8[e] = __uint_as_float(sw[0]);
              c8[4 + e] = __uint_as_float(sw[1]);
            }
            if (p.s != 1.0f) {
#pragma unroll
              for (int e = 0; e < 8; ++e) c8[e] *= p.s;
            }
            cb = n0 + ct * 32 + 8 * (2 * gp + h);
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) c8[e] = acc[8 * gp + e] * p.s;
            cb = n0 + ct * 32 + 16 * gp + 4 * h;   // c8[0..3] at cb, c8[4..7] at cb + 8
          }
          if (!valid) continue;
          // operands of the fused adds: 8 columns as one (bf16) or two (fp32) 16-B row chunks
          auto chunk8 = [&](const T* rowp, bool vec, float (&g8)[8]) {
            if (BF) {
              if (vec && cb + 8 <= p.Nout) {
                const u32x4_a4 u = *reinterpret_cast<const u32x4_a4*>(rowp + cb);
                g8[0] = bf16lo(u.x); g8[1] = bf16hi(u.x); g8[2] = bf16lo(u.y); g8[3] = bf16hi(u.y);
                g8[4] = bf16lo(u.z); g8[5] = bf16hi(u.z); g8[6] = bf16lo(u.w); g8[7] = bf16hi(u.w);
              } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) g8[e] = cb + e < p.Nout ? to_f32<T>(rowp[cb + e]) : 0.f;
              }
            } else {
              // fp32: the lane's two runs of 4 columns as 16-byte chunks (a lane reading its own row element by element made
              // every load instruction touch 64 cache lines for 4 bytes each)
#pragma unroll
              for (int q4 = 0; q4 < 2; ++q4) {
                const int c = cb + 8 * q4;
                if (vec && c + 4 <= p.Nout) {
                  const u32x4_a4 u = *reinterpret_cast<const u32x4_a4*>(rowp + c);
                  g8[4 * q4] = __uint_as_float(u.x); g8[4 * q4 + 1] = __uint_as_float(u.y);
                  g8[4 * q4 + 2] = __uint_as_float(u.z); g8[4 * q4 + 3] = __uint_as_float(u.w);
                } else {
#pragma unroll
                  for (int e = 0; e < 4; ++e) g8[4 * q4 + e] = c + e < p.Nout ? to_f32<T>(rowp[c + e]) : 0.f;
                }
              }
            }
          };
          if (MODE == MODE_FWD) {
            if (p.R && !(RDST_DBGV(p.dbg) & 2)) {
              float g8[8];
              chunk8(p.R + row * p.ldr, r_vec, g8);
#pragma unroll
              for (int e = 0; e < 8; ++e) c8[e] += g8[e];
            }
          } else if (!p.dA) {
            if (p.Xa && p.in_act) {
              float g8[8];
              chunk8(p.Xa + row * p.ldxa, xa_vec, g8);
#pragma unroll
              for (int e = 0; e < 8; ++e) c8[e] *= act_grad<BF || SP>(g8[e], p.in_act);
            }
            if (p.Acc) {
              float g8[8];
              chunk8(p.Acc + row * p.ldacc, ac_vec, g8);
#pragma unroll
              for (int e = 0; e < 8; ++e) c8[e] += g8[e];
            }
          }
          if (MODE == MODE_DGRAD && p.dA) {  // fp32 destination (dgrad in front of a LayerNorm, generic path)
            float* dst = p.dA + row * p.Nout;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const int c = BF ? cb + e : cb + (e & 3) + 8 * (e >> 2);
              if (c < p.Nout) dst[c] = c8[e];
            }
            continue;
          }
          if (RDST_DBGV(p.dbg) & 1) continue;
          T* yrow = p.Y + row * p.ldy;
          if (BF) {
            if (y_vec && cb + 8 <= p.Nout) {
              u32x4_a4 u;
              u.x = pack_bf16x2(c8[0], c8[1]); u.y = pack_bf16x2(c8[2], c8[3]);
              u.z = pack_bf16x2(c8[4], c8[5]); u.w = pack_bf16x2(c8[6], c8[7]);
              *reinterpret_cast<u32x4_a4*>(yrow + cb) = u;
            } else {
#pragma unroll
              for (int e = 0; e < 8; ++e) if (cb + e < p.Nout) yrow[cb + e] = from_f32<T>(c8[e]);
            }
          } else {
#pragma unroll
            for (int q4 = 0; q4 < 2; ++q4) {
              const int c = cb + 8 * q4;
              if (c + 4 <= p.Nout && y_vec) {   // (16-byte stores of dword-aligned rows)
                u32x4_a4 u;
                u.x = __float_as_uint(c8[4 * q4]); u.y = __float_as_uint(c8[4 * q4 + 1]);
                u.z = __float_as_uint(c8[4 * q4 + 2]); u.w = __float_as_uint(c8[4 * q4 + 3]);
                *reinterpret_cast<u32x4_a4*>(yrow + c) = u;
              } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (c + e < p.Nout) yrow[c + e] = from_f32<T>(c8[4 * q4 + e]);
              }
            }
          }
        }
      }
    }
  }
}

// ---- dgrad of a Linear that sits behind a LayerNorm, with the LayerNorm backward fused in ----------
// dA = s * dY @ W  (32 x K tile set per slab, K <= 128) is never written: per slab the column tiles are
// produced TWICE (the MFMAs are cheap, the kernel is HBM-bound): pass 1 bounces each tile through the
// wave's LDS buffer and accumulates, row-major, s1 = sum_k dA*gamma, s2 = sum_k dA*gamma*xhat and the
// per-column d(gamma) / d(beta) partials; pass 2 rebuilds the tile and stores
//   dX = rstd * (gamma*dA - s1/K - xhat*s2/K)  (+ dX)
// with 8-16-B row stores.  d(gamma), d(beta): one [2][K] slab row per workgroup, reduced in fixed order.
template <typename T>
struct LnDgradArgs {
  const T* dY; int64_t lddy;      // (M, N)
  const float* Wt; int N, K;      // nn.Linear weight (N, K)
  const T* X; int64_t ldx; const float* stats; const float* gamma;
  T* dX; int64_t lddx; const T* Acc; int64_t ldacc;   // dX = Acc + ...
  float* slab;                    // [grid][2][K]
  int64_t M; float s;
  int Tn, ldw;
};

template <typename T, int TMAX>
__global__ void __launch_bounds__(512) lin_dgrad_ln_kernel(const LnDgradArgs<T> p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using MM = Mma<T>;
  constexpr int KP = MM::KP, HP = MM::HP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  const int Tn = p.Tn, K = p.K;
  const int kpad = ((K + 31) / 32) * 32, nct = kpad / 32;
  char* Ws = smem;                                                      // [kpad][ldw]: W^T rows (k), n-contiguous
  float* eps = reinterpret_cast<float*>(smem + (size_t)kpad * p.ldw) + wave * 1024;
  float* red = reinterpret_cast<float*>(smem + (size_t)kpad * p.ldw) + 8 * 1024;  // [8 waves][2][128]
  stage_packs_batched<T, 4>(kpad * 2 * Tn, p.N, p.K, tid, 512, [&](int idx, const float*& src, int& k0, char*& dst, bool& ok) {
    const int k = idx / (2 * Tn), ph = idx - k * (2 * Tn);
    ok = k < K;
    k0 = ph * HP;                                  // contraction index n: element n at Wt[n*K + k]
    src = p.Wt + k + (int64_t)k0 * p.K;
    dst = Ws + (size_t)k * p.ldw + ph * 16;
  });
  __syncthreads();
  const int chunk = lane & 7;
  float gam[4][4], dg[4][4], db[4][4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int col = c * 32 + chunk * 4 + q;
      gam[c][q] = col < K ? p.gamma[col] : 0.f;
      dg[c][q] = 0.f;
      db[c][q] = 0.f;
    }
  const float invK = 1.0f / (float)K;
  const int64_t nslabs = (p.M + 31) / 32;
  for (int64_t slab = (int64_t)blockIdx.x * 8 + wave; slab < nslabs; slab += (int64_t)gridDim.x * 8) {
    const int64_t row = slab * 32 + r;
    const bool valid = row < p.M;
    const T* arow = p.dY + (valid ? row : 0) * p.lddy;
    Pack16 a[TMAX];
#pragma unroll
    for (int t = 0; t < TMAX; ++t)
      if (t < Tn) a[t] = load_pack<T>(arow, t * KP + h * HP, p.N, valid);
    float s1[4], s2[4], mean[4], rstd[4];
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int64_t rr = slab * 32 + (lane >> 3) + 8 * ps;
      s1[ps] = 0.f; s2[ps] = 0.f;
      mean[ps] = rr < p.M ? p.stats[2 * rr] : 0.f;
      rstd[ps] = rr < p.M ? p.stats[2 * rr + 1] : 0.f;
    }
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (c < nct) {
          f32x16 acc;
#pragma unroll
          for (int v = 0; v < 16; ++v) acc[v] = 0.f;
          const char* wrow = Ws + (size_t)(c * 32 + r) * p.ldw + h * 16;
#pragma unroll
          for (int t = 0; t < TMAX; ++t)
            if (t < Tn) {
              const Pack16 b = *reinterpret_cast<const Pack16*>(wrow + t * 32);
              MM::mma(acc, a[t], b);
            }
#pragma unroll
          for (int v = 0; v < 16; ++v) eps[acc_row(v, h) * 32 + r] = acc[v] * p.s;
          __builtin_amdgcn_wave_barrier();
          const int col = c * 32 + chunk * 4;
#pragma unroll
          for (int ps = 0; ps < 4; ++ps) {
            const int rowl = (lane >> 3) + 8 * ps;
            const int64_t rr = slab * 32 + rowl;
            const float4 d4 = *reinterpret_cast<const float4*>(eps + rowl * 32 + chunk * 4);
            if (rr < p.M && col < K) {
              const float da[4] = {d4.x, d4.y, d4.z, d4.w};
              float xh[4];
              const T* xp = p.X + rr * p.ldx + col;
              if (col + 4 <= K && (reinterpret_cast<uintptr_t>(xp) & 3) == 0) {
                if (sizeof(T) == 2) {
                  const u32x2_a4 u = *reinterpret_cast<const u32x2_a4*>(xp);
                  xh[0] = bf16lo(u.x); xh[1] = bf16hi(u.x); xh[2] = bf16lo(u.y); xh[3] = bf16hi(u.y);
                } else {
                  const u32x4_a4 u = *reinterpret_cast<const u32x4_a4*>(xp);
                  xh[0] = __uint_as_float(u.x); xh[1] = __uint_as_float(u.y); xh[2] = __uint_as_float(u.z); xh[3] = __uint_as_float(u.w);
                }
              } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) xh[q] = col + q < K ? to_f32<T>(xp[q]) : 0.f;
              }
#pragma unroll
              for (int q = 0; q < 4; ++q) xh[q] = (col + q < K) ? (xh[q] - mean[ps]) * rstd[ps] : 0.f;
              if (pass == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                  const float gq = da[q] * gam[c][q];
                  s1[ps] += gq;
                  s2[ps] = fmaf(gq, xh[q], s2[ps]);
                  dg[c][q] = fmaf(da[q], xh[q], dg[c][q]);
                  db[c][q] += da[q];
                }
              } else {
                float f[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) f[q] = rstd[ps] * (da[q] * gam[c][q] - s1[ps] - xh[q] * s2[ps]);
                T* dst = p.dX + rr * p.lddx + col;
                const bool vec = col + 4 <= K && (reinterpret_cast<uintptr_t>(dst) & 3) == 0;
                if (p.Acc) {
                  const T* ap = p.Acc + rr * p.ldacc + col;
                  if (col + 4 <= K && sizeof(T) == 2 && (reinterpret_cast<uintptr_t>(ap) & 3) == 0) {
                    const u32x2_a4 u = *reinterpret_cast<const u32x2_a4*>(ap);
                    f[0] += bf16lo(u.x); f[1] += bf16hi(u.x); f[2] += bf16lo(u.y); f[3] += bf16hi(u.y);
                  } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) if (col + q < K) f[q] += to_f32<T>(ap[q]);
                  }
                }
                if (vec && sizeof(T) == 2) {
                  u32x2_a4 u;
                  u.x = pack_bf16x2(f[0], f[1]); u.y = pack_bf16x2(f[2], f[3]);
                  *reinterpret_cast<u32x2_a4*>(dst) = u;
                } else if (vec) {
                  u32x4_a4 u;
                  u.x = __float_as_uint(f[0]); u.y = __float_as_uint(f[1]); u.z = __float_as_uint(f[2]); u.w = __float_as_uint(f[3]);
                  *reinterpret_cast<u32x4_a4*>(dst) = u;
                } else {
#pragma unroll
                  for (int q = 0; q < 4; ++q) if (col + q < K) dst[q] = from_f32<T>(f[q]);
                }
              }
            }
          }
          __builtin_amdgcn_wave_barrier();
        }
      if (pass == 0) {
        // finish the row sums: the 8 lanes that share a row differ in lane bits 0..2
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
#pragma unroll
          for (int o = 1; o < 8; o <<= 1) {
            s1[ps] += __shfl_xor(s1[ps], o, 64);
            s2[ps] += __shfl_xor(s2[ps], o, 64);
          }
          s1[ps] *= invK;
          s2[ps] *= invK;
        }
      }
    }
  }
  // d(gamma), d(beta): lanes with equal `chunk` own the same columns -> reduce over lane bits 3..5, then waves
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int o = 8; o < 64; o <<= 1) {
        dg[c][q] += __shfl_xor(dg[c][q], o, 64);
        db[c][q] += __shfl_xor(db[c][q], o, 64);
      }
      if (lane < 8) {
        red[(wave * 2 + 0) * 128 + c * 32 + chunk * 4 + q] = dg[c][q];
        red[(wave * 2 + 1) * 128 + c * 32 + chunk * 4 + q] = db[c][q];
      }
    }
  __syncthreads();
  for (int i = tid; i < 2 * K; i += 512) {
    const int which = i / K, k = i - which * K;
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) a += red[(w * 2 + which) * 128 + k];
    p.slab[(int64_t)blockIdx.x * 2 * K + i] = a;
  }
}

template <typename T, int MODE>
int launch_lin(LinArgs<T>& p, hipStream_t st, const char* what) {
  using MM = Mma<T>;
  p.Tn = (p.Kc + MM::KP - 1) / MM::KP;
  if (p.Tn > 32) return RDST_ENOTSUP;
  { const char* e = rdst_dbg_getenv("RDST_LIN_DEBUG"); p.dbg = e ? atoi(e) : 0; }
  {  // the slab loads move whole 16-B chunks of dword-aligned rows (a chunk's tail is never shorter than 16 B)
    const int64_t rowbytes = (int64_t)p.Kc * (int64_t)sizeof(T);
    const bool coal = ((uintptr_t)p.A & 3) == 0 && (p.lda * sizeof(T)) % 4 == 0 && rowbytes % 4 == 0 && rowbytes >= 16 &&
                      (rowbytes % 128 == 0 || rowbytes % 128 >= 16);
    if (!coal) return RDST_ENOTSUP;
  }
  p.ldw = lds_row_bytes(p.Kc, sizeof(T));
  const int npad = ((p.Nout + 31) / 32) * 32;
  const int abuf_bytes = 8 * 32 * 144;         // wave-private staging tiles of the coalesced slab loads
  int nch = ((112 * 1024) / p.ldw) / 32 * 32;  // weights resident in LDS (+ staging, gamma/beta/bias) under 160 KB
  if (nch < 32) return RDST_ENOTSUP;
  if (nch > npad) nch = npad;
  if (nch < npad) {
    // One chunk after all if the Nout rows themselves fit: the rows Nout .. npad-1 of the last column tile then lie over
    // gamma / beta / bias and the staging tiles (not staged; read as garbage by the MFMAs:
    // they only feed output columns >= Nout, which are never stored).  A second chunk re-reads, re-normalises and
    // re-activates every slab (fp32 fc2 at C = 120, 240 x 120 weights: chunks of 96 + 24 columns, 288 -> 1xx us).
    const size_t need = (size_t)p.Nout * p.ldw + (size_t)(2 * p.Tn * MM::KP + npad) * sizeof(float) + 16 + abuf_bytes;
    if (need <= 160 * 1024 && (size_t)npad * p.ldw <= need) nch = p.Nout;
  }
  p.nch = nch;
  if (p.Tn * MM::KP > 512 || npad > 512) return RDST_ENOTSUP;   // one parameter value per thread in the prologue
  p.aoff = (int)(((size_t)nch * p.ldw + (size_t)(2 * p.Tn * MM::KP + npad) * sizeof(float) + 15) / 16 * 16);
  const size_t smem = (size_t)p.aoff + abuf_bytes;
  if (smem > 160 * 1024) return RDST_ENOTSUP;
  const int64_t nslabs = (p.M + 31) / 32;
  int64_t grid = (nslabs + 7) / 8;
  int64_t cap = 256;  // persistent: one 8-wave workgroup per CU (the kernel's register count admits no second one)
  { const char* e = rdst_dbg_getenv("RDST_LIN_GRID"); if (e && atoi(e) > 0) cap = atoi(e); }
  if (grid > cap) grid = cap;
  static int want_stamps = -1;
  if (want_stamps < 0) { const char* e = rdst_dbg_getenv("RDST_LIN_STAMPS"); want_stamps = e ? atoi(e) : 0; }
  unsigned long long* hst = nullptr;
  if (want_stamps > 0) {
    (void)hipMalloc((void**)&p.stamps, (size_t)grid * 16 * 8);
    (void)hipMemsetAsync(p.stamps, 0, (size_t)grid * 16 * 8, st);
  }
#define RDST_LIN_LAUNCH(TM)                                                                                          \
  {                                                                                                                  \
    auto kern = lin_mfma_kernel<T, TM, MODE>;                                                                        \
    if constexpr (sizeof(T) == 4)                                                                                    \
      if (rdst_split()) kern = lin_mfma_kernel<T, TM, MODE, true>;                                                   \
    if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), smem, st, p);                                          \
  }
  if (p.Tn <= 8) RDST_LIN_LAUNCH(8) else if (p.Tn <= 16) RDST_LIN_LAUNCH(16) else RDST_LIN_LAUNCH(32)
#undef RDST_LIN_LAUNCH
  if (want_stamps > 0) {
    (void)hipStreamSynchronize(st);
    hst = (unsigned long long*)malloc((size_t)grid * 16 * 8);
    (void)hipMemcpy(hst, p.stamps, (size_t)grid * 16 * 8, hipMemcpyDeviceToHost);
    (void)hipFree(p.stamps);
    if (--want_stamps == 0) {
      double sum[16] = {0}; int cnt[16] = {0};
      for (int64_t w = 0; w < grid; ++w)
        for (int k = 1; k < 16; ++k) {
          if (!hst[w * 16 + k]) continue;
          sum[k] += (double)(hst[w * 16 + k] - hst[w * 16 + k - 1]); cnt[k]++;
        }
      fprintf(stderr, "[lin stamps %s Kc=%d Nout=%d grid=%lld] mean ticks between consecutive stamps\n", what, p.Kc, p.Nout, (long long)grid);
      for (int k = 1; k < 16; ++k) if (cnt[k]) fprintf(stderr, "  %2d: %9.0f (n=%d)\n", k, sum[k] / cnt[k], cnt[k]);
    }
    free(hst);
  }
  return rdst_launch_status(what);
}

// ---- dgrad + LayerNorm backward, register-resident form -------------------------------------------
// Used when d(gamma) / d(beta) come from the weight-gradient side (linear_wgrad_ln_mfma), so this kernel
// only produces dX.  dA^T = W^T-tiles . dY^T is kept TRANSPOSED (LayerNorm channel k in the registers,
// token on the lane): all NCT <= 4 channel tiles of a 32-token slab stay in the accumulators, one
// v_permlane32_swap per register turns them into runs of 8 consecutive channels of the lane's own row,
// the row sums s1 = sum_k dA gamma, s2 = sum_k dA gamma xhat are in-lane sums plus one half swap, and
//   dX = rstd * (gamma dA - s1/K - xhat s2/K) (+ dX_add)
// leaves as 16-B row stores.  No LDS bounce, no second pass over the MFMAs.  dY streams in coalesced
// 128-B column chunks through the wave's LDS tile (next chunk prefetched while this one is multiplied).
template <typename T, int NCT, bool SP = false>
__global__ void __launch_bounds__(512) lin_dgrad_ln2_kernel(const LnDgradArgs<T> p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using MM = Mma<T, SP>;
  constexpr int KP = MM::KP, HP = MM::HP;
  constexpr bool BF = sizeof(T) == 2;
  constexpr int ABUF_LD = 144;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  const int Tn = p.Tn, K = p.K;
  constexpr int kpad = NCT * 32;
  char* Ws = smem;                                                       // [kpad][ldw]: W^T rows (k), n-contiguous packs
  float* gamL = reinterpret_cast<float*>(smem + (size_t)kpad * p.ldw);   // [kpad]
  char* abuf = reinterpret_cast<char*>(gamL + kpad) + wave * (32 * ABUF_LD);
  const int rowbytes = p.N * (int)sizeof(T);
  const int nkc = (rowbytes + 127) / 128;
  const int crow = lane >> 3, cchk = lane & 7;
  const int64_t nslabs = (p.M + 31) / 32;
  const int64_t slab0 = (int64_t)blockIdx.x * 8 + wave, sstep = (int64_t)gridDim.x * 8;

  auto issue_chunk = [&](Pack16 (&rw)[4], int64_t slab, int kc) {
    int off = kc * 128 + cchk * 16;
    if (off + 16 > rowbytes) off = rowbytes - 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int64_t row = slab * 32 + 8 * i + crow;
      row = row < p.M ? row : p.M - 1;
      const u32x4_a4 v = *reinterpret_cast<const u32x4_a4*>(reinterpret_cast<const char*>(p.dY + row * p.lddy) + off);
      rw[i].w[0] = v.x; rw[i].w[1] = v.y; rw[i].w[2] = v.z; rw[i].w[3] = v.w;
    }
  };
  Pack16 raw[4];
  if (slab0 < nslabs) issue_chunk(raw, slab0, 0);   // in flight while the weights are staged
  lds_zero16(Ws, kpad * p.ldw, tid, 512);
  __syncthreads();
  // W (N, K) read row-wise (coalesced), scattered transposed: LDS row = LayerNorm channel k, columns n contiguous
  stage_scatter<T>(p.Wt, p.N, K, (int64_t)K, tid, 512, Ws, [&](int n, int k) { return k * p.ldw + n * (int)sizeof(T); });
  for (int i = tid; i < kpad; i += 512) gamL[i] = i < K ? p.gamma[i] : 0.f;
  __syncthreads();
  if constexpr (MM::SPLIT) {   // the scattered image holds floats: split it in place, pack by pack
    const int ppr = 2 * Tn;
    for (int i = tid; i < kpad * ppr; i += 512) {
      const int k = i / ppr, ph = i - k * ppr;
      Pack16* q = reinterpret_cast<Pack16*>(Ws + (size_t)k * p.ldw + ph * 16);
      *q = MM::op(*q);
    }
    __syncthreads();
  }
  const float invK = 1.0f / (float)K;

  for (int64_t slab = slab0; slab < nslabs; slab += sstep) {
    f32x16 acc[NCT];
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[c][v] = 0.f;
    for (int kc = 0; kc < nkc; ++kc) {
      // raw chunk kc -> wave tile -> fragments; the next chunk (or the next slab's first) goes in flight
      int off = kc * 128 + cchk * 16;
      const bool act = off < rowbytes;
      if (off + 16 > rowbytes) off = rowbytes - 16;
      const int loc = off - kc * 128;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (act) *reinterpret_cast<Pack16*>(abuf + (8 * i + crow) * ABUF_LD + loc) = raw[i];
      {
        const bool last = kc + 1 == nkc;
        const int64_t ns = last ? (slab + sstep < nslabs ? slab + sstep : slab) : slab;
        issue_chunk(raw, ns, last ? 0 : kc + 1);   // always (re)defines the whole prefetch set
      }
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        const int t = 4 * kc + tt;
        if (t < Tn) {
          Pack16 a = *reinterpret_cast<const Pack16*>(abuf + r * ABUF_LD + tt * 32 + h * 16);
          if (t == Tn - 1) {   // elements past the row's end inside the last k-step: zero (may be NaN bits)
            const int kl = t * KP + h * HP;
            if (kl + HP > p.N) {
              float f[HP];
              MM::unpack(a, f);
#pragma unroll
              for (int e = 0; e < HP; ++e) f[e] = (kl + e < p.N) ? f[e] : 0.f;
              a = MM::pack(f);
            }
          }
          a = MM::op(a);
#pragma unroll
          for (int c = 0; c < NCT; ++c) {
            const Pack16 wa = *reinterpret_cast<const Pack16*>(Ws + (size_t)(c * 32 + r) * p.ldw + t * 32 + h * 16);
            MM::mma(acc[c], wa, a);   // rows = LayerNorm channels, cols = tokens
          }
        }
      }
    }
    const int64_t row = slab * 32 + r;
    const bool valid = row < p.M;
    const int64_t rs_ = valid ? row : p.M - 1;
    const float2 st = *reinterpret_cast<const float2*>(p.stats + 2 * rs_);
    const float mean = st.x, rstd = st.y;
    const T* xrow = p.X + rs_ * p.ldx;
    // 8-channel runs of the lane's row: run (c, gp) starts at channel c*32 + 16*gp + 8*h (bf16) /
    // two 4-channel runs at c*32 + 16*gp + 4*h and + 8 (fp32)
    auto col_of = [&](int c, int gp, int e) { return BF ? c * 32 + 16 * gp + 8 * h + e : c * 32 + 16 * gp + 4 * h + (e & 3) + 8 * (e >> 2); };
    auto load8 = [&](const T* rowp, int c, int gp, float (&g8)[8]) {
      const int cb = col_of(c, gp, 0);
      if (BF && cb + 8 <= K && (reinterpret_cast<uintptr_t>(rowp + cb) & 3) == 0) {
        const u32x4_a4 u = *reinterpret_cast<const u32x4_a4*>(rowp + cb);
        g8[0] = bf16lo(u.x); g8[1] = bf16hi(u.x); g8[2] = bf16lo(u.y); g8[3] = bf16hi(u.y);
        g8[4] = bf16lo(u.z); g8[5] = bf16hi(u.z); g8[6] = bf16lo(u.w); g8[7] = bf16hi(u.w);
      } else if (!BF && cb + 12 <= K && (reinterpret_cast<uintptr_t>(rowp) & 3) == 0) {
        // fp32: the two 4-channel runs of the lane as 16-byte chunks of its dword-aligned row (element by element every load
        // instruction touched 64 cache lines for 4 bytes each)
#pragma unroll
        for (int q4 = 0; q4 < 2; ++q4) {
          const u32x4_a4 u = *reinterpret_cast<const u32x4_a4*>(rowp + cb + 8 * q4);
          g8[4 * q4] = __uint_as_float(u.x); g8[4 * q4 + 1] = __uint_as_float(u.y);
          g8[4 * q4 + 2] = __uint_as_float(u.z); g8[4 * q4 + 3] = __uint_as_float(u.w);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int cc = col_of(c, gp, e);
          g8[e] = cc < K ? to_f32<T>(rowp[cc]) : 0.f;
        }
      }
    };
    float G[NCT][2][8];   // gamma * dA in run layout; x-hat is rebuilt from a second (cache-hit) read of the row
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        float c8[8];
        if (BF) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[c][8 * gp + e]),
                                                             __float_as_uint(acc[c][8 * gp + 4 + e]), false, false);
            c8[e] = __uint_as_float(sw[0]);
            c8[4 + e] = __uint_as_float(sw[1]);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) c8[e] = acc[c][8 * gp + e];
        }
        float x8[8];
        load8(xrow, c, gp, x8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int cc = col_of(c, gp, e);
          const float g = c8[e] * p.s * gamL[cc];           // gamma = 0 past K
          const float xh = cc < K ? (x8[e] - mean) * rstd : 0.f;
          G[c][gp][e] = g;
          s1 += g;
          s2 = fmaf(g, xh, s2);
        }
      }
    {
      const auto a1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(s1), __float_as_uint(s1), false, false);
      s1 = (__uint_as_float(a1[0]) + __uint_as_float(a1[1])) * invK;
      const auto a2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(s2), __float_as_uint(s2), false, false);
      s2 = (__uint_as_float(a2[0]) + __uint_as_float(a2[1])) * invK;
    }
    if (!valid) continue;
    T* drow = p.dX + row * p.lddx;
#pragma unroll
    for (int c = 0; c < NCT; ++c)
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        const int cb = col_of(c, gp, 0);
        if (cb >= K) continue;
        float o8[8], x8[8];
        load8(xrow, c, gp, x8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xh = (x8[e] - mean) * rstd;   // channels past K are never stored
          o8[e] = rstd * (G[c][gp][e] - s1 - xh * s2);
        }
        if (p.Acc) {
          float a8[8];
          load8(p.Acc + row * p.ldacc, c, gp, a8);
#pragma unroll
          for (int e = 0; e < 8; ++e) o8[e] += a8[e];
        }
        if (BF && cb + 8 <= K && (reinterpret_cast<uintptr_t>(drow + cb) & 3) == 0) {
          u32x4_a4 u;
          u.x = pack_bf16x2(o8[0], o8[1]); u.y = pack_bf16x2(o8[2], o8[3]);
          u.z = pack_bf16x2(o8[4], o8[5]); u.w = pack_bf16x2(o8[6], o8[7]);
          *reinterpret_cast<u32x4_a4*>(drow + cb) = u;
        } else if (!BF && cb + 12 <= K && (reinterpret_cast<uintptr_t>(drow) & 3) == 0) {
#pragma unroll
          for (int q4 = 0; q4 < 2; ++q4) {
            u32x4_a4 u;
            u.x = __float_as_uint(o8[4 * q4]); u.y = __float_as_uint(o8[4 * q4 + 1]);
            u.z = __float_as_uint(o8[4 * q4 + 2]); u.w = __float_as_uint(o8[4 * q4 + 3]);
            *reinterpret_cast<u32x4_a4*>(drow + cb + 8 * q4) = u;
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int cc = col_of(c, gp, e);
            if (cc < K) drow[cc] = from_f32<T>(o8[e]);
          }
        }
      }
  }
}

template <typename T> bool rows_ok(const void*, int64_t) { return true; }  // load_pack checks alignment per access

// ------------------------------------------------------------------------------------------------
// wgrad
// ------------------------------------------------------------------------------------------------
constexpr int WG_WAVES = 8;     // waves per workgroup, one workgroup per CU: 2 waves per SIMD, 256 VGPRs each
constexpr int WG_THREADS = 64 * WG_WAVES;
constexpr int WG_MAXT = 6;      // accumulator tiles per wave (8 waves -> 48 tiles = 360 x 121 max)
constexpr int WG_STRIPE = 32;   // token rows staged per step

template <typename T>
struct WgradArgs {
  const T* X; int64_t ldx; const float* lnw; const float* lnb; const float* stats; int in_act;
  const T* dY; int64_t lddy;
  float* slab;
  int64_t M; int K; int N; int Kx;
  int64_t rows_per_wg;
  int ldn, ldk;  // LDS row strides, bytes
  int NT, KT;
};

// The whole kernel is a memory pipeline: 2*M*(K+N)*elt bytes stream through once and the MFMA work is
// small (2 k-steps per tile and 32-row stripe), so what matters is (a) how many bytes each CU keeps in
// flight and (b) how few instructions a stripe costs (8 waves: the kernel is instruction-issue bound
// before it is bandwidth bound).  PF stripes are prefetched into registers ahead of the one being
// multiplied, the LDS tiles are double buffered (one barrier per stripe), LayerNorm's gamma/beta sit
// in LDS, and every address is a precomputed per-thread offset plus a wave-uniform base.
// XF: transform of X while it is staged — 0 none, 1 LayerNorm, 2 GELU, 3 any other activation,
//     4 x-hat = (x - mean) * rstd only (the affine part and d(gamma)/d(beta) are finished by the reduction)
// SZ: staging-plan size — 0: up to 384 x 256 columns, 2 stripes in flight; 1: N <= 256, K+1 <= 128, 3 stripes;
//     2: N <= 128, K+1 <= 128, 5 stripes (narrow layers are bound by bytes in flight, not by bandwidth)
// SP (fp32 rows only): the split arithmetic of RDST_F32X3.  stash() splits every staged pack into its bf16 hi and lo terms and
// keeps them as two bf16 planes of the tile row ([hi: 32 NT][lo: 32 NT] elements), so multiply() reads both operands TRANSPOSED
// with ds_read_b64_tr_b16 like the bf16 kernel: a k-step is 8 tokens (4 per lane half: hi in slots 0-3, lo in 4-7), two MFMAs.
template <typename T, int PFX, int XF, int SZ = 0, bool SP = false>
__global__ void __launch_bounds__(WG_THREADS) lin_wgrad_mfma_kernel(const WgradArgs<T> p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using MM = Mma<T, SP>;
  constexpr bool SPL = MM::SPLIT;
  constexpr int HP = MM::HP;
  constexpr int ES = (int)sizeof(T);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, h = lane >> 5;
  const int buf_bytes = WG_STRIPE * (p.ldn + p.ldk);
  float* gam = reinterpret_cast<float*>(smem + 2 * (size_t)buf_bytes);  // [KT*32] gamma, then [KT*32] beta
  float* bet = gam + p.KT * 32;
  if (XF == 1)
    for (int k = tid; k < p.KT * 32; k += WG_THREADS) {
      gam[k] = k < p.K ? p.lnw[k] : 0.f;
      bet[k] = k < p.K ? p.lnb[k] : 0.f;
    }
  f32x16 acc[WG_MAXT];
#pragma unroll
  for (int j = 0; j < WG_MAXT; ++j)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[j][v] = 0.f;
  const int ntiles = p.NT * p.KT;
  const int64_t m_begin = (int64_t)blockIdx.x * p.rows_per_wg;
  const int64_t m_end = (m_begin + p.rows_per_wg < p.M) ? m_begin + p.rows_per_wg : p.M;
  const int npk = p.NT * 32 / HP, kpk = p.KT * 32 / HP;

  // Per-thread staging plan (loop invariant): which (stripe row, pack) of dY / X this thread moves:
  // 32 rows x (NT*32 <= 384 | KT*32 <= 256) elements over 512 threads.
  // A pack that would run past the row's last element is loaded as the row's LAST full 16 bytes and
  // shifted down in registers (never reads outside the row); what is shifted in above the row's end is
  // don't-care: it only reaches accumulator rows / columns that are never stored, or is overwritten by
  // the X transform (zeros, ones column).  Rows are dword aligned (checked by the launcher).
  constexpr int DYMAX = SZ == 0 ? (24 + HP - 1) / HP : SZ == 1 ? 16 / HP : 8 / HP;
  constexpr int XMAX = SZ == 0 ? 16 / HP : 8 / HP;
  constexpr int PF = sizeof(T) == 2 ? (SZ == 0 ? PFX : SZ == 1 ? 3 : 5) : PFX;
  int dy_row[DYMAX], x_row[XMAX];       // stripe row
  int dy_col[DYMAX], x_col[XMAX];       // byte offset of the 16-B load inside the global row
  int dy_sh[DYMAX], x_sh[XMAX];         // right shift in bytes after the load (0 = full pack)
  int dy_lds[DYMAX], x_lds[XMAX];       // byte offset inside the LDS tile
  int x_k0[XMAX];
#pragma unroll
  for (int i = 0; i < DYMAX; ++i) {
    // a slot past the stripe's pack count repeats an earlier item (same loads, same LDS stores): no per-slot
    // predicate survives into the loop (each one was an exec-mask region and a live SGPR pair per stripe)
    const int idx = (tid + WG_THREADS * i) % (WG_STRIPE * npk);
    const int row = idx / npk, pk = idx - row * npk;
    int k0 = pk * HP;
    dy_row[i] = row;
    dy_lds[i] = row * p.ldn + pk * (SPL ? 8 : 16);
    if (k0 >= p.N) k0 = 0;  // slot entirely past the row: any in-row pack will do (feeds unstored rows)
    dy_sh[i] = (k0 + HP > p.N) ? (k0 + HP - p.N) * ES : 0;
    dy_col[i] = k0 * ES - dy_sh[i];
  }
#pragma unroll
  for (int i = 0; i < XMAX; ++i) {
    const int idx = (tid + WG_THREADS * i) % (WG_STRIPE * kpk);
    const int row = idx / kpk, pk = idx - row * kpk;
    const int k0 = pk * HP;
    x_row[i] = row;
    x_lds[i] = WG_STRIPE * p.ldn + row * p.ldk + pk * (SPL ? 8 : 16);
    x_k0[i] = k0;
    const int kl = k0 < p.K ? k0 : 0;
    x_sh[i] = (kl + HP > p.K) ? (kl + HP - p.K) * ES : 0;
    x_col[i] = kl * ES - x_sh[i];
  }
  auto shift_pack = [&](Pack16& q, int sh_bytes) {  // q >>= 8*sh_bytes over the 128-bit pack
    const int dq = sh_bytes >> 2;
    if (dq & 1) { q.w[0] = q.w[1]; q.w[1] = q.w[2]; q.w[2] = q.w[3]; }
    if (dq & 2) { q.w[0] = q.w[2]; q.w[1] = q.w[3]; }
    const uint32_t bs = (uint32_t)(sh_bytes & 3);
    q.w[0] = __builtin_amdgcn_alignbyte(q.w[1], q.w[0], bs);
    q.w[1] = __builtin_amdgcn_alignbyte(q.w[2], q.w[1], bs);
    q.w[2] = __builtin_amdgcn_alignbyte(q.w[3], q.w[2], bs);
    q.w[3] = __builtin_amdgcn_alignbyte(q.w[3], q.w[3], bs);
  };
  auto ld16 = [&](const char* base, uint32_t off) {
    const u32x4_a4 v = *reinterpret_cast<const u32x4_a4*>(base + off);
    Pack16 q;
    q.w[0] = v.x; q.w[1] = v.y; q.w[2] = v.z; q.w[3] = v.w;
    return q;
  };
  Pack16 rdy[PF][DYMAX], rx[PF][XMAX];
  float rmean[PF][XMAX], rrstd[PF][XMAX];
  const int64_t ldy_b = p.lddy * ES, ldx_b = p.ldx * ES;
  auto prefetch = [&](int set, int64_t m0) {
    const int left = (int)(m_end - m0 < WG_STRIPE ? m_end - m0 : WG_STRIPE);  // valid rows of the stripe (>= 1)
    const char* dyb = reinterpret_cast<const char*>(p.dY) + m0 * ldy_b;  // wave-uniform bases
    const char* xb = reinterpret_cast<const char*>(p.X) + m0 * ldx_b;
    const float* stb = p.stats + 2 * m0;
#pragma unroll
    for (int i = 0; i < DYMAX; ++i)
      {
        const int row = dy_row[i] < left ? dy_row[i] : left - 1;  // rows past the range: any real row (their X rows are zeroed)
        rdy[set][i] = ld16(dyb, (uint32_t)(row * (int)ldy_b + dy_col[i]));
      }
#pragma unroll
    for (int i = 0; i < XMAX; ++i)
      {
        const int row = x_row[i] < left ? x_row[i] : left - 1;
        rx[set][i] = ld16(xb, (uint32_t)(row * (int)ldx_b + x_col[i]));
        if (XF == 1 || XF == 4) {
          const float2 ms = *reinterpret_cast<const float2*>(stb + 2 * row);
          rmean[set][i] = ms.x;
          rrstd[set][i] = ms.y;
        }
      }
  };
  // registers -> LDS tile `b`, with the X-side transform (LayerNorm / activation / ones column)
  auto stash = [&](int set, int64_t m0, int b) {
    char* tile = smem + b * buf_bytes;
    const int left = (int)(m_end - m0 < WG_STRIPE ? m_end - m0 : WG_STRIPE);
#pragma unroll
    for (int i = 0; i < DYMAX; ++i)
      {
        Pack16 q = rdy[set][i];
        if (dy_sh[i]) shift_pack(q, dy_sh[i]);
        if constexpr (SPL) {
          const Pack16 sp = MM::op(q);
          *reinterpret_cast<uint2*>(tile + dy_lds[i]) = make_uint2(sp.w[0], sp.w[1]);
          *reinterpret_cast<uint2*>(tile + dy_lds[i] + p.NT * 64) = make_uint2(sp.w[2], sp.w[3]);
        } else {
          *reinterpret_cast<Pack16*>(tile + dy_lds[i]) = q;
        }
      }
#pragma unroll
    for (int i = 0; i < XMAX; ++i)
      {
        const bool valid = x_row[i] < left;
        const int k0 = x_k0[i];
        Pack16 q = rx[set][i];
        if (x_sh[i]) shift_pack(q, x_sh[i]);
        if (XF != 0 || !valid || k0 + HP > p.K) {
          float f[HP];
          MM::unpack(q, f);
          if (XF == 1) {
            const float4* g4 = reinterpret_cast<const float4*>(gam + k0);
            const float4* b4 = reinterpret_cast<const float4*>(bet + k0);
            float gv[HP], bv[HP];
#pragma unroll
            for (int e4 = 0; e4 < HP / 4; ++e4) {
              const float4 g = g4[e4], bb = b4[e4];
              gv[4 * e4] = g.x; gv[4 * e4 + 1] = g.y; gv[4 * e4 + 2] = g.z; gv[4 * e4 + 3] = g.w;
              bv[4 * e4] = bb.x; bv[4 * e4 + 1] = bb.y; bv[4 * e4 + 2] = bb.z; bv[4 * e4 + 3] = bb.w;
            }
            const float rs = rrstd[set][i], mu = rmean[set][i];
#pragma unroll
            for (int e = 0; e < HP; ++e) f[e] = (f[e] - mu) * rs * gv[e] + bv[e];
          } else if (XF == 4) {
            const float rs = rrstd[set][i], mu = rmean[set][i];
#pragma unroll
            for (int e = 0; e < HP; ++e) f[e] = (f[e] - mu) * rs;
          } else if (XF == 2) {
#pragma unroll
            for (int e = 0; e < HP; ++e) f[e] = (sizeof(T) == 2 || SPL) ? gelu_fast(f[e]) : gelu_erf(f[e]);
          } else if (XF == 3) {
#pragma unroll
            for (int e = 0; e < HP; ++e) f[e] = apply_act(f[e], p.in_act);
          }
          if (!valid || k0 + HP > p.K) {
#pragma unroll
            for (int e = 0; e < HP; ++e) {   // selects, not branches; ones column: dW[:, K] = sum_m dY = d(bias)
              const int kk = k0 + e;
              f[e] = !valid ? 0.f : (kk > p.K ? 0.f : (kk == p.K ? 1.0f : f[e]));
            }
          }
          q = MM::pack(f);
        }
        if constexpr (SPL) {
          const Pack16 sp = MM::op(q);
          *reinterpret_cast<uint2*>(tile + x_lds[i]) = make_uint2(sp.w[0], sp.w[1]);
          *reinterpret_cast<uint2*>(tile + x_lds[i] + p.KT * 64) = make_uint2(sp.w[2], sp.w[3]);
        } else {
          *reinterpret_cast<Pack16*>(tile + x_lds[i]) = q;
        }
      }
  };
  // per-wave tile list and per-lane fragment offsets (loop invariant)
  int tA[WG_MAXT], tB[WG_MAXT];
  {
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    constexpr bool TR = sizeof(T) == 2 || SPL;       // transposed 16-bit reads
    constexpr int RH = SPL ? 4 : 8;                  // token rows per lane half and k-step
    const int laneA = TR ? (RH * h + q) * p.ldn + (16 * (g & 1) + 4 * pp) * 2 : h * p.ldn + r * 4;
    const int laneB = TR ? (RH * h + q) * p.ldk + (16 * (g & 1) + 4 * pp) * 2 : h * p.ldk + r * 4;
#pragma unroll
    for (int j = 0; j < WG_MAXT; ++j) {
      const int ti = wave + WG_WAVES * j;
      const int nt = ti / p.KT, kt = ti - nt * p.KT;
      tA[j] = laneA + nt * 32 * (TR ? 2 : ES);
      tB[j] = WG_STRIPE * p.ldn + laneB + kt * 32 * (TR ? 2 : ES);
    }
  }
  // number of tiles of this wave (tiles wave, wave+8, ...): a wave-uniform count keeps the tile guards scalar
  const int my_tiles = __builtin_amdgcn_readfirstlane(ntiles > wave ? (ntiles - wave + WG_WAVES - 1) / WG_WAVES : 0);
  auto multiply = [&](int b) {
    const char* tile = smem + b * buf_bytes;
    if constexpr (sizeof(T) == 2 || SPL) {
      // transposed fragment reads (a 16-lane group covers 16 columns, 4 token rows per read), software
      // pipelined: the next tile's fragments are in flight while this tile's MFMAs issue
      // (split mode: the second read of a pack is the SAME 4 token rows in the lo plane, a k-step is 8 rows)
      typedef __attribute__((address_space(3))) s16x4_t* lds_p;
      constexpr int KR = SPL ? 8 : 16;
      constexpr int NMS = WG_STRIPE / KR;
      const int offA = SPL ? p.NT * 64 : 4 * p.ldn, offB = SPL ? p.KT * 64 : 4 * p.ldk;
      constexpr int NB = SPL ? 1 : 2;   // (split mode: 4 k-steps per stripe — a second fragment set would not fit the registers)
      Pack16 fa[NB][NMS], fb[NB][NMS];
      auto frags = [&](int j, Pack16 (&A)[NMS], Pack16 (&B)[NMS]) {
#pragma unroll
        for (int ms = 0; ms < NMS; ++ms) {
          const char* ta = tile + ms * KR * p.ldn;  // wave uniform
          const char* tb = tile + ms * KR * p.ldk;
          const s16x4_t a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(ta + tA[j]));
          const s16x4_t a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(ta + offA + tA[j]));
          const s16x4_t b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(tb + tB[j]));
          const s16x4_t b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(tb + offB + tB[j]));
          const uint2 ua0 = __builtin_bit_cast(uint2, a0), ua1 = __builtin_bit_cast(uint2, a1);
          const uint2 ub0 = __builtin_bit_cast(uint2, b0), ub1 = __builtin_bit_cast(uint2, b1);
          A[ms].w[0] = ua0.x; A[ms].w[1] = ua0.y; A[ms].w[2] = ua1.x; A[ms].w[3] = ua1.y;
          B[ms].w[0] = ub0.x; B[ms].w[1] = ub0.y; B[ms].w[2] = ub1.x; B[ms].w[3] = ub1.y;
        }
      };
      if (NB == 2 && my_tiles > 0) frags(0, fa[0], fb[0]);
#pragma unroll
      for (int j = 0; j < WG_MAXT; ++j) {
        if (j < my_tiles) {
          if (NB == 1) frags(j, fa[0], fb[0]);
          else if (j + 1 < WG_MAXT && j + 1 < my_tiles) frags(j + 1, fa[(j + 1) & (NB - 1)], fb[(j + 1) & (NB - 1)]);
#pragma unroll
          for (int ms = 0; ms < NMS; ++ms) MM::mma(acc[j], fa[j & (NB - 1)][ms], fb[j & (NB - 1)][ms]);
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < WG_MAXT; ++j) {
        if (j < my_tiles) {
#pragma unroll 8
          for (int s2 = 0; s2 < WG_STRIPE / 2; ++s2) {
            const float av = *reinterpret_cast<const float*>(tile + 2 * s2 * p.ldn + tA[j]);
            const float bv = *reinterpret_cast<const float*>(tile + 2 * s2 * p.ldk + tB[j]);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[j], 0, 0, 0);
          }
        }
      }
    }
  };

  // every prefetch defines every staging register (past the range it re-reads the first stripe): a
  // conditionally defined register is live around the whole loop and invites spills
#pragma unroll
  for (int s = 0; s < PF; ++s) {
    const int64_t ms = m_begin + (int64_t)s * WG_STRIPE;
    if (m_begin < m_end) prefetch(s, ms < m_end ? ms : m_begin);
  }
  __syncthreads();  // gamma / beta staged
  int b = 0;
  for (int64_t mg = m_begin; mg < m_end; mg += (int64_t)PF * WG_STRIPE) {
#pragma unroll
    for (int s = 0; s < PF; ++s) {
      const int64_t m0 = mg + (int64_t)s * WG_STRIPE;
      if (m0 < m_end) {
        // tile b was last read two stripes ago and every wave has passed the barrier in between
        stash(s, m0, b);
        __syncthreads();
        const int64_t mn = m0 + (int64_t)PF * WG_STRIPE;
        prefetch(s, mn < m_end ? mn : m0);  // PF stripes ahead, in flight while this and the next ones are multiplied
        multiply(b);
        b ^= 1;
      }
    }
  }
  float* my = p.slab + (int64_t)blockIdx.x * p.N * p.Kx;
#pragma unroll
  for (int j = 0; j < WG_MAXT; ++j) {
    const int ti = wave + WG_WAVES * j;
    if (ti < ntiles) {
      const int nt = ti / p.KT, kt = ti - nt * p.KT;
      const int k = kt * 32 + r;
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int n = nt * 32 + acc_row(v, h);
        if (n < p.N && k < p.Kx) my[(int64_t)n * p.Kx + k] = acc[j][v];
      }
    }
  }
}

// (the slab sums and the LayerNorm finish live in reduce_batch.hip)
}  // namespace

int wgrad_reduce_launch(const float* slab, int nwg, int N, int K, float s, float* dW, float* dbias, hipStream_t st) {
  rbatch::SumJob j{};
  j.slab = slab; j.nwg = nwg; j.stride = (int64_t)N * (K + 1); j.tot = N * (K + 1); j.map = rbatch::MAP_LINEAR;
  j.out = dW; j.out2 = dbias; j.a = K; j.b = K + 1; j.s = s;
  return rbatch::sum(j, st);
}
int wgrad_sum_launch(const float* slab, int nwg, int tot, float* G, hipStream_t st) {
  rbatch::SumJob j{};
  j.slab = slab; j.nwg = nwg; j.stride = tot; j.tot = tot; j.map = rbatch::MAP_COPY; j.out = G;
  return rbatch::sum(j, st);
}
// launcher of the LayerNorm finish for other translation units (mlp_mfma.hip)
int wgrad_ln_finish_launch(const float* G, const float* Wt, const float* ln_w, const float* ln_b, int N, int K, float s,
                           float* dW, float* dbias, float* dln_w, float* dln_b, hipStream_t st) {
  rbatch::FinJob j{};
  j.G = G; j.Wt = Wt; j.gamma = ln_w; j.beta = ln_b; j.N = N; j.K = K; j.Kx = K + 1; j.s = s;
  j.dW = dW; j.dbias = dbias; j.dgamma = dln_w; j.dbeta = dln_b;
  return rbatch::finish(j, st);
}

template <typename T>
int linear_fwd_mfma(const T* X, int64_t ldx, const float* ln_w, const float* ln_b, int in_act, const float* Wt,
                    const float* bias, const T* R, int64_t ldr, T* Y, int64_t ldy, float* stats, int64_t M, int K,
                    int N, float s, hipStream_t st) {
  if (mfma_disabled() || !rows_ok<T>(X, ldx)) return RDST_ENOTSUP;
  LinArgs<T> p{};
  p.A = X; p.lda = ldx; p.lnw = ln_w; p.lnb = ln_b; p.in_act = in_act; p.Wt = Wt; p.wK = K; p.bias = bias;
  p.R = R; p.ldr = ldr; p.Y = Y; p.ldy = ldy; p.stats = stats; p.M = M; p.Kc = K; p.Nout = N; p.s = s;
  return launch_lin<T, MODE_FWD>(p, st, "lin_fwd_mfma");
}

template <typename T>
int linear_dgrad_mfma(const T* X, int64_t ldx, bool has_ln, int in_act, const float* Wt, const T* dY, int64_t lddy,
                      T* dX, int64_t lddx, const T* acc, int64_t ldacc, float* dA, int64_t M, int K, int N, float s,
                      hipStream_t st) {
  if (mfma_disabled() || !rows_ok<T>(dY, lddy)) return RDST_ENOTSUP;
  LinArgs<T> p{};
  p.A = dY; p.lda = lddy; p.in_act = in_act; p.Wt = Wt; p.wK = K; p.Y = dX; p.ldy = lddx;
  p.dA = has_ln ? dA : nullptr; p.Xa = X; p.ldxa = ldx; p.Acc = acc; p.ldacc = ldacc;
  p.M = M; p.Kc = N; p.Nout = K; p.s = s;
  return launch_lin<T, MODE_DGRAD>(p, st, "lin_dgrad_mfma");
}

// dgrad + LayerNorm backward in one kernel; writes dX and a [*nslab][2][K] slab of d(gamma)/d(beta) partials
template <typename T>
int linear_dgrad_ln_mfma(const T* X, int64_t ldx, const float* stats, const float* gamma, const float* Wt, const T* dY,
                         int64_t lddy, T* dX, int64_t lddx, const T* acc, int64_t ldacc, float* slab, int* nslab,
                         int64_t M, int K, int N, float s, hipStream_t st) {
  using MM = Mma<T>;
  if (mfma_disabled() || K > 128 || !dX) return RDST_ENOTSUP;
  LnDgradArgs<T> p{};
  p.dY = dY; p.lddy = lddy; p.Wt = Wt; p.N = N; p.K = K; p.X = X; p.ldx = ldx; p.stats = stats; p.gamma = gamma;
  p.dX = dX; p.lddx = lddx; p.Acc = acc; p.ldacc = ldacc; p.slab = slab; p.M = M; p.s = s;
  p.Tn = (N + MM::KP - 1) / MM::KP;
  if (p.Tn > 32) return RDST_ENOTSUP;
  p.ldw = lds_row_bytes(N, sizeof(T));
  const int kpad = ((K + 31) / 32) * 32;
  const size_t smem = (size_t)kpad * p.ldw + 8 * 4096 + 8 * 2 * 128 * sizeof(float);
  if (smem > 160 * 1024) return RDST_ENOTSUP;
  const int64_t nslabs = (M + 31) / 32;
  int64_t grid = (nslabs + 7) / 8;
  const int64_t cap = 256;   // persistent, one workgroup per CU (185+ VGPRs: a second one would not be resident anyway)
  if (grid > cap) grid = cap;
  *nslab = (int)grid;
#define RDST_LND_LAUNCH(TM)                                                                                          \
  {                                                                                                                  \
    auto kern = lin_dgrad_ln_kernel<T, TM>;                                                                          \
    if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), smem, st, p);                                          \
  }
  if (p.Tn <= 8) RDST_LND_LAUNCH(8) else if (p.Tn <= 16) RDST_LND_LAUNCH(16) else RDST_LND_LAUNCH(32)
#undef RDST_LND_LAUNCH
  return rdst_launch_status("lin_dgrad_ln_mfma");
}

int linear_wgrad_max_wgs(int N);
size_t linear_wgrad_mfma_slab_floats(int64_t M, int K, int N) {
  (void)M;
  // 256 workgroups, or up to 1024 small ones for the narrow projections (linear_ln_bwd_fused_bf16)
  return (size_t)linear_wgrad_max_wgs(N) * N * (K + 1);
}
int linear_wgrad_max_wgs(int N) { return N <= 128 ? 1024 : N <= 192 ? 512 : 256; }

namespace {
template <typename T>
int wgrad_impl(const T* X, int64_t ldx, const float* ln_w, const float* ln_b, const float* stats, int in_act,
               const T* dY, int64_t lddy, float* dW, float* dbias, float* slab, int64_t M, int K, int N, float s,
               const float* Wt_fin, float* G, float* dln_w, float* dln_b, hipStream_t st) {
  const bool lnfin = G != nullptr;   // LayerNorm finish: x-hat slabs, then dW/dbias/d(gamma)/d(beta) from G
  if (mfma_disabled() || !rows_ok<T>(X, ldx) || !rows_ok<T>(dY, lddy)) return RDST_ENOTSUP;
  WgradArgs<T> p{};
  p.X = X; p.ldx = ldx; p.lnw = ln_w; p.lnb = ln_b; p.stats = stats; p.in_act = in_act; p.dY = dY; p.lddy = lddy;
  p.slab = slab; p.M = M; p.K = K; p.N = N; p.Kx = K + 1;
  p.NT = (N + 31) / 32; p.KT = (p.Kx + 31) / 32;
  if (p.NT * p.KT > WG_WAVES * WG_MAXT || p.NT > 12 || p.KT > 8) return RDST_ENOTSUP;
  // the kernel moves whole 16-B packs: dword-aligned rows and at least one pack per row
  if (K < Mma<T>::HP || N < Mma<T>::HP || ((uintptr_t)X & 3) || ((uintptr_t)dY & 3) || (ldx * sizeof(T)) % 4 ||
      (lddy * sizeof(T)) % 4 || (K * sizeof(T)) % 4 || (N * sizeof(T)) % 4)
    return RDST_ENOTSUP;
  // LDS row strides: bf16 rows are read by ds_read_b64_tr_b16 (4 token rows x 64 B per 32 lanes):
  // stride = 64 (mod 256) bytes puts the 4 rows on disjoint bank ranges; fp32 rows are read 32
  // consecutive floats at a time, any stride works.
  bool split = false;
  if constexpr (sizeof(T) == 4) split = rdst_split();
  auto stride = [split](int elems) {   // (split mode: two bf16 planes in the bytes of the fp32 row, read like bf16 rows)
    const int b = elems * (int)sizeof(T);
    if (sizeof(T) == 4 && !split) return b;
    return b <= 64 ? 64 : ((b - 64 + 255) / 256) * 256 + 64;
  };
  p.ldn = stride(p.NT * 32);
  p.ldk = stride(p.KT * 32);
  const size_t smem = (size_t)2 * WG_STRIPE * (p.ldn + p.ldk) + (size_t)2 * p.KT * 32 * sizeof(float);
  if (smem > 160 * 1024) return RDST_ENOTSUP;
  int64_t nwg = (M + WG_STRIPE - 1) / WG_STRIPE;
  if (nwg > 256) nwg = 256;
  p.rows_per_wg = (((M + nwg - 1) / nwg + WG_STRIPE - 1) / WG_STRIPE) * WG_STRIPE;
  nwg = (M + p.rows_per_wg - 1) / p.rows_per_wg;
  constexpr int PF = sizeof(T) == 2 ? 2 : 1;  // register sets of prefetched stripes
  const int xf = lnfin ? 4 : ln_w ? 1 : in_act == RDST_ACT_GELU ? 2 : in_act ? 3 : 0;
  // plan size: packs per stripe over 512 threads (bf16 only; the fp32 parity mode keeps the general plan)
  const int sz = sizeof(T) != 2 ? 0 : (p.NT * 32 <= 128 && p.KT * 32 <= 128) ? 2 : (p.NT * 32 <= 256 && p.KT * 32 <= 128) ? 1 : 0;
#define RDST_WG_LAUNCH(XF)                                                                                            \
  {                                                                                                                  \
    auto kern = sz == 2 ? lin_wgrad_mfma_kernel<T, PF, XF, 2> : sz == 1 ? lin_wgrad_mfma_kernel<T, PF, XF, 1>        \
                                                                         : lin_wgrad_mfma_kernel<T, PF, XF, 0>;      \
    if constexpr (sizeof(T) == 4)                                                                                    \
      if (split) kern = lin_wgrad_mfma_kernel<T, PF, XF, 0, true>;                                                   \
    if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(WG_THREADS), smem, st, p);                                    \
  }
  if (xf == 0) RDST_WG_LAUNCH(0) else if (xf == 1) RDST_WG_LAUNCH(1) else if (xf == 2) RDST_WG_LAUNCH(2) else if (xf == 4) RDST_WG_LAUNCH(4) else RDST_WG_LAUNCH(3)
#undef RDST_WG_LAUNCH
  if (int rc = rdst_launch_status("lin_wgrad_mfma")) return rc;
  const int tot = N * p.Kx;
  if (lnfin) {
    rbatch::SumJob sj{};
    sj.slab = slab; sj.nwg = (int)nwg; sj.stride = tot; sj.tot = tot; sj.map = rbatch::MAP_COPY; sj.out = G;
    if (int rc = rbatch::sum(sj, st)) return rc;
    rbatch::FinJob fj{};
    fj.G = G; fj.Wt = Wt_fin; fj.gamma = ln_w; fj.beta = ln_b; fj.N = N; fj.K = K; fj.Kx = p.Kx; fj.s = s;
    fj.dW = dW; fj.dbias = dbias; fj.dgamma = dln_w; fj.dbeta = dln_b;
    return rbatch::finish(fj, st);
  }
  rbatch::SumJob sj{};
  sj.slab = slab; sj.nwg = (int)nwg; sj.stride = tot; sj.tot = tot; sj.map = rbatch::MAP_LINEAR;
  sj.out = dW; sj.out2 = dbias; sj.a = K; sj.b = p.Kx; sj.s = s;
  return rbatch::sum(sj, st);
}
}  // namespace

template <typename T>
int linear_wgrad_mfma(const T* X, int64_t ldx, const float* ln_w, const float* ln_b, const float* stats, int in_act,
                      const T* dY, int64_t lddy, float* dW, float* dbias, float* slab, int64_t M, int K, int N, float s,
                      hipStream_t st) {
  return wgrad_impl<T>(X, ldx, ln_w, ln_b, stats, in_act, dY, lddy, dW, dbias, slab, M, K, N, s, nullptr, nullptr, nullptr,
                       nullptr, st);
}

// LayerNorm-fused Linear: dW, dbias AND d(gamma), d(beta) from one pass over (x-hat, dY).  G: N*(K+1) floats of scratch.
template <typename T>
int linear_wgrad_ln_mfma(const T* X, int64_t ldx, const float* ln_w, const float* ln_b, const float* stats, const float* Wt,
                         const T* dY, int64_t lddy, float* dW, float* dbias, float* dln_w, float* dln_b, float* slab,
                         float* G, int64_t M, int K, int N, float s, hipStream_t st) {
  if (!ln_w || !ln_b || !stats || !Wt || !G) return RDST_ENOTSUP;
  return wgrad_impl<T>(X, ldx, ln_w, ln_b, stats, 0, dY, lddy, dW, dbias, slab, M, K, N, s, Wt, G, dln_w, dln_b, st);
}

// dgrad + LayerNorm backward, dX only (d(gamma)/d(beta) come from linear_wgrad_ln_mfma); K <= 128
template <typename T>
int linear_dgrad_ln2_mfma(const T* X, int64_t ldx, const float* stats, const float* gamma, const float* Wt, const T* dY,
                          int64_t lddy, T* dX, int64_t lddx, const T* acc, int64_t ldacc, int64_t M, int K, int N, float s,
                          hipStream_t st) {
  using MM = Mma<T>;
  if (mfma_disabled() || K > 128 || !dX) return RDST_ENOTSUP;
  const int64_t rowbytes = (int64_t)N * (int64_t)sizeof(T);
  if (((uintptr_t)dY & 3) || (lddy * sizeof(T)) % 4 || rowbytes % 4 || rowbytes < 16 || !(rowbytes % 128 == 0 || rowbytes % 128 >= 16))
    return RDST_ENOTSUP;
  LnDgradArgs<T> p{};
  p.dY = dY; p.lddy = lddy; p.Wt = Wt; p.N = N; p.K = K; p.X = X; p.ldx = ldx; p.stats = stats; p.gamma = gamma;
  p.dX = dX; p.lddx = lddx; p.Acc = acc; p.ldacc = ldacc; p.slab = nullptr; p.M = M; p.s = s;
  p.Tn = (N + MM::KP - 1) / MM::KP;
  if (p.Tn > 32) return RDST_ENOTSUP;
  p.ldw = lds_row_bytes(N, sizeof(T));
  const int nct = (K + 31) / 32, kpad = nct * 32;
  const size_t smem = (size_t)kpad * p.ldw + (size_t)kpad * sizeof(float) + 8 * 32 * 144;
  if (smem > 160 * 1024) return RDST_ENOTSUP;
  const int64_t nslabs = (M + 31) / 32;
  int64_t grid = (nslabs + 7) / 8;
  if (grid > 256) grid = 256;   // persistent, one workgroup per CU
#define RDST_LND2_LAUNCH(NC)                                                                                          \
  {                                                                                                                  \
    auto kern = lin_dgrad_ln2_kernel<T, NC>;                                                                         \
    if constexpr (sizeof(T) == 4)                                                                                    \
      if (rdst_split()) kern = lin_dgrad_ln2_kernel<T, NC, true>;                                                    \
    if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), smem, st, p);                                          \
  }
  if (nct == 1) RDST_LND2_LAUNCH(1) else if (nct == 2) RDST_LND2_LAUNCH(2) else if (nct == 3) RDST_LND2_LAUNCH(3) else RDST_LND2_LAUNCH(4)
#undef RDST_LND2_LAUNCH
  return rdst_launch_status("lin_dgrad_ln2_mfma");
}

#define INST(T)                                                                                                        \
  template int linear_dgrad_ln_mfma<T>(const T*, int64_t, const float*, const float*, const float*, const T*, int64_t, \
                                       T*, int64_t, const T*, int64_t, float*, int*, int64_t, int, int, float,         \
                                       hipStream_t);                                                                   \
  template int linear_fwd_mfma<T>(const T*, int64_t, const float*, const float*, int, const float*, const float*,     \
                                  const T*, int64_t, T*, int64_t, float*, int64_t, int, int, float, hipStream_t);     \
  template int linear_dgrad_mfma<T>(const T*, int64_t, bool, int, const float*, const T*, int64_t, T*, int64_t,       \
                                    const T*, int64_t, float*, int64_t, int, int, float, hipStream_t);                \
  template int linear_wgrad_mfma<T>(const T*, int64_t, const float*, const float*, const float*, int, const T*,       \
                                    int64_t, float*, float*, float*, int64_t, int, int, float, hipStream_t);         \
  template int linear_wgrad_ln_mfma<T>(const T*, int64_t, const float*, const float*, const float*, const float*,     \
                                       const T*, int64_t, float*, float*, float*, float*, float*, float*, int64_t,    \
                                       int, int, float, hipStream_t);                                                 \
  template int linear_dgrad_ln2_mfma<T>(const T*, int64_t, const float*, const float*, const float*, const T*, int64_t, \
                                        T*, int64_t, const T*, int64_t, int64_t, int, int, float, hipStream_t);
INST(float)
INST(bf16)
