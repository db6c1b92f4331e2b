// K4/K5 for the E1 shapes in the RDST_F32X3 arithmetic: 3x3 / pad 1 convolution, forward and dgrad, on fp32 ROWS with every matrix
// operand as two bf16 terms (hi = bf16(v), lo = bf16(v - hi)) and three v_mfma_f32_32x32x16_bf16 per product (hi.hi + hi.lo + lo.hi).
// It is conv3_mfma.hip (bf16 rows: weights stationary in registers, input rows rolling through an LDS ring by LDS-DMA) rebuilt for
// 4-byte pixels; the round-5 split mode ran the row-stripe kernel of conv_mfma.hip, which re-stages and re-splits the fp32 weights
// of a column chunk in every workgroup (226 us per launch where the bf16 kernel takes 37):
//   * the weights are PREPACKED hi / lo fragment pairs (pack.h: conv3x_pack_block) and live in registers for the whole kernel — a
//     wave's share is at most 90 fragments, as in the bf16 kernel, so the 150 -> 60 forward splits its contraction over wave pairs
//     (KSPLIT 2: the two halves exchange one accumulator tile each through LDS) and the 60 -> 240 forward runs as two launches of
//     four channel tiles;
//   * the RAW fp32 pixels of an input row arrive by LDS-DMA (pixels outside the image and pad slots are zeros); when a step's new
//     rows have landed ONE conversion pass, shared by the four waves, rewrites them in place as 64-byte groups
//     [8 hi | 8 hi | 8 lo | 8 lo] of 16 channels: the B operand of a (tap, k-step) is two ds_read_b128 at row / pixel offsets of the
//     same image; one more barrier per step;
//   * accumulators transposed (output channel in the registers, pixel on the lane), bias = initial value, scale folded into the
//     weights, residual / dX_add and the stores as 16-byte fp32 row chunks after one v_permlane32_swap per register pair.
// Shapes: 150 -> 60 forward, 60 -> 60 forward / dgrad, 60 -> 240 + PixelShuffle(2) forward, 60 -> 150 dgrad.  The 240 -> 60 dgrad of
// the upsampler convolutions and everything else stay on conv_mfma.hip.
#include "conv.h"
#include "mfma.h"
#include "pack.h"
#include <type_traits>

namespace {


__global__ void __launch_bounds__(256) conv3x_pack_kernel(const float* __restrict__ Wc, uint32_t* __restrict__ out, int Cin,
                                                          int Cout, int K, int N, int ksteps, int ctiles, int mode, float s, int cmul,
                                                          int coff) {
  conv3x_pack_block((int)blockIdx.x, Wc, out, Cin, Cout, K, N, ksteps, ctiles, mode, s, cmul, coff);
}

struct C3Args {
  const float* A; int64_t lda;    // input rows
  int a_bytes;                    // extent of A in bytes (< 2^31): the buffer descriptor's range
  int ymul, yoff, xmul, xoff;     // pixel (y, x) of the conv's grid = memory pixel (y ymul + yoff, x xmul + xoff) of A (a sub-pixel view)
  const uint32_t* Wp;             // packed hi / lo weight fragments of this launch's channel tiles (tile 0 = channel n0)
  const float* bias;              // (N) or null
  const float* R; int64_t ldr;    // forward residual / dgrad dX_add, output geometry; or null
  float* Y; int64_t ldy;
  int B, H, W;                    // the conv's resolution
  int n0, N;                      // first output channel of this launch, total output channels (store mask)
  float s;
  int SH, nys, nstrips;           // strip height, strips per image column, total strips
  unsigned long long* stamps;     // debug build: [grid][8] cycle counters (RDST_C3X_STAMPS), else null

};

constexpr int pix_stride(int K) {  // bytes: 64 per 16 channels (raw fp32, then [8 hi | 8 hi | 8 lo | 8 lo]) + one pad slot: odd slot count
  return (K + 15) / 16 * 64 + 16;
}

template <int K, int CW, int NT, int KSPLIT, int PSLOTS, int RPS, bool UNSHUF, bool PSTORE>
struct C3Cfg {
  static constexpr int KSTEPS = (K + 15) / 16;
  static constexpr int KSH = (KSTEPS + KSPLIT - 1) / KSPLIT;
  static constexpr int PSTRIDE = pix_stride(K);
  static constexpr int SPP = PSTRIDE / 16;                 // 16-B slots per pixel
  static constexpr int RPIECES = (34 * SPP + 63) / 64;     // 1-KB LDS-DMA pieces per ring row
  static constexpr int ROWB = RPIECES * 1024;
  static constexpr int NR = 2 * RPS + 2;
  static constexpr int RPW = RPS / PSLOTS;                 // output rows per wave and step
  static constexpr int XBUF = KSPLIT > 1 ? 4 * NT * 4096 : 0;   // accumulator exchange, per wave NT tiles of 4 KB
  static constexpr int BIAS_OFF = NR * ROWB + 64;          // 64 B of zeroed slack behind the ring (k-step over-read)
  static constexpr int XBUF_OFF = BIAS_OFF + CW * NT * 32 * 4;
  static constexpr int SMEM = XBUF_OFF + XBUF;
  static constexpr int DSLOTS = (4 * K + 15) / 16;         // slots of a pixel that carry data
  static constexpr int NA_MAX = 60;                        // weight fragments kept in AGPRs (240 of the 256)
  static_assert(CW * KSPLIT * PSLOTS == 4, "4 waves");
  static_assert(RPS % PSLOTS == 0, "rows per step");
  static_assert(!UNSHUF, "the un-shuffled dgrad stays on conv_mfma.hip");
  static_assert(KSPLIT == 1 || RPW == 2, "the K halves exchange one row each");
};

template <int K, int CW, int NT, int KSPLIT, int PSLOTS, int RPS, bool UNSHUF, bool PSTORE>
__global__ void __launch_bounds__(256, 1) conv3x_kernel(const C3Args p) {
  using CF = C3Cfg<K, CW, NT, KSPLIT, PSLOTS, RPS, UNSHUF, PSTORE>;
  constexpr int KSTEPS = CF::KSTEPS, KSH = CF::KSH, PSTRIDE = CF::PSTRIDE, ROWB = CF::ROWB, NR = CF::NR, RPW = CF::RPW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cg = wave % CW, kh = (wave / CW) % KSPLIT, ps = wave / (CW * KSPLIT);
  const int H = p.H, W = p.W;
  unsigned long long tprev = RDST_DBGV(p.stamps) ? __builtin_readcyclecounter() : 0ull;
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define STAMP_ADD(k)                                                   \
  if (RDST_DBGV(p.stamps)) {                                           \
    const unsigned long long tn_ = __builtin_readcyclecounter();       \
    tacc[k] += tn_ - tprev;                                            \
    tprev = tn_;                                                       \
  }
  // buffer descriptor of the input tensor (raw buffer, byte offsets, range = a_bytes: reads behind it return zeros), in SGPRs
  typedef uint32_t u32x4s_t __attribute__((ext_vector_type(4)));
  u32x4s_t rsrc;
  rsrc.x = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)p.A);
  rsrc.y = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)p.A >> 32) & 0xffffu);
  rsrc.z = __builtin_amdgcn_readfirstlane((uint32_t)p.a_bytes);
  rsrc.w = 0x00020000u;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  lds_zero16(smem, CF::BIAS_OFF, tid, 256);
  float* biasL = reinterpret_cast<float*>(smem + CF::BIAS_OFF);
  for (int i = tid; i < CW * NT * 32; i += 256) biasL[i] = (p.bias && p.n0 + i < p.N) ? p.bias[p.n0 + i] * p.s : 0.f;
  __syncthreads();

  // ---- strips and their input rows -------------------------------------------------------------------------
  // rel row q of a strip = input row y0 - 1 + q, ring slot q % NR.  A ring row is RPIECES LDS-DMA pieces of 1 KB
  // (buffer_load_dwordx4 ... lds: lane l of a piece fills 16-B slot l; the SOURCE address is per lane): slot -> (pixel,
  // 16-B chunk of its channel row); pad slots, pixels outside the image and the bytes behind the tensor's end are
  // out-of-range offsets of the buffer descriptor, which the hardware turns into zeros.  No staging registers, no
  // ds_write; the pieces of a row set are dealt round-robin to the 4 waves.
  const int nxs = W / 32;
  struct Strip { int b, y0, x0, nrows; };
  auto decode = [&](int strip) {
    const int xs = strip % nxs, tq = strip / nxs;
    const int ysg = tq % p.nys;
    Strip s;
    s.b = tq / p.nys; s.y0 = ysg * p.SH; s.x0 = xs * 32;
    s.nrows = (H - s.y0 < p.SH) ? H - s.y0 : p.SH;
    return s;
  };
  auto lane_off = [&](const Strip& sp, int pi) {   // byte offset of this lane's 16-B chunk inside an input row, < 0: zeros
    const int sidx = pi * 64 + lane;
    const int px = sidx / CF::SPP, sl = sidx - px * CF::SPP;
    const int x = sp.x0 - 1 + px;
    const bool ok = x >= 0 && x < W && px < 34 && sl < CF::DSLOTS;
    int off;
    off = (x * p.xmul + p.xoff) * ((int)p.lda * 4) + sl * 16;
    return ok ? off : -1;
  };
  auto dma = [&](const Strip& sp, int rel, int pi, int loff) {
    const int y = sp.y0 - 1 + rel;
    const bool rowok = y >= 0 && y < H;
    const int rowbase = (int)((((int64_t)sp.b * (H * p.ymul) + (int64_t)y * p.ymul + p.yoff) * ((int64_t)W * p.xmul)) * (p.lda * 4));
    const int off = (rowok && loff >= 0) ? rowbase + loff : p.a_bytes;   // out of range -> the DMA writes zeros
    // Inline asm, not the builtin: the compiler orders every later ds_read behind a builtin LDS-DMA with s_waitcnt
    // vmcnt(0) (it cannot tell that the slots differ), which exposes the whole HBM latency in every step.  The waits are
    // placed by hand: vmcnt(0) after a step's MFMAs, in front of the barrier that publishes the rows.
    const uint32_t ldst = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)((rel % NR) * ROWB + pi * 1024));
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(off), "s"(ldst), "s"(rsrc) : "memory");
  };
  auto first_rows = [&](const Strip& sp) {           // rel rows 0 .. RPS + 1
    for (int q = wave; q < (RPS + 2) * CF::RPIECES; q += 4) {
      const int rr = q / CF::RPIECES, pi = q - rr * CF::RPIECES;
      dma(sp, rr, pi, lane_off(sp, pi));
    }
  };
  // the conversion pass of ring rows rel0 .. rel0 + nrel - 1 (landed, published by a barrier): raw fp32 -> hi / lo groups, in place
  auto convert_rows = [&](int rel0, int nrel) {
    for (int idx = tid; idx < nrel * 34 * KSTEPS; idx += 256) {
      const int rr = idx / (34 * KSTEPS), rem = idx - rr * (34 * KSTEPS), px = rem / KSTEPS, g = rem - px * KSTEPS;
      char* grp = smem + ((rel0 + rr) % NR) * ROWB + px * PSTRIDE + g * 64;
      float f[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(grp + 16 * q);
        // (channels past K: the tail of the pixel's last 16-byte slot belongs to the next channels of the same memory row)
        f[4 * q] = g * 16 + 4 * q < K ? v.x : 0.f; f[4 * q + 1] = g * 16 + 4 * q + 1 < K ? v.y : 0.f;
        f[4 * q + 2] = g * 16 + 4 * q + 2 < K ? v.z : 0.f; f[4 * q + 3] = g * 16 + 4 * q + 3 < K ? v.w : 0.f;
      }
      Pack16 h0, l0, h1, l1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        h0.w[e] = pack_bf16x2(f[2 * e], f[2 * e + 1]);
        l0.w[e] = pack_bf16x2(f[2 * e] - bf16lo(h0.w[e]), f[2 * e + 1] - bf16hi(h0.w[e]));
        h1.w[e] = pack_bf16x2(f[8 + 2 * e], f[9 + 2 * e]);
        l1.w[e] = pack_bf16x2(f[8 + 2 * e] - bf16lo(h1.w[e]), f[9 + 2 * e] - bf16hi(h1.w[e]));
      }
      Pack16* dst = reinterpret_cast<Pack16*>(grp);
      dst[0] = h0; dst[1] = h1; dst[2] = l0; dst[3] = l1;
    }
  };
  if ((int)blockIdx.x < p.nstrips) first_rows(decode(blockIdx.x));   // in flight together with the weight loads below

  // ---- this wave's weight fragments: registers for the whole kernel -------------------------------------------
  // The register file is 256 VGPRs + 256 AGPRs per lane; an MFMA takes its A operand from either.  Left alone, the
  // compiler keeps "spilling" fragments to AGPRs and copies each back (4 v_accvgpr_read per MFMA) under maximal
  // register pressure, which also serialises every ds_read with its MFMA.  So the first NA fragments are LOADED INTO
  // AGPRs by hand (inline asm, "=a": the value's register class is then the accumulator file and the MFMA reads it
  // there) and only the rest live in VGPRs.
  constexpr int NFR = 2 * NT * 9 * KSH;   // hi and lo fragment of every (tile, tap, k-step): f = 2 * ((t * 9 + tap) * KSH + kk) + lo
  constexpr int NA = NFR < CF::NA_MAX ? NFR : CF::NA_MAX, NV = NFR - NA;
  typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
  u32x4_t wfa[NA > 0 ? NA : 1];
  Pack16 wfv[NV > 0 ? NV : 1];
  auto wsrc = [&](int f2, bool& real) {
    const int f = f2 >> 1, lo = f2 & 1;
    const int t = f / (9 * KSH), rem = f - t * (9 * KSH), tap = rem / KSH, kk = rem - tap * KSH;
    const int ks = kh * KSH + kk;
    real = ks < KSTEPS;                      // the short K half pads with a zero fragment (its partner has KSH real ones)
    return reinterpret_cast<const char*>(p.Wp) + (((((int64_t)(cg * NT + t) * 9 + tap) * KSTEPS + (real ? ks : KSTEPS - 1)) * 2 + lo) * 64 + lane) * 16;
  };
#pragma unroll
  for (int f = 0; f < NA; ++f) {
    bool real;
    const char* src = wsrc(f, real);
    asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(wfa[f]) : "v"(src) : "memory");
  }
#pragma unroll
  for (int f = 0; f < NV; ++f) {
    bool real;
    const u32x4_a4 v = *reinterpret_cast<const u32x4_a4*>(wsrc(NA + f, real));
    wfv[f].w[0] = real ? v.x : 0u; wfv[f].w[1] = real ? v.y : 0u; wfv[f].w[2] = real ? v.z : 0u; wfv[f].w[3] = real ? v.w : 0u;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int f = 0; f < NA; ++f) {
    bool real;
    (void)wsrc(f, real);
    if (KSPLIT > 1 && !real) wfa[f] = u32x4_t{0u, 0u, 0u, 0u};
    asm volatile("" : "+a"(wfa[f]));         // ordered behind the wait: every use of the fragment depends on this
  }
  auto wfrag = [&](int t, int tap, int kk, int lo) {
    const int f = 2 * ((t * 9 + tap) * KSH + kk) + lo;
    return f < NA ? __builtin_bit_cast(bf16x8_t, wfa[f < NA ? f : 0]) : __builtin_bit_cast(bf16x8_t, wfv[f < NA ? 0 : f - NA]);
  };
  __syncthreads();   // (the first strip's rows landed: the weight wait above covered them)
  if ((int)blockIdx.x < p.nstrips) convert_rows(0, RPS + 2);
  __syncthreads();
  STAMP_ADD(0);
  for (int strip = blockIdx.x; strip < p.nstrips; strip += gridDim.x) {
    const Strip sp = decode(strip);
    const int b = sp.b, y0 = sp.y0, x0 = sp.x0, nrows = sp.nrows;
    const int nsteps = (nrows + RPS - 1) / RPS;
    if (strip != (int)blockIdx.x) {
      first_rows(sp);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      convert_rows(0, RPS + 2);
      __syncthreads();
    }
    // the pieces this wave loads in every step: fixed (row-in-step, piece) pairs -> their lane offsets are loop invariants
    constexpr int NPW = (RPS * CF::RPIECES + 3) / 4;
    int loff[NPW];
#pragma unroll
    for (int k = 0; k < NPW; ++k) loff[k] = lane_off(sp, (wave + 4 * k) % CF::RPIECES);
    STAMP_ADD(1);

    for (int j = 0; j < nsteps; ++j) {
      const bool more = j + 1 < nsteps;
      // residual / dX_add chunks of this step's outputs: in flight during the MFMAs
      constexpr int RFIN = KSPLIT == 2 ? 1 : RPW;           // rows this wave finishes per step
      uint32_t rpre[RFIN][NT][2][8];
      if constexpr (!PSTORE) {
        if (p.R) {
#pragma unroll
          for (int i = 0; i < RFIN; ++i) {
            const int yo = j * RPS + (KSPLIT == 2 ? kh : i) * PSLOTS + ps;
            const int yc = y0 + (yo < nrows ? yo : 0);
            const int pix = (b * H + yc) * W + x0 + r;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
              for (int gp = 0; gp < 2; ++gp) {
                const int cb = p.n0 + (cg * NT + t) * 32 + 8 * (2 * gp + h);
                const int nv = p.N - cb;
                const float* rp = p.R + (pix * (int)p.ldr + (nv > 0 ? cb : 0));
                if (nv >= 8) {
                  const u32x4_a4 q4 = *reinterpret_cast<const u32x4_a4*>(rp), q5 = *reinterpret_cast<const u32x4_a4*>(rp + 4);
                  rpre[i][t][gp][0] = q4.x; rpre[i][t][gp][1] = q4.y; rpre[i][t][gp][2] = q4.z; rpre[i][t][gp][3] = q4.w;
                  rpre[i][t][gp][4] = q5.x; rpre[i][t][gp][5] = q5.y; rpre[i][t][gp][6] = q5.z; rpre[i][t][gp][7] = q5.w;
                } else {
#pragma unroll
                  for (int d = 0; d < 8; ++d) rpre[i][t][gp][d] = (d < nv) ? __float_as_uint(rp[d]) : 0u;
                }
              }
          }
        }
      }
      STAMP_ADD(2);

      f32x16 acc[RPW][NT];
      bool live[RPW];
#pragma unroll
      for (int i = 0; i < RPW; ++i) {
        const int yo = j * RPS + i * PSLOTS + ps;        // output row inside the strip
        live[i] = yo < nrows;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          if (kh == 0) {   // bias = initial accumulator: register group g4 holds channels 8 g4 + 4 h .. + 3
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
              const float4 bq = *reinterpret_cast<const float4*>(biasL + (cg * NT + t) * 32 + 8 * g4 + 4 * h);
              acc[i][t][4 * g4] = bq.x; acc[i][t][4 * g4 + 1] = bq.y; acc[i][t][4 * g4 + 2] = bq.z; acc[i][t][4 * g4 + 3] = bq.w;
            }
          } else {
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][t][v] = 0.f;
          }
        }
        if (live[i]) {
          // one B fragment per (tap, k-step), read PD fragments ahead of the MFMA that consumes it (the wave is alone
          // on its SIMD: nothing else hides the LDS latency)
          constexpr int NSEQ = 9 * KSH, PD = (2 * NT * 9 * KSH > 80) ? 4 : 6, DSTEP = 3;   // (two packs per step: hi and lo)
          const char* base[3];
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) base[ky] = smem + ((yo + ky) % NR) * ROWB + r * PSTRIDE + h * 16 + kh * (KSH * 64);
          auto rd = [&](int idx, int lo) {
            const int ky = idx / (3 * KSH), rem = idx - ky * (3 * KSH), kx = rem / KSH, kk = rem - kx * KSH;
            return *reinterpret_cast<const Pack16*>(base[ky] + kx * PSTRIDE + kk * 64 + 32 * lo);
          };
          Pack16 bq[PD], bql[PD];
#pragma unroll
          for (int u = 0; u < PD; ++u) { bq[u] = rd(u, 0); bql[u] = rd(u, 1); }
#pragma unroll
          for (int idx = 0; idx < NSEQ; ++idx) {
            const int tap = idx / KSH, kk = idx - tap * KSH;
            const Pack16 cur = bq[idx % PD], curl = bql[idx % PD];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
              acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfrag(t, tap, kk, 1), __builtin_bit_cast(bf16x8_t, cur), acc[i][t], 0, 0, 0);
              acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfrag(t, tap, kk, 0), __builtin_bit_cast(bf16x8_t, curl), acc[i][t], 0, 0, 0);
              acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfrag(t, tap, kk, 0), __builtin_bit_cast(bf16x8_t, cur), acc[i][t], 0, 0, 0);
            }
            if (idx + PD < NSEQ) { bq[idx % PD] = rd(idx + PD, 0); bql[idx % PD] = rd(idx + PD, 1); }
            // the next step's rows (they land in slots no wave reads in this step): one DMA piece every DSTEP MFMAs of
            // the wave's first row, so that its issue (60-180 cycles each) runs in the shadow of the matrix pipe
            if (i == 0 && idx % DSTEP == DSTEP - 1 && idx / DSTEP < NPW) {
              constexpr int kq = 0;
              (void)kq;
              const int k = idx / DSTEP;
              const int q = wave + 4 * k;
              if (more && q < RPS * CF::RPIECES) {
                const int rr = q / CF::RPIECES, pi = q - rr * CF::RPIECES;
                dma(sp, (j + 1) * RPS + 2 + rr, pi, loff[k]);
              }
            }
            __builtin_amdgcn_sched_barrier(0);   // keep the read PD fragments ahead of its MFMA (the scheduler pairs them up otherwise)
          }
        }
      }

      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next step's rows have landed (issued >= half a step ago)
      STAMP_ADD(3);
      int fin = 0;                                        // the row (index into acc) this wave finishes
      if constexpr (KSPLIT == 2) {
        // the two K halves of a channel group hold partial sums of the same two rows: half kh gives away row 1 - kh
        // and finishes row kh
        float* xb = reinterpret_cast<float*>(smem + CF::XBUF_OFF);
        float* mine = xb + (size_t)wave * NT * 1024;
        const int partner = (ps * KSPLIT + (1 - kh)) * CW + cg;
        const float* theirs = xb + (size_t)partner * NT * 1024;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int v = 0; v < 16; ++v) mine[(t * 16 + v) * 64 + lane] = kh == 0 ? acc[1][t][v] : acc[0][t][v];
        __syncthreads();
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int v = 0; v < 16; ++v) {
            const float o = theirs[(t * 16 + v) * 64 + lane];
            acc[0][t][v] = (kh == 0 ? acc[0][t][v] : acc[1][t][v]) + o;
          }
        live[0] = kh == 0 ? live[0] : live[1];
        fin = kh;
      }

      STAMP_ADD(4);
      // ---- epilogue -------------------------------------------------------------------------------------------
#pragma unroll
      for (int i = 0; i < (KSPLIT == 2 ? 1 : RPW); ++i) {
        if (!live[i]) continue;
        const int yo = j * RPS + (KSPLIT == 2 ? fin : i) * PSLOTS + ps;
        const int y = y0 + yo, x = x0 + r;

#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int cb0 = p.n0 + (cg * NT + t) * 32;      // first channel of the tile
          if constexpr (PSTORE) {
            // PixelShuffle(2): conv channel n = 4 c' + 2 i + j -> channel c' of output pixel (2y+i, 2x+j).  Register v
            // holds n = 8 (v>>2) + 4h + (v&3): for q = v & 3 the lane owns c' = 2 (v>>2) + h; one swap per register pair
            // gives each lane half 4 consecutive c' of that sub-pixel
            const int cq = cb0 / 4 + 4 * h;
            const int Co = p.N / 4;
            float* ybase = p.Y + (((b * (2 * H) + 2 * y) * (2 * W) + 2 * x) * (int)p.ldy + cq);   // (extent < 2^31 bytes)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const auto s0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[i][t][q]), __float_as_uint(acc[i][t][8 + q]), false, false);
              const auto s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[i][t][4 + q]), __float_as_uint(acc[i][t][12 + q]), false, false);
              float* dst = ybase + ((q >> 1) * 2 * W + (q & 1)) * (int)p.ldy;
              if (cq + 4 <= Co) {
                u32x4_a4 u;
                u.x = s0[0]; u.y = s0[1]; u.z = s1[0]; u.w = s1[1];
                *reinterpret_cast<u32x4_a4*>(dst) = u;
              } else if (cq + 2 <= Co) {
                dst[0] = __uint_as_float(s0[0]); dst[1] = __uint_as_float(s0[1]);
              }
            }
          } else {
            const int pix = (b * H + y) * W + x;   // (extents < 2^31 bytes: 32-bit element offsets)
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
              float c8[8];
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[i][t][8 * gp + e]),
                                                                 __float_as_uint(acc[i][t][8 * gp + 4 + e]), false, false);
                c8[e] = __uint_as_float(sw[0]);
                c8[4 + e] = __uint_as_float(sw[1]);
              }
              const int cb = cb0 + 8 * (2 * gp + h);      // the lane's 8 consecutive channels
              const int nv = p.N - cb;                    // valid channels from cb on (N is even)
              if (nv <= 0) continue;
              if (p.R) {
#pragma unroll
                for (int d = 0; d < 8; ++d) c8[d] += __uint_as_float(rpre[i][t][gp][d]);
              }
              float* yp = p.Y + (pix * (int)p.ldy + cb);
              if (nv >= 8) {
                u32x4_a4 u, u2;
                u.x = __float_as_uint(c8[0]); u.y = __float_as_uint(c8[1]); u.z = __float_as_uint(c8[2]); u.w = __float_as_uint(c8[3]);
                u2.x = __float_as_uint(c8[4]); u2.y = __float_as_uint(c8[5]); u2.z = __float_as_uint(c8[6]); u2.w = __float_as_uint(c8[7]);
                *reinterpret_cast<u32x4_a4*>(yp) = u;
                *reinterpret_cast<u32x4_a4*>(yp + 4) = u2;
              } else {
#pragma unroll
                for (int d = 0; d < 8; ++d)
                  if (d < nv) yp[d] = c8[d];
              }
            }
          }
        }
      }
      STAMP_ADD(5);
      __syncthreads();   // the next step's raw rows are published (every wave waited for its pieces above); this step's reads are done
      if (more) {
        convert_rows((j + 1) * RPS + 2, RPS);
        __syncthreads();
      }
      STAMP_ADD(6);
    }
  }
  if (RDST_DBGV(p.stamps) && tid == 0)
    for (int k = 0; k < 8; ++k) p.stamps[(size_t)blockIdx.x * 8 + k] = tacc[k];
#undef STAMP_ADD
}

template <int K, int CW, int NT, int KSPLIT, int PSLOTS, int RPS, bool UNSHUF, bool PSTORE>
int launch_c3(C3Args& p, int ctile0, hipStream_t st, const char* what) {
  using CF = C3Cfg<K, CW, NT, KSPLIT, PSLOTS, RPS, UNSHUF, PSTORE>;
  p.Wp += (int64_t)ctile0 * 9 * CF::KSTEPS * 512;   // (uint32 units: 2 KB per (tile, tap, k-step))
  p.n0 = ctile0 * 32;
  int SH = 16;
  auto count = [&](int sh) { return (int64_t)p.B * ((p.H + sh - 1) / sh) * (p.W / 32); };
  while (count(SH) < 256 && SH > 2 * RPS) SH /= 2;
  if (SH < RPS) SH = RPS;
  p.SH = SH;
  p.nys = (p.H + SH - 1) / SH;
  const int64_t ns = count(SH);
  if (ns >= (1ll << 31)) return RDST_ENOTSUP;
  p.nstrips = (int)ns;
  const int grid = ns < 256 ? (int)ns : 256;
  auto kern = conv3x_kernel<K, CW, NT, KSPLIT, PSLOTS, RPS, UNSHUF, PSTORE>;
  // (per launch: the attribute is per DEVICE, a process-wide "done" flag would leave a second GPU without it)
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, CF::SMEM);
  p.stamps = rdst_stamps_begin("RDST_C3X_STAMPS", grid, 8, st);
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), CF::SMEM, st, p);
  rdst_stamps_end(what, p.stamps, grid, 8, st);   // 0 weights, 1 first rows, 2 dma issue, 3 mfma, 4 exchange, 5 epilogue, 6 barrier
  return rdst_launch_status(what);
}

int pack(const float* Wc, uint32_t* out, int Cin, int Cout, int K, int N, int mode, float s, hipStream_t st, int cmul = 1, int coff = 0) {
  hipLaunchKernelGGL(conv3x_pack_kernel, dim3((unsigned)conv3x_pack_blocks(K, N)), dim3(256), 0, st, Wc, out, Cin, Cout, K, N,
                     (K + 15) / 16, (N + 31) / 32, mode, s, cmul, coff);
  return rdst_launch_status("conv3x_pack");
}

bool rows_aligned(const void* ptr) { return ((uintptr_t)ptr & 3) == 0; }

}  // namespace

size_t conv3x_pack_bytes(int Cin, int Cout) {
  // the larger of the forward image (K = Cin, N = Cout) and the dgrad image (K = Cout, N = Cin); hi + lo
  const size_t f = (size_t)((Cout + 31) / 32) * 9 * ((Cin + 15) / 16) * 2048;
  const size_t d = (size_t)((Cin + 31) / 32) * 9 * ((Cout + 15) / 16) * 2048;
  return (f > d ? f : d) + 256;
}

int conv3x_fwd_shape(int Cin, int Cout, int ks, int r, bool has_res, int in_act) {
  if (ks != 3 || in_act) return 0;
  if (Cin == 150 && Cout == 60 && r == 1) return 1;
  if (Cin == 60 && Cout == 60 && r == 1) return 2;
  if (Cin == 60 && Cout == 240 && r == 2 && !has_res) return 3;
  return 0;
}

// Forward.  RDST_ENOTSUP = not one of the covered shapes (the caller falls back to conv_mfma.hip).
int conv3x_fwd_f32(const float* X, int64_t ldx, int in_act, const float* Wc, const float* bias, const float* R, int64_t ldr,
                   float* Y, int64_t ldy, const ConvGeom& g, float s, void* wpack, bool prepacked, hipStream_t st) {
  if (!wpack || g.pad != 1 || g.W % 32 || ((uintptr_t)wpack & 15)) return RDST_ENOTSUP;
  const int shape = conv3x_fwd_shape(g.Cin, g.Cout, g.ks, g.r, R != nullptr, in_act);
  if (!shape) return RDST_ENOTSUP;
  if (!rows_aligned(X) || !rows_aligned(Y) || (R && !rows_aligned(R))) return RDST_ENOTSUP;
  uint32_t* wp = reinterpret_cast<uint32_t*>(wpack);
  if (!prepacked)
    if (int rc = pack(Wc, wp, g.Cin, g.Cout, g.Cin, g.Cout, PK_FWD, s, st)) return rc;
  const int64_t abytes = ((g.pixels() - 1) * ldx + g.Cin) * 4;
  const int64_t opix = g.pixels() * g.r * g.r;
  if (abytes >= (1ll << 31) || opix * ldy * 4 >= (1ll << 31) || (R && opix * ldr * 4 >= (1ll << 31))) return RDST_ENOTSUP;
  C3Args p{};
  p.A = X; p.lda = ldx; p.a_bytes = (int)abytes; p.Wp = wp; p.bias = bias; p.R = R; p.ldr = ldr; p.Y = Y; p.ldy = ldy;
  p.B = g.B; p.H = g.H; p.W = g.W; p.N = g.Cout; p.s = s; p.ymul = 1; p.xmul = 1;
  if (shape == 1) return launch_c3<150, 2, 1, 2, 1, 2, false, false>(p, 0, st, "conv3x_fwd_150_60");
  if (shape == 2) return launch_c3<60, 2, 1, 1, 2, 2, false, false>(p, 0, st, "conv3x_fwd_60_60");
  C3Args pa = p;
  if (int rc = launch_c3<60, 4, 1, 1, 1, 2, false, true>(pa, 0, st, "conv3x_fwd_60_240_ps_a")) return rc;
  return launch_c3<60, 4, 1, 1, 1, 2, false, true>(p, 4, st, "conv3x_fwd_60_240_ps_b");
}

// dgrad: dX = dX_add + s * conv^T(dY).  dY is in the OUTPUT geometry: for conv + PixelShuffle(2) the shuffled tensor as it lies — four
// launches of the 60 -> 60 form, one per sub-pixel q = 2 i + j (dY through a stride-2 pixel view, the weights of conv channels
// 4 c' + q), each accumulating onto the last.
int conv3x_dgrad_f32(const float* Wc, const float* dY, int64_t lddy, float* dX, int64_t lddx, const float* acc, int64_t ldacc,
                     int in_act, const ConvGeom& g, float s, void* wpack, hipStream_t st) {
  if (!wpack || g.ks != 3 || g.pad != 1 || in_act || g.W % 32 || ((uintptr_t)wpack & 15)) return RDST_ENOTSUP;
  if (!rows_aligned(dY) || !rows_aligned(dX) || (acc && !rows_aligned(acc))) return RDST_ENOTSUP;
  int shape = 0;
  if (g.Cin == 150 && g.Cout == 60 && g.r == 1) shape = 1;
  else if (g.Cin == 60 && g.Cout == 60 && g.r == 1) shape = 2;
  else if (g.Cin == 60 && g.Cout == 240 && g.r == 2) shape = 3;
  if (!shape) return RDST_ENOTSUP;
  const int64_t abytes = ((g.pixels() * g.r * g.r - 1) * lddy + g.Cout / (g.r * g.r)) * 4;
  if (abytes >= (1ll << 31) || g.pixels() * lddx * 4 >= (1ll << 31) || (acc && g.pixels() * ldacc * 4 >= (1ll << 31))) return RDST_ENOTSUP;
  uint32_t* wp = reinterpret_cast<uint32_t*>(wpack);
  C3Args p{};
  p.A = dY; p.lda = lddy; p.a_bytes = (int)abytes; p.Wp = wp; p.bias = nullptr; p.R = acc; p.ldr = ldacc; p.Y = dX; p.ldy = lddx;
  p.B = g.B; p.H = g.H; p.W = g.W; p.N = g.Cin; p.s = s; p.ymul = 1; p.xmul = 1;
  if (shape == 3) {
    constexpr int IMG = 2 * 9 * 4 * 512;   // uint32 per sub-pixel image: 2 channel tiles x 9 taps x 4 k-steps x 2 KB
    for (int q = 0; q < 4; ++q)
      if (int rc = pack(Wc, wp + (size_t)q * IMG, g.Cin, g.Cout, 60, g.Cin, PK_DGRAD, s, st, 4, q)) return rc;
    for (int q = 0; q < 4; ++q) {
      C3Args pq = p;
      pq.Wp = wp + (size_t)q * IMG; pq.ymul = 2; pq.yoff = q >> 1; pq.xmul = 2; pq.xoff = q & 1;
      if (q > 0) { pq.R = dX; pq.ldr = lddx; }
      if (int rc = launch_c3<60, 2, 1, 1, 2, 2, false, false>(pq, 0, st, "conv3x_dgrad_240_60_q")) return rc;
    }
    return 0;
  }
  if (int rc = pack(Wc, wp, g.Cin, g.Cout, g.Cout, g.Cin, PK_DGRAD, s, st)) return rc;
  if (shape == 1) {
    C3Args pa = p;
    if (int rc = launch_c3<60, 4, 1, 1, 1, 2, false, false>(pa, 0, st, "conv3x_dgrad_60_150a")) return rc;
    return launch_c3<60, 1, 1, 1, 4, 4, false, false>(p, 4, st, "conv3x_dgrad_60_150b");
  }
  return launch_c3<60, 2, 1, 1, 2, 2, false, false>(p, 0, st, "conv3x_dgrad_60_60");
}
