// K4/K5 on the gfx950 matrix cores: 3x3 (and 1x1) convolution on token-major rows as an implicit GEMM
// without im2col: forward, dgrad (= the same kernel with the weights read transposed and the taps
// mirrored) and wgrad.
//
// forward / dgrad: persistent 8-wave workgroups.  A chunk of output channels x ALL taps of the weight
// tensor is converted to the compute type once per workgroup and stays in LDS ([tap][col][ci],
// ci-contiguous, padded so ds_read_b128 is conflict-free); each wave walks 32-pixel slabs: for every
// tap it loads the neighbour pixel's channel row straight into MFMA fragment shape (zero outside the
// image, the activation of the '3conv' variant applied in registers) and multiplies it against every
// column tile of the chunk; accumulators live across the 9 taps.  Epilogue: bias, scale, residual,
// PixelShuffle addressing (fwd) or activation gradient / accumulate (dgrad) on the 32x32 tile.
// A pixel-shuffled dY is un-shuffled once into scratch so dgrad and wgrad read plain rows.
//
// wgrad: contraction over pixels, one workgroup per (pixel range, kernel row ky); stripes of 32 pixels
// of dY and of the three kx-shifted input rows are staged in LDS and read transposed
// (ds_read_b64_tr_b16) / element-wise (fp32).  d(bias) rides on a ones column of the centre tap.
#include "conv.h"
#include "mfma.h"
#include <stdlib.h>

int slab_reduce(const float* slab, float* out, int S, int64_t n, hipStream_t st);

namespace {

bool mfma_disabled() {
  static int v = -1;
  if (v < 0) {
    const char* e = rdst_dbg_getenv("RDST_DISABLE_MFMA");
    v = (e && e[0] == '1') ? 1 : 0;
  }
  return v == 1;
}

constexpr int CMODE_FWD = 0, CMODE_DGRAD = 1;
constexpr int CV_MAXCT = 3;  // column tiles per chunk (96 output channels)

template <typename T>
struct ConvArgs {
  const T* A; int64_t lda; int CA;   // rows contracted per tap: X (fwd, CA = Cin) / dY (dgrad, CA = Cout)
  const float* Wc;                   // (Cout, Cin, ks, ks)
  const float* bias;
  const T* R; int64_t ldr;
  T* Y; int64_t ldy;                 // fwd: Y (output geometry), dgrad: dX
  const T* Xa; int64_t ldxa;         // dgrad: X for act'
  int in_act; const T* Acc; int64_t ldacc;   // dgrad: + dX_add
  ConvGeom g;
  int Nout;                          // fwd: Cout, dgrad: Cin
  float s;
  int Tn, ldw, nch;
  int eps_off;
  int dbg;   // RDST_CONV_DEBUG ablation: 1 skip tile loads, 2 skip MFMAs, 4 skip stores, 8 skip the slab loop
};

template <typename T, int TMAX, int MODE, bool SP = false>
__global__ void __launch_bounds__(512) conv_mfma_kernel(const ConvArgs<T> p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using MM = Mma<T, SP>;   // SP: RDST_F32X3 (mfma.h) — weights split as they are staged, a pixel row's fragments once per tap
  constexpr int KP = MM::KP, HP = MM::HP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  const int Tn = p.Tn;
  const ConvGeom g = p.g;
  const int ntap = g.ks * g.ks;
  const int64_t P = g.pixels();
  const int64_t nslabs = (P + 31) / 32;
  float* eps = reinterpret_cast<float*>(smem + (p.eps_off < 0 ? 0 : p.eps_off)) + wave * 1024;  // wave-private epilogue tile

  for (int n0 = 0; n0 < p.Nout; n0 += p.nch) {
    __syncthreads();
    const int nc = (p.Nout - n0 < p.nch) ? p.Nout - n0 : p.nch;
    const int ncp = ((nc + 31) / 32) * 32;
    // stage W[tap][col][ci-pack] for this chunk
    stage_packs_batched<T, 4, SP>(ntap * ncp * 2 * Tn, p.CA, MODE == CMODE_FWD ? (int64_t)ntap : (int64_t)g.Cin * ntap, tid, 512,
                              [&](int idx, const float*& src, int& k0, char*& dst, bool& ok) {
                                const int ph = idx % (2 * Tn);
                                const int rest = idx / (2 * Tn);
                                const int n = rest % ncp, tap = rest / ncp;
                                ok = n < nc;
                                k0 = ph * HP;
                                // fwd:   B[k = ci][col = co] = Wc[co][ci][tap]: stride over ci is ks*ks
                                // dgrad: contraction over co, output col = ci, mirrored tap: Wc[co][ci][ntap-1-tap]
                                src = MODE == CMODE_FWD
                                          ? p.Wc + ((int64_t)(n0 + n) * g.Cin + k0) * ntap + tap
                                          : p.Wc + ((int64_t)k0 * g.Cin + (n0 + n)) * ntap + (ntap - 1 - tap);
                                dst = smem + ((size_t)tap * ncp + n) * p.ldw + ph * 16;
                              });
    __syncthreads();
    const int nct = ncp / 32;

    for (int64_t slab = (int64_t)blockIdx.x * 8 + wave; slab < nslabs; slab += (int64_t)gridDim.x * 8) {
      const int64_t pix = slab * 32 + r;
      const bool pvalid = pix < P;
      int b, y, x;
      g.decode(pvalid ? pix : 0, b, y, x);
      f32x16 acc[CV_MAXCT];
#pragma unroll
      for (int c = 0; c < CV_MAXCT; ++c)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[c][v] = 0.f;
      for (int tap = 0; tap < ntap; ++tap) {
        const int ky = tap / g.ks, kx = tap - ky * g.ks;
        const int yy = y + ky - g.pad, xx = x + kx - g.pad;
        const bool valid = pvalid && yy >= 0 && yy < g.H && xx >= 0 && xx < g.W;
        const T* arow = p.A + (valid ? (((int64_t)b * g.H + yy) * g.W + xx) : 0) * p.lda;
        Pack16 a[TMAX];
#pragma unroll
        for (int t = 0; t < TMAX; ++t)
          if (t < Tn) a[t] = load_pack<T>(arow, t * KP + h * HP, p.CA, valid);
        if (MODE == CMODE_FWD && p.in_act) {
#pragma unroll
          for (int t = 0; t < TMAX; ++t)
            if (t < Tn) {
              float f[HP];
              MM::unpack(a[t], f);
#pragma unroll
              for (int e = 0; e < HP; ++e) f[e] = apply_act(f[e], p.in_act);
              a[t] = MM::pack(f);
            }
        }
        if constexpr (MM::SPLIT) {
#pragma unroll
          for (int t = 0; t < TMAX; ++t)
            if (t < Tn) a[t] = MM::op(a[t]);
        }
        const char* wtap = smem + ((size_t)tap * ncp + r) * p.ldw + h * 16;
#pragma unroll
        for (int c = 0; c < CV_MAXCT; ++c)
          if (c < nct) {
            const char* wrow = wtap + (size_t)c * 32 * p.ldw;
#pragma unroll
            for (int t = 0; t < TMAX; ++t)
              if (t < Tn) {
                const Pack16 bb = *reinterpret_cast<const Pack16*>(wrow + t * 32);
                MM::mma_da(acc[c], a[t], bb);   // (split mode: the pixel fragment is the one kept across the column tiles)
              }
          }
      }
      // epilogue
#pragma unroll
      for (int c = 0; c < CV_MAXCT; ++c)
        if (c < nct) {
          const int col = n0 + c * 32 + r;
          const float bv = (MODE == CMODE_FWD && p.bias && col < p.Nout) ? p.bias[col] : 0.f;
          if (p.eps_off < 0 || (MODE == CMODE_FWD && g.r > 1)) {
            // element stores: PixelShuffle scatters consecutive channels to different pixels, or no LDS
            // is left for the row-wise bounce (240-channel dgrad)
            if (col < p.Nout) {
#pragma unroll
              for (int v = 0; v < 16; ++v) {
                const int64_t pp = slab * 32 + acc_row(v, h);
                if (pp < P) {
                  if (MODE == CMODE_FWD) {
                    int b2, y2, x2;
                    g.decode(pp, b2, y2, x2);
                    int64_t row; int ch;
                    g.out_rc(b2, y2, x2, col, row, ch);
                    float val = (acc[c][v] + bv) * p.s;
                    if (p.R) val += to_f32<T>(p.R[row * p.ldr + ch]);
                    p.Y[row * p.ldy + ch] = from_f32<T>(val);
                  } else {
                    float val = acc[c][v] * p.s;
                    if (p.in_act) val *= act_grad(to_f32<T>(p.Xa[pp * p.ldxa + col]), p.in_act);
                    if (p.Acc) val += to_f32<T>(p.Acc[pp * p.ldacc + col]);
                    p.Y[pp * p.ldy + col] = from_f32<T>(val);
                  }
                }
              }
            }
          } else {
            float vals[16];
#pragma unroll
            for (int v = 0; v < 16; ++v) vals[v] = (MODE == CMODE_FWD) ? (acc[c][v] + bv) * p.s : acc[c][v] * p.s;
            TileEpilogue ep{};
            if (MODE == CMODE_FWD) {
              ep.R = p.R; ep.ldr = p.ldr; ep.Y = p.Y; ep.ldy = p.ldy;
            } else {
              ep.Xa = p.Xa; ep.ldxa = p.ldxa; ep.act = p.in_act; ep.Y = p.Y; ep.ldy = p.ldy; ep.Acc = p.Acc; ep.ldacc = p.ldacc;
            }
            tile_store_rows<T>(eps, vals, lane, slab * 32, P, n0 + c * 32, p.Nout, ep);
          }
        }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// forward / dgrad for 3x3 / pad 1 convs whose image width is a multiple of 32: the row-stripe form.
// A slab is 32 consecutive pixels of ONE image row.  For each kernel row ky the 34 neighbour pixels
// x0-1 .. x0+32 of image row y+ky-1 are read ONCE, coalesced (8 lanes x 16 B of a pixel, 8 pixels per
// instruction), in 128-B channel chunks, into a wave-private LDS tile; the three kx taps are row offsets
// of the fragment reads (the generic kernel above re-reads every neighbour row per tap in fragment
// shape: 9x the load instructions at 32 B per row).  Accumulators are kept TRANSPOSED (output channel in
// the registers, pixel on the lane) and leave as 16-B row stores straight from the registers; for
// PixelShuffle(2) the four sub-pixels of a lane are 8-B runs of 4 consecutive output channels.
// The next tile's chunks are prefetched while the current one is multiplied.
// ------------------------------------------------------------------------------------------------
template <typename T, int MODE, int PF, bool SP = false>
__global__ void __launch_bounds__(512) conv_rows_kernel(const ConvArgs<T> p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using MM = Mma<T, SP>;   // SP: RDST_F32X3 — the scattered weight image is split in place, a tile fragment when it is read
  constexpr int KP = MM::KP, HP = MM::HP;
  constexpr bool BF = sizeof(T) == 2;
  constexpr int TLD = 144;          // tile row: 128 B of the pixel's channels + 16 B pad (odd 16-B slots)
  constexpr int TROWS = 34;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  const int Tn = p.Tn;
  const ConvGeom g = p.g;
  const int64_t P = g.pixels();
  const int64_t nslabs = P / 32;    // W % 32 == 0
  char* tile = smem + p.eps_off + wave * (TROWS * TLD);
  float* biasL = reinterpret_cast<float*>(smem + p.eps_off + 8 * TROWS * TLD);
  const int npad = ((p.Nout + 31) / 32) * 32;
  if (MODE == CMODE_FWD)
    for (int i = tid; i < npad; i += 512) biasL[i] = (p.bias && i < p.Nout) ? p.bias[i] : 0.f;
  const int rowbytes = p.CA * (int)sizeof(T);
  const int nkc = (rowbytes + 127) / 128;
  const int crow = lane >> 3, cchk = lane & 7;

  // raw chunk (ky, kc) of the slab at (b, y, x0): 5 instructions x 8 tile rows
  auto issue = [&](Pack16 (&rw)[5], int b, int y, int x0, int ky, int kc) {
    if (RDST_DBGV(p.dbg) & 1) return;
    int yy = y + ky - 1;
    yy = yy < 0 ? 0 : (yy >= g.H ? g.H - 1 : yy);        // rows outside the image are skipped by the consumer
    int off = kc * 128 + cchk * 16;
    if (off + 16 > rowbytes) off = rowbytes - 16;
    const T* rowbase = p.A + (((int64_t)b * g.H + yy) * g.W) * p.lda;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      int j = 8 * i + crow;                               // tile row: pixel x0 - 1 + j
      j = j > TROWS - 1 ? TROWS - 1 : j;
      int xx = x0 - 1 + j;
      xx = xx < 0 ? 0 : (xx >= g.W ? g.W - 1 : xx);      // the image's edge columns are zeroed when the tile is written
      const u32x4_a4 v = *reinterpret_cast<const u32x4_a4*>(reinterpret_cast<const char*>(rowbase + (int64_t)xx * p.lda) + off);
      rw[i].w[0] = v.x; rw[i].w[1] = v.y; rw[i].w[2] = v.z; rw[i].w[3] = v.w;
    }
  };

  for (int n0 = 0; n0 < p.Nout; n0 += p.nch) {
    __syncthreads();
    const int nc = (p.Nout - n0 < p.nch) ? p.Nout - n0 : p.nch;
    const int ncp = ((nc + 31) / 32) * 32;
    // weights of this column chunk: LDS image [tap][column n][contraction k], k contiguous; the tap stride is
    // padded by one 16-B slot so the nine taps of the scattered stores fall on different banks
    const int tapst = ncp * p.ldw + 16;
    if (!(RDST_DBGV(p.dbg) & 32)) lds_zero16(smem, 9 * tapst, tid, 512);
    __syncthreads();
    if (RDST_DBGV(p.dbg) & 16) {
    } else if (MODE == CMODE_FWD) {
      // Wc[co][ci][tap]: rows co = n0 .. n0+nc-1, CA * 9 floats of each (CA < Cin: a launch over a slice of the input
      // channels, p.Wc then points at the slice's first channel)
      stage_scatter<T>(p.Wc + (int64_t)n0 * g.Cin * 9, nc, p.CA * 9, (int64_t)g.Cin * 9, tid, 512, smem, [&](int n, int j) {
        const int ci = j / 9, tap = j - ci * 9;
        return tap * tapst + n * p.ldw + ci * (int)sizeof(T);
      });
    } else {
      // dgrad: contraction over co, output column = ci, mirrored tap: per co the segment ci = n0 .. n0+nc-1
      stage_scatter<T>(p.Wc + (int64_t)n0 * 9, p.CA, nc * 9, (int64_t)g.Cin * 9, tid, 512, smem, [&](int co, int j) {
        const int n = j / 9, tap = j - n * 9;
        return (8 - tap) * tapst + n * p.ldw + co * (int)sizeof(T);
      });
    }
    __syncthreads();
    if constexpr (MM::SPLIT) {
      const int ppr = 2 * Tn;
      for (int i = tid; i < 9 * ncp * ppr; i += 512) {
        const int ph = i % ppr, rest = i / ppr;
        const int n = rest % ncp, tap = rest / ncp;
        Pack16* q = reinterpret_cast<Pack16*>(smem + (size_t)tap * tapst + (size_t)n * p.ldw + ph * 16);
        *q = MM::op(*q);
      }
      __syncthreads();
    }
    const int nct = ncp / 32;

    for (int64_t slab = (int64_t)blockIdx.x * 8 + wave; slab < ((RDST_DBGV(p.dbg) & 8) ? 0 : nslabs); slab += (int64_t)gridDim.x * 8) {
      int b, y, x0;
      g.decode(slab * 32, b, y, x0);
      f32x16 acc[CV_MAXCT];
#pragma unroll
      for (int c = 0; c < CV_MAXCT; ++c) {
        if (MODE == CMODE_FWD && c < nct) {  // bias = initial accumulator: register group g4 holds channels 8 g4 + 4h .. +3
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const float4 bq = *reinterpret_cast<const float4*>(biasL + n0 + c * 32 + 8 * g4 + 4 * h);
            acc[c][4 * g4] = bq.x; acc[c][4 * g4 + 1] = bq.y; acc[c][4 * g4 + 2] = bq.z; acc[c][4 * g4 + 3] = bq.w;
          }
        } else {
#pragma unroll
          for (int v = 0; v < 16; ++v) acc[c][v] = 0.f;
        }
      }
      // PF chunks are in flight ahead of the one being multiplied (a step's MFMAs are short against the load
      // latency and only two waves share a SIMD)
      const int nsteps = 3 * nkc;
      Pack16 raw[PF][5];
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int st0 = u < nsteps ? u : nsteps - 1;
        const int k0y = st0 / nkc;
        issue(raw[u], b, y, x0, k0y, st0 - k0y * nkc);
      }
      for (int sg = 0; sg < nsteps; sg += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
          const int step = sg + u;
          if (step >= nsteps) break;
          const int ky = step / nkc, kc = step - ky * nkc;
          // raw -> tile (edge columns zeroed, input activation applied once), then the chunk PF steps ahead goes in flight
          {
            int off = kc * 128 + cchk * 16;
            const bool act = off < rowbytes;
            if (off + 16 > rowbytes) off = rowbytes - 16;
            const int loc = off - kc * 128;
#pragma unroll
            for (int i = 0; i < 5; ++i) {
              const int j = 8 * i + crow;
              if (j < TROWS && act) {
                const int xx = x0 - 1 + j;
                Pack16 q = raw[u][i];
                if (xx < 0 || xx >= g.W) { q.w[0] = 0u; q.w[1] = 0u; q.w[2] = 0u; q.w[3] = 0u; }
                else if (MODE == CMODE_FWD && p.in_act) {
                  float f[HP];
                  MM::unpack(q, f);
#pragma unroll
                  for (int e = 0; e < HP; ++e) f[e] = apply_act(f[e], p.in_act);
                  q = MM::pack(f);
                }
                *reinterpret_cast<Pack16*>(tile + j * TLD + loc) = q;
              }
            }
          }
          {
            const int ns = step + PF < nsteps ? step + PF : step;   // always redefines the whole set
            const int nky = ns / nkc;
            issue(raw[u], b, y, x0, nky, ns - nky * nkc);
          }
          const int yy = y + ky - 1;
          if (yy < 0 || yy >= g.H) continue;                       // zero padding: this kernel row adds nothing
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const int tap = ky * 3 + kx;
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
              const int t = 4 * kc + tt;
              if (t < Tn) {
                Pack16 a = *reinterpret_cast<const Pack16*>(tile + (r + kx) * TLD + tt * 32 + h * 16);
                if (t == Tn - 1) {   // elements past the last channel inside the last k-step: zero (may be NaN bits)
                  const int kl = t * KP + h * HP;
                  if (kl + HP > p.CA) {
                    float f[HP];
                    MM::unpack(a, f);
#pragma unroll
                    for (int e = 0; e < HP; ++e) f[e] = (kl + e < p.CA) ? f[e] : 0.f;
                    a = MM::pack(f);
                  }
                }
                a = MM::op(a);
                const char* wrow = smem + (size_t)tap * tapst + (size_t)r * p.ldw + t * 32 + h * 16;
#pragma unroll
                for (int c = 0; c < CV_MAXCT; ++c)
                  if (c < nct) {
                    const Pack16 wa = *reinterpret_cast<const Pack16*>(wrow + (size_t)c * 32 * p.ldw);
                    if (!(RDST_DBGV(p.dbg) & 2)) MM::mma(acc[c], wa, a);   // rows = output channels, cols = pixels
                  }
              }
            }
          }
        }
      }
      // epilogue
      if (RDST_DBGV(p.dbg) & 4) continue;
      const int64_t pix = slab * 32 + r;
      const int x = x0 + r;
#pragma unroll
      for (int c = 0; c < CV_MAXCT; ++c)
        if (c < nct) {
          if (MODE == CMODE_FWD && g.r == 2) {
            // PixelShuffle(2): conv channel n = 4 c' + 2 i + j -> channel c' of output pixel (2y+i, 2x+j).  Register
            // v holds n = 8 (v>>2) + 4h + (v&3): for q = v & 3 = 2i + j the lane owns c' = 2 (v>>2) + h; one swap per
            // register gives each lane half 4 consecutive c' of that sub-pixel.
            const int cq = (n0 + c * 32) / 4 + 4 * h;            // first of the lane's 4 output channels
            const int Co = g.Cout / 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const auto s0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[c][q] * p.s), __float_as_uint(acc[c][8 + q] * p.s), false, false);
              const auto s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[c][4 + q] * p.s), __float_as_uint(acc[c][12 + q] * p.s), false, false);
              const float o0 = __uint_as_float(s0[0]), o1 = __uint_as_float(s0[1]), o2 = __uint_as_float(s1[0]), o3 = __uint_as_float(s1[1]);
              const int i = q >> 1, j = q & 1;
              const int64_t orow = ((int64_t)b * (2 * g.H) + 2 * y + i) * (int64_t)(2 * g.W) + 2 * x + j;
              T* dst = p.Y + orow * p.ldy + cq;
              if (cq + 4 <= Co && (reinterpret_cast<uintptr_t>(dst) & 3) == 0) {
                if (BF) {
                  u32x2_a4 u;
                  u.x = pack_bf16x2(o0, o1); u.y = pack_bf16x2(o2, o3);
                  *reinterpret_cast<u32x2_a4*>(dst) = u;
                } else {
                  u32x4_a4 u;
                  u.x = __float_as_uint(o0); u.y = __float_as_uint(o1); u.z = __float_as_uint(o2); u.w = __float_as_uint(o3);
                  *reinterpret_cast<u32x4_a4*>(dst) = u;
                }
              } else {
                const float o[4] = {o0, o1, o2, o3};
#pragma unroll
                for (int e = 0; e < 4; ++e) if (cq + e < Co) dst[e] = from_f32<T>(o[e]);
              }
            }
            continue;
          }
#pragma unroll
          for (int gp = 0; gp < 2; ++gp) {
            float c8[8];
            int cb;
            if (BF) {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[c][8 * gp + e] * p.s),
                                                                 __float_as_uint(acc[c][8 * gp + 4 + e] * p.s), false, false);
                c8[e] = __uint_as_float(sw[0]);
                c8[4 + e] = __uint_as_float(sw[1]);
              }
              cb = n0 + c * 32 + 8 * (2 * gp + h);
            } else {
#pragma unroll
              for (int e = 0; e < 8; ++e) c8[e] = acc[c][8 * gp + e] * p.s;
              cb = n0 + c * 32 + 16 * gp + 4 * h;
            }
            auto colof = [&](int e) { return BF ? cb + e : cb + (e & 3) + 8 * (e >> 2); };
            auto chunk8 = [&](const T* rowp, float (&g8)[8]) {
              if (BF && cb + 8 <= p.Nout && (reinterpret_cast<uintptr_t>(rowp + cb) & 3) == 0) {
                const u32x4_a4 u = *reinterpret_cast<const u32x4_a4*>(rowp + cb);
                g8[0] = bf16lo(u.x); g8[1] = bf16hi(u.x); g8[2] = bf16lo(u.y); g8[3] = bf16hi(u.y);
                g8[4] = bf16lo(u.z); g8[5] = bf16hi(u.z); g8[6] = bf16lo(u.w); g8[7] = bf16hi(u.w);
              } else if (!BF && cb + 12 <= p.Nout && (reinterpret_cast<uintptr_t>(rowp) & 3) == 0) {
                // fp32: the lane's two runs of 4 channels as 16-byte chunks of its dword-aligned row (element by element every
                // load instruction touched 64 cache lines for 4 bytes each)
#pragma unroll
                for (int q4 = 0; q4 < 2; ++q4) {
                  const u32x4_a4 u = *reinterpret_cast<const u32x4_a4*>(rowp + cb + 8 * q4);
                  g8[4 * q4] = __uint_as_float(u.x); g8[4 * q4 + 1] = __uint_as_float(u.y);
                  g8[4 * q4 + 2] = __uint_as_float(u.z); g8[4 * q4 + 3] = __uint_as_float(u.w);
                }
              } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) g8[e] = colof(e) < p.Nout ? to_f32<T>(rowp[colof(e)]) : 0.f;
              }
            };
            if (MODE == CMODE_FWD) {
              if (p.R) {
                float g8[8];
                chunk8(p.R + pix * p.ldr, g8);
#pragma unroll
                for (int e = 0; e < 8; ++e) c8[e] += g8[e];
              }
            } else {
              if (p.in_act) {
                float g8[8];
                chunk8(p.Xa + pix * p.ldxa, g8);
#pragma unroll
                for (int e = 0; e < 8; ++e) c8[e] *= act_grad(g8[e], p.in_act);
              }
              if (p.Acc) {
                float g8[8];
                chunk8(p.Acc + pix * p.ldacc, g8);
#pragma unroll
                for (int e = 0; e < 8; ++e) c8[e] += g8[e];
              }
            }
            T* yrow = p.Y + pix * p.ldy;
            if (BF && cb + 8 <= p.Nout && (reinterpret_cast<uintptr_t>(yrow + cb) & 3) == 0) {
              u32x4_a4 u;
              u.x = pack_bf16x2(c8[0], c8[1]); u.y = pack_bf16x2(c8[2], c8[3]);
              u.z = pack_bf16x2(c8[4], c8[5]); u.w = pack_bf16x2(c8[6], c8[7]);
              *reinterpret_cast<u32x4_a4*>(yrow + cb) = u;
            } else if (!BF && cb + 12 <= p.Nout && (reinterpret_cast<uintptr_t>(yrow) & 3) == 0) {
#pragma unroll
              for (int q4 = 0; q4 < 2; ++q4) {
                u32x4_a4 u;
                u.x = __float_as_uint(c8[4 * q4]); u.y = __float_as_uint(c8[4 * q4 + 1]);
                u.z = __float_as_uint(c8[4 * q4 + 2]); u.w = __float_as_uint(c8[4 * q4 + 3]);
                *reinterpret_cast<u32x4_a4*>(yrow + cb + 8 * q4) = u;
              }
            } else {
#pragma unroll
              for (int e = 0; e < 8; ++e) if (colof(e) < p.Nout) yrow[colof(e)] = from_f32<T>(c8[e]);
            }
          }
        }
    }
  }
}

// row-stripe kernel: 3x3 / pad 1, W % 32 == 0, PixelShuffle factor 1 or 2 (2: forward, no residual),
// dword-aligned rows of at least 16 B whose last 128-B chunk is not shorter than 16 B
template <typename T, int MODE>
int launch_conv_rows(ConvArgs<T>& p, hipStream_t st, const char* what) {
  using MM = Mma<T>;
  const ConvGeom& g = p.g;
  if (g.ks != 3 || g.pad != 1 || g.W % 32 != 0 || rdst_dbg_getenv("RDST_CONV_V1")) return RDST_ENOTSUP;
  if (MODE == CMODE_FWD && !(g.r == 1 || (g.r == 2 && !p.R && g.Cout % 4 == 0))) return RDST_ENOTSUP;
  const int64_t rowbytes = (int64_t)p.CA * (int64_t)sizeof(T);
  if (((uintptr_t)p.A & 3) || (p.lda * sizeof(T)) % 4 || rowbytes % 4 || rowbytes < 16 || !(rowbytes % 128 == 0 || rowbytes % 128 >= 16))
    return RDST_ENOTSUP;
  p.Tn = (p.CA + MM::KP - 1) / MM::KP;
  if (p.Tn > 16) return RDST_ENOTSUP;
  p.ldw = lds_row_bytes(p.CA, sizeof(T));
  const int npad = ((p.Nout + 31) / 32) * 32;
  const size_t extra = (size_t)8 * 34 * 144 + (size_t)npad * sizeof(float);
  int nch = (int)((160 * 1024 - extra - 9 * 16) / ((size_t)9 * p.ldw)) / 32 * 32;
  if (nch > 32 * CV_MAXCT) nch = 32 * CV_MAXCT;
  if (nch < 32) return RDST_ENOTSUP;
  if (nch > npad) nch = npad;
  p.nch = nch;
  p.eps_off = 9 * (nch * p.ldw + 16);   // here: offset of the wave tiles
  { const char* e = rdst_dbg_getenv("RDST_CONV_DEBUG"); p.dbg = e ? atoi(e) : 0; }
  const size_t smem = (size_t)p.eps_off + extra;
  const int64_t nslabs = g.pixels() / 32;
  int64_t grid = (nslabs + 7) / 8;
  if (grid > 256) grid = 256;
  static int pf = -1;
  if (pf < 0) { const char* e = rdst_dbg_getenv("RDST_CONV_PF"); pf = e ? atoi(e) : 3; }
  auto kern = pf == 1 ? conv_rows_kernel<T, MODE, 1> : pf == 2 ? conv_rows_kernel<T, MODE, 2> : conv_rows_kernel<T, MODE, 3>;
  if constexpr (sizeof(T) == 4)
    if (rdst_split()) kern = pf == 1 ? conv_rows_kernel<T, MODE, 1, true> : pf == 2 ? conv_rows_kernel<T, MODE, 2, true> : conv_rows_kernel<T, MODE, 3, true>;
  if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), smem, st, p);
  return rdst_launch_status(what);
}

template <typename T, int MODE>
int launch_conv(ConvArgs<T>& p, hipStream_t st, const char* what) {
  using MM = Mma<T>;
  {
    const int rc = launch_conv_rows<T, MODE>(p, st, what);
    if (rc != RDST_ENOTSUP) return rc;
  }
  p.Tn = (p.CA + MM::KP - 1) / MM::KP;
  if (p.Tn > 16) return RDST_ENOTSUP;
  p.ldw = lds_row_bytes(p.CA, sizeof(T));
  const int ntap = p.g.ks * p.g.ks;
  const int npad = ((p.Nout + 31) / 32) * 32;
  bool rows = true;  // row-wise epilogue needs 8 x 4 KB of LDS besides the weights
  int nch = (int)((124 * 1024) / ((size_t)ntap * p.ldw)) / 32 * 32;
  if (nch < 32) {
    rows = false;
    nch = (int)((156 * 1024) / ((size_t)ntap * p.ldw)) / 32 * 32;
  }
  if (nch > 32 * CV_MAXCT) nch = 32 * CV_MAXCT;
  if (nch < 32) return RDST_ENOTSUP;
  if (nch > npad) nch = npad;
  p.nch = nch;
  p.eps_off = rows ? ntap * nch * p.ldw : -1;
  const size_t smem = (size_t)ntap * nch * p.ldw + (rows ? 8 * 4096 : 0);
  const int64_t nslabs = (p.g.pixels() + 31) / 32;
  int64_t grid = (nslabs + 7) / 8;
  if (grid > 256) grid = 256;
#define RDST_CONV_LAUNCH(TM)                                                                                         \
  {                                                                                                                  \
    auto kern = conv_mfma_kernel<T, TM, MODE>;                                                                       \
    if constexpr (sizeof(T) == 4)                                                                                    \
      if (rdst_split()) kern = conv_mfma_kernel<T, TM, MODE, true>;                                                  \
    if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); \
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), smem, st, p);                                          \
  }
  if (p.Tn <= 4) RDST_CONV_LAUNCH(4) else if (p.Tn <= 8) RDST_CONV_LAUNCH(8) else RDST_CONV_LAUNCH(16)
#undef RDST_CONV_LAUNCH
  return rdst_launch_status(what);
}

template <typename T> bool rows_ok(const void*, int64_t) { return true; }  // load_pack checks alignment per access

// dY (B, H*r, W*r, C/r^2) pixel-shuffled rows -> plain (B*H*W, C) rows in nn.PixelShuffle channel order
template <typename T>
__global__ void __launch_bounds__(256) unshuffle_kernel(const T* __restrict__ dY, int64_t ld, T* __restrict__ out,
                                                        ConvGeom g) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // over pixels * Cout, co fastest
  const int64_t tot = g.pixels() * g.Cout;
  if (i >= tot) return;
  const int co = (int)(i % g.Cout);
  const int64_t pix = i / g.Cout;
  int b, y, x;
  g.decode(pix, b, y, x);
  int64_t row; int c;
  g.out_rc(b, y, x, co, row, c);
  out[i] = dY[row * ld + c];
}

// PixelShuffle(2), fp32: one thread builds the 4 conv channels 4c .. 4c+3 of one input pixel (output channel c of the four
// sub-pixels) from four 4-byte reads (consecutive threads: consecutive c of the same four rows) and writes them with one
// 16-byte store (the element-wise kernel above pays two 64-bit divisions per element: 296 us per call at 128 x 128 x 240)
__global__ void __launch_bounds__(256) unshuffle2_f32_kernel(const float* __restrict__ dY, int64_t ld, float* __restrict__ out,
                                                             ConvGeom g) {
  const int Co = g.Cout / 4;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= g.pixels() * Co) return;
  const int c = (int)(i % Co);
  const int64_t pix = i / Co;
  int b, y, x;
  g.decode(pix, b, y, x);
  u32x4_a4 o;
  uint32_t v[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int64_t row = ((int64_t)b * (2 * g.H) + 2 * y + (q >> 1)) * (int64_t)(2 * g.W) + 2 * x + (q & 1);
    v[q] = __float_as_uint(dY[row * ld + c]);
  }
  o.x = v[0]; o.y = v[1]; o.z = v[2]; o.w = v[3];
  *reinterpret_cast<u32x4_a4*>(out + pix * g.Cout + 4 * c) = o;
}

// PixelShuffle(2), bf16, even channel counts: one thread builds 8 consecutive conv channels of one input pixel
// (channels 4c .. 4c+7 = output channels c, c+1 of the four sub-pixels) from four 4-B reads and writes them with one
// 16-B store (the element-wise kernel above moves 2 B per load and per store).
__global__ void __launch_bounds__(256) unshuffle2_bf16_kernel(const bf16* __restrict__ dY, int64_t ld, bf16* __restrict__ out,
                                                              ConvGeom g) {
  const int Co = g.Cout / 4;                       // channels of the shuffled tensor
  const int groups = g.Cout / 8;                   // 8-channel groups per input pixel
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= g.pixels() * groups) return;
  const int gi = (int)(i % groups);
  const int64_t pix = i / groups;
  int b, y, x;
  g.decode(pix, b, y, x);
  const int c0 = 2 * gi;                           // output channels c0, c0 + 1
  uint32_t v[4];                                   // sub-pixel q = 2 i + j: {c0, c0 + 1}
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int64_t row = ((int64_t)b * (2 * g.H) + 2 * y + (q >> 1)) * (int64_t)(2 * g.W) + 2 * x + (q & 1);
    v[q] = *reinterpret_cast<const uint32_t*>(dY + row * ld + c0);
  }
  (void)Co;
  // conv channel 4c + q: [c0: q0 q1 q2 q3][c0+1: q0 q1 q2 q3]
  u32x4_a4 o;
  o.x = (v[0] & 0xffffu) | (v[1] << 16);
  o.y = (v[2] & 0xffffu) | (v[3] << 16);
  o.z = (v[0] >> 16) | (v[1] & 0xffff0000u);
  o.w = (v[2] >> 16) | (v[3] & 0xffff0000u);
  *reinterpret_cast<u32x4_a4*>(out + pix * g.Cout + 8 * gi) = o;
}

// ------------------------------------------------------------------------------------------------
// wgrad: dW[co][ci][ky][kx] = s * sum_p dY[p][co] * in_act(X)[p + (ky,kx) - pad][ci]
// ------------------------------------------------------------------------------------------------
constexpr int CW_MAXT = 6;
constexpr int CW_STRIPE = 32;

template <typename T>
struct ConvWgradArgs {
  const T* X; int64_t ldx; int in_act;
  const T* dY; int64_t lddy;  // plain rows (B*H*W, Cout)
  float* slab;                // [nm][ks][Cout][ks][CinP]
  ConvGeom g;
  int64_t pix_per_wg;
  int ldn, ldk;               // LDS strides (bytes)
  int NT, KT, CinP;           // CinP = KT*32 (padded Cin incl. the ones column)
  int ones_col;               // column of the centre tap carrying 1.0 (d(bias)), or -1
};

template <typename T>
__global__ void __launch_bounds__(512) conv_wgrad_mfma_kernel(const ConvWgradArgs<T> p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using MM = Mma<T>;
  constexpr int HP = MM::HP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  const ConvGeom g = p.g;
  const int ks = g.ks;
  const int ky = blockIdx.y;  // kernel row handled by this workgroup
  char* dYs = smem;
  char* Xs = smem + (size_t)CW_STRIPE * p.ldn;  // [kx][32][ldk]
  f32x16 acc[CW_MAXT];
#pragma unroll
  for (int j = 0; j < CW_MAXT; ++j)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[j][v] = 0.f;
  const int ntiles = p.NT * ks * p.KT;
  const int64_t P = g.pixels();
  const int64_t p_begin = (int64_t)blockIdx.x * p.pix_per_wg;
  const int64_t p_end = (p_begin + p.pix_per_wg < P) ? p_begin + p.pix_per_wg : P;
  const int npk = p.NT * 32 / HP, kpk = p.KT * 32 / HP;

  for (int64_t p0 = p_begin; p0 < p_end; p0 += CW_STRIPE) {
    __syncthreads();
    for (int idx = tid; idx < CW_STRIPE * npk; idx += 512) {
      const int row = idx / npk, pk = idx - row * npk;
      const bool valid = p0 + row < p_end;
      const Pack16 v = load_pack<T>(p.dY + (valid ? (p0 + row) : 0) * p.lddy, pk * HP, g.Cout, valid);
      *reinterpret_cast<Pack16*>(dYs + (size_t)row * p.ldn + pk * 16) = v;
    }
    for (int idx = tid; idx < ks * CW_STRIPE * kpk; idx += 512) {
      const int pk = idx % kpk;
      const int rest = idx / kpk;
      const int row = rest % CW_STRIPE, kx = rest / CW_STRIPE;
      const int64_t pix = p0 + row;
      const bool pvalid = pix < p_end;
      int b, y, x;
      g.decode(pvalid ? pix : 0, b, y, x);
      const int yy = y + ky - g.pad, xx = x + kx - g.pad;
      const bool valid = pvalid && yy >= 0 && yy < g.H && xx >= 0 && xx < g.W;
      const int k0 = pk * HP;
      Pack16 v = load_pack<T>(p.X + (valid ? (((int64_t)b * g.H + yy) * g.W + xx) : 0) * p.ldx, k0, g.Cin, valid);
      float f[HP];
      MM::unpack(v, f);
      if (p.in_act) {
#pragma unroll
        for (int e = 0; e < HP; ++e) f[e] = apply_act(f[e], p.in_act);
      }
      if (p.ones_col >= 0 && ky == g.pad && kx == g.pad && pvalid) {
#pragma unroll
        for (int e = 0; e < HP; ++e)
          if (k0 + e == p.ones_col) f[e] = 1.0f;
      }
      *reinterpret_cast<Pack16*>(Xs + ((size_t)kx * CW_STRIPE + row) * p.ldk + pk * 16) = MM::pack(f);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < CW_MAXT; ++j) {
      const int ti = wave + 8 * j;
      if (ti < ntiles) {
        const int nt = ti / (ks * p.KT);
        const int rem = ti - nt * (ks * p.KT);
        const int kx = rem / p.KT, kt = rem - kx * p.KT;
        const char* Xk = Xs + (size_t)kx * CW_STRIPE * p.ldk;
        if constexpr (sizeof(T) == 2) {
          const int gq = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
          const int colA = nt * 32 + 16 * (gq & 1) + 4 * pp;
          const int colB = kt * 32 + 16 * (gq & 1) + 4 * pp;
#pragma unroll
          for (int ms = 0; ms < CW_STRIPE / 16; ++ms) {
            const int rowb = ms * 16 + 8 * h + q;
            typedef __attribute__((address_space(3))) s16x4_t* lds_p;
            const s16x4_t a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(dYs + (size_t)rowb * p.ldn + colA * 2));
            const s16x4_t a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(dYs + (size_t)(rowb + 4) * p.ldn + colA * 2));
            const s16x4_t b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(Xk + (size_t)rowb * p.ldk + colB * 2));
            const s16x4_t b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(Xk + (size_t)(rowb + 4) * p.ldk + colB * 2));
            const uint2 ua0 = __builtin_bit_cast(uint2, a0), ua1 = __builtin_bit_cast(uint2, a1);
            const uint2 ub0 = __builtin_bit_cast(uint2, b0), ub1 = __builtin_bit_cast(uint2, b1);
            Pack16 a, bq;
            a.w[0] = ua0.x; a.w[1] = ua0.y; a.w[2] = ua1.x; a.w[3] = ua1.y;
            bq.w[0] = ub0.x; bq.w[1] = ub0.y; bq.w[2] = ub1.x; bq.w[3] = ub1.y;
            MM::mma(acc[j], a, bq);
          }
        } else {
          const float* dYf = reinterpret_cast<const float*>(dYs);
          const float* Xf = reinterpret_cast<const float*>(Xk);
          const int lda = p.ldn / 4, ldb = p.ldk / 4;
#pragma unroll 8
          for (int s2 = 0; s2 < CW_STRIPE / 2; ++s2) {
            const float av = dYf[(2 * s2 + h) * lda + nt * 32 + r];
            const float bv = Xf[(2 * s2 + h) * ldb + kt * 32 + r];
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[j], 0, 0, 0);
          }
        }
      }
    }
  }
  // slab[m-block][ky][co][kx][CinP]
  float* my = p.slab + (((int64_t)blockIdx.x * ks + ky) * g.Cout) * ks * p.CinP;
#pragma unroll
  for (int j = 0; j < CW_MAXT; ++j) {
    const int ti = wave + 8 * j;
    if (ti < ntiles) {
      const int nt = ti / (ks * p.KT);
      const int rem = ti - nt * (ks * p.KT);
      const int kx = rem / p.KT, kt = rem - kx * p.KT;
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int co = nt * 32 + acc_row(v, h);
        if (co < g.Cout) my[((int64_t)co * ks + kx) * p.CinP + kt * 32 + r] = acc[j][v];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// wgrad for 3x3 / pad 1 convs whose image width is a multiple of 32: the pipelined form.
// A stripe is 32 consecutive pixels of ONE image row, so the inputs of all three kx taps are the 34
// consecutive pixels x0-1 .. x0+32 of row y+ky-1: they are staged once and the tap is a row offset of
// the transposed fragment reads.  Structure as the Linear wgrad (linear_mfma.hip): 8 waves x up to 6
// accumulator tiles, PF stripes prefetched in registers, double-buffered LDS tiles (one barrier per
// stripe), precomputed per-thread staging plan, software-pipelined fragment reads.  One workgroup per
// (pixel range, ky); slab layout and reduction are those of the generic kernel above.
// XF: 0 = no input activation, 3 = run-time activation.
// ------------------------------------------------------------------------------------------------

// SL = pixels per stripe (32, or 128 for shapes with at most 2x2 tiles per tap: fewer barriers per byte)
// SP (fp32 rows): RDST_F32X3, as in the Linear weight gradient (linear_mfma.hip): stash() keeps the bf16 hi and lo terms of a
// staged pack as two planes of the tile row, multiply() reads both operands transposed; a k-step is 8 pixels, two MFMAs.
template <typename T, int PF, int XF, int SL, bool SP = false>
__global__ void __launch_bounds__(512) conv_wgrad_rows_kernel(const ConvWgradArgs<T> p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using MM = Mma<T, SP>;
  constexpr bool SPL = MM::SPLIT;
  constexpr int HP = MM::HP;
  constexpr int ES = (int)sizeof(T);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, h = lane >> 5;
  const ConvGeom g = p.g;
  const int ky = blockIdx.y;
  constexpr int XR = SL + 2;   // rows of the X tile: pixels x0-1 .. x0+SL
  const int buf_bytes = SL * p.ldn + XR * p.ldk;
  f32x16 acc[CW_MAXT];
#pragma unroll
  for (int j = 0; j < CW_MAXT; ++j)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[j][v] = 0.f;
  const int ntiles = p.NT * 3 * p.KT;
  const int64_t P = g.pixels();
  const int64_t p_begin = (int64_t)blockIdx.x * p.pix_per_wg;
  const int64_t p_end = (p_begin + p.pix_per_wg < P) ? p_begin + p.pix_per_wg : P;
  const int npk = p.NT * 32 / HP, kpk = p.KT * 32 / HP;
  const bool small_n = g.Cout < HP;   // e.g. the 60 -> 1 tail conv: dY rows are shorter than a pack

  // per-thread staging plan (loop invariant)
  constexpr int DYMAX = 16 / HP, XMAX = 24 / HP;   // 32 x 256 | 34 x 256 elements over 512 threads
  int dy_row[DYMAX], dy_col[DYMAX], dy_sh[DYMAX], dy_lds[DYMAX];
  int x_row[XMAX], x_col[XMAX], x_sh[XMAX], x_lds[XMAX], x_k0[XMAX];
#pragma unroll
  for (int i = 0; i < DYMAX; ++i) {
    const int idx = tid + 512 * i;
    const int row = idx / npk, pk = idx - row * npk;
    int k0 = pk * HP;
    dy_row[i] = (!small_n && idx < SL * npk) ? row : -1;
    dy_lds[i] = row * p.ldn + pk * (SPL ? 8 : 16);
    if (k0 >= g.Cout) k0 = 0;
    dy_sh[i] = (k0 + HP > g.Cout) ? (k0 + HP - g.Cout) * ES : 0;
    dy_col[i] = k0 * ES - dy_sh[i];
  }
#pragma unroll
  for (int i = 0; i < XMAX; ++i) {
    const int idx = tid + 512 * i;
    const int row = idx / kpk, pk = idx - row * kpk;
    const int k0 = pk * HP;
    x_row[i] = idx < XR * kpk ? row : -1;
    x_lds[i] = SL * p.ldn + row * p.ldk + pk * (SPL ? 8 : 16);
    x_k0[i] = k0;
    const int kl = k0 < g.Cin ? k0 : 0;
    x_sh[i] = (kl + HP > g.Cin) ? (kl + HP - g.Cin) * ES : 0;
    x_col[i] = kl * ES - x_sh[i];
  }
  // small-Cout mode: thread t < 32*Cout moves ONE element of dY per stripe
  const int sm_row = tid / (g.Cout > 0 ? g.Cout : 1), sm_col = tid - sm_row * g.Cout;
  const bool sm_act = small_n && tid < SL * g.Cout;
  auto shift_pack = [&](Pack16& q, int sh_bytes) {
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
  // stripe geometry (wave uniform): rows of the X tile are pixels x0-1 .. x0+SL of image row y+ky-1.
  // Positions advance incrementally (no divisions inside the loop).
  struct Pos { int b, y, x0; int64_t p0; };
  auto advance = [&](Pos& q, int nstripes) {
    q.p0 += (int64_t)nstripes * SL;
    q.x0 += nstripes * SL;
    while (q.x0 >= g.W) { q.x0 -= g.W; if (++q.y == g.H) { q.y = 0; ++q.b; } }
  };
  struct Stripe { int left, lo, hi; bool row_ok; int64_t xpix; };
  auto locate = [&](const Pos& q) {
    Stripe s;
    const int yy = q.y + ky - 1;
    s.row_ok = yy >= 0 && yy < g.H;
    s.lo = q.x0 == 0 ? 1 : 0;                        // first / last tile row that is inside the image
    s.hi = q.x0 + SL == g.W ? SL : SL + 1;
    s.left = (int)(p_end - q.p0 < SL ? p_end - q.p0 : SL);
    s.xpix = s.row_ok ? ((int64_t)q.b * g.H + yy) * g.W + q.x0 - 1 : q.p0 - 1;   // any readable pixel when the row is outside
    return s;
  };
  Pack16 rdy[PF][DYMAX], rx[PF][XMAX];
  T rsm[PF];
  const int64_t ldy_b = p.lddy * ES, ldx_b = p.ldx * ES;
  auto prefetch = [&](int set, const Pos& pq) {
    const Stripe s = locate(pq);
    const int64_t p0 = pq.p0;
    const char* dyb = reinterpret_cast<const char*>(p.dY) + p0 * ldy_b;
    const char* xb = reinterpret_cast<const char*>(p.X) + s.xpix * ldx_b;
#pragma unroll
    for (int i = 0; i < DYMAX; ++i)
      if (dy_row[i] >= 0) {
        const int row = dy_row[i] < s.left ? dy_row[i] : s.left - 1;
        rdy[set][i] = ld16(dyb, (uint32_t)(row * (int)ldy_b + dy_col[i]));
      }
    if (small_n) rsm[set] = sm_act && sm_row < s.left ? p.dY[(p0 + sm_row) * p.lddy + sm_col] : from_f32<T>(0.f);
#pragma unroll
    for (int i = 0; i < XMAX; ++i)
      if (x_row[i] >= 0) {
        int row = x_row[i] < s.lo ? s.lo : x_row[i];
        row = row > s.hi ? s.hi : row;
        rx[set][i] = ld16(xb, (uint32_t)(row * (int)ldx_b + x_col[i]));
      }
  };
  auto stash = [&](int set, const Pos& pq, int b) {
    char* tile = smem + b * buf_bytes;
    const Stripe s = locate(pq);
#pragma unroll
    for (int i = 0; i < DYMAX; ++i)
      if (dy_row[i] >= 0) {
        Pack16 q = rdy[set][i];
        if (dy_sh[i]) shift_pack(q, dy_sh[i]);
        if (dy_row[i] >= s.left) { q.w[0] = 0u; q.w[1] = 0u; q.w[2] = 0u; q.w[3] = 0u; }   // pixels past the range add nothing
        if constexpr (SPL) {
          const Pack16 sp = MM::op(q);
          *reinterpret_cast<uint2*>(tile + dy_lds[i]) = make_uint2(sp.w[0], sp.w[1]);
          *reinterpret_cast<uint2*>(tile + dy_lds[i] + p.NT * 64) = make_uint2(sp.w[2], sp.w[3]);
        } else {
          *reinterpret_cast<Pack16*>(tile + dy_lds[i]) = q;
        }
      }
    if (sm_act) {
      if constexpr (SPL) {   // one element: its hi and lo terms into the two planes
        const float f = to_f32<T>(rsm[set]);
        const uint32_t hi = pack_bf16x2(f, 0.f);
        const uint32_t lo = pack_bf16x2(f - bf16lo(hi), 0.f);
        *reinterpret_cast<uint16_t*>(tile + sm_row * p.ldn + sm_col * 2) = (uint16_t)hi;
        *reinterpret_cast<uint16_t*>(tile + sm_row * p.ldn + p.NT * 64 + sm_col * 2) = (uint16_t)lo;
      } else {
        *reinterpret_cast<T*>(tile + sm_row * p.ldn + sm_col * ES) = rsm[set];
      }
    }
#pragma unroll
    for (int i = 0; i < XMAX; ++i)
      if (x_row[i] >= 0) {
        const bool valid = s.row_ok && x_row[i] >= s.lo && x_row[i] <= s.hi;
        const int k0 = x_k0[i];
        Pack16 q = rx[set][i];
        if (x_sh[i]) shift_pack(q, x_sh[i]);
        if (XF != 0 || !valid || (k0 <= g.Cin && k0 + HP > g.Cin)) {
          float f[HP];
          MM::unpack(q, f);
          if (XF != 0) {
#pragma unroll
            for (int e = 0; e < HP; ++e) f[e] = apply_act(f[e], p.in_act);
          }
#pragma unroll
          for (int e = 0; e < HP; ++e) {
            if (!valid) f[e] = 0.f;                 // zero padding of the conv
            if (k0 + e == g.Cin) f[e] = 1.0f;       // ones column: d(bias) on the centre tap
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
  // per-wave tile list and per-lane fragment offsets (loop invariant); tile = (nt, kx, kt)
  int tA[CW_MAXT], tB[CW_MAXT];
  {
    const int gq = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    constexpr bool TR = sizeof(T) == 2 || SPL;   // transposed 16-bit reads
    constexpr int RH = SPL ? 4 : 8;              // pixel rows per lane half and k-step
    const int laneA = TR ? (RH * h + q) * p.ldn + (16 * (gq & 1) + 4 * pp) * 2 : h * p.ldn + r * 4;
    const int laneB = TR ? (RH * h + q) * p.ldk + (16 * (gq & 1) + 4 * pp) * 2 : h * p.ldk + r * 4;
#pragma unroll
    for (int j = 0; j < CW_MAXT; ++j) {
      const int ti = wave + 8 * j;
      const int nt = ti / (3 * p.KT);
      const int rem = ti - nt * (3 * p.KT);
      const int kx = rem / p.KT, kt = rem - kx * p.KT;
      tA[j] = laneA + nt * 32 * (TR ? 2 : ES);
      tB[j] = SL * p.ldn + laneB + kx * p.ldk + kt * 32 * (TR ? 2 : ES);   // tap = row offset into the 34-row tile
    }
  }
  const int my_tiles = __builtin_amdgcn_readfirstlane(ntiles > wave ? (ntiles - wave + 7) / 8 : 0);
  auto multiply = [&](int b) {
#pragma unroll
    for (int sub = 0; sub < SL / 32; ++sub) {
      const char* tile = smem + b * buf_bytes;
      const int offA = sub * 32 * p.ldn, offB = sub * 32 * p.ldk;
      if constexpr (sizeof(T) == 2 || SPL) {
        typedef __attribute__((address_space(3))) s16x4_t* lds_p;
        constexpr int KR = SPL ? 8 : 16;   // pixel rows per k-step (split mode: the second read is the lo plane of the same rows)
        constexpr int NMS = 32 / KR;
        constexpr int NB = SPL ? 1 : 2;
        const int o2A = SPL ? p.NT * 64 : 4 * p.ldn, o2B = SPL ? p.KT * 64 : 4 * p.ldk;
        Pack16 fa[NB][NMS], fb[NB][NMS];
        auto frags = [&](int j, Pack16 (&A)[NMS], Pack16 (&B)[NMS]) {
#pragma unroll
          for (int ms = 0; ms < NMS; ++ms) {
            const char* ta = tile + offA + ms * KR * p.ldn;
            const char* tb = tile + offB + ms * KR * p.ldk;
            const s16x4_t a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(ta + tA[j]));
            const s16x4_t a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(ta + o2A + tA[j]));
            const s16x4_t b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(tb + tB[j]));
            const s16x4_t b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(tb + o2B + tB[j]));
            const uint2 ua0 = __builtin_bit_cast(uint2, a0), ua1 = __builtin_bit_cast(uint2, a1);
            const uint2 ub0 = __builtin_bit_cast(uint2, b0), ub1 = __builtin_bit_cast(uint2, b1);
            A[ms].w[0] = ua0.x; A[ms].w[1] = ua0.y; A[ms].w[2] = ua1.x; A[ms].w[3] = ua1.y;
            B[ms].w[0] = ub0.x; B[ms].w[1] = ub0.y; B[ms].w[2] = ub1.x; B[ms].w[3] = ub1.y;
          }
        };
        if (NB == 2 && my_tiles > 0) frags(0, fa[0], fb[0]);
#pragma unroll
        for (int j = 0; j < CW_MAXT; ++j) {
          if (j < my_tiles) {
            if (NB == 1) frags(j, fa[0], fb[0]);
            else if (j + 1 < CW_MAXT && j + 1 < my_tiles) frags(j + 1, fa[(j + 1) & (NB - 1)], fb[(j + 1) & (NB - 1)]);
#pragma unroll
            for (int ms = 0; ms < NMS; ++ms) MM::mma(acc[j], fa[j & (NB - 1)][ms], fb[j & (NB - 1)][ms]);
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < CW_MAXT; ++j) {
          if (j < my_tiles) {
#pragma unroll 8
            for (int s2 = 0; s2 < 16; ++s2) {
              const float av = *reinterpret_cast<const float*>(tile + offA + 2 * s2 * p.ldn + tA[j]);
              const float bv = *reinterpret_cast<const float*>(tile + offB + 2 * s2 * p.ldk + tB[j]);
              acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[j], 0, 0, 0);
            }
          }
        }
      }
    }
  };

  Pos cur, nxt;   // stripe being stashed / stripe being prefetched
  {
    int b0, y0, x00;
    g.decode(p_begin < P ? p_begin : 0, b0, y0, x00);
    cur.b = b0; cur.y = y0; cur.x0 = x00; cur.p0 = p_begin;
    nxt = cur;
  }
  // every prefetch defines every staging register (past the range it re-reads the current stripe): a
  // conditionally defined register would be live around the whole loop, get spilled, and the spill
  // makes the wave wait for its load
#pragma unroll
  for (int s = 0; s < PF; ++s) {
    prefetch(s, nxt.p0 < p_end ? nxt : cur);
    advance(nxt, 1);
  }
  int b = 0;
  while (cur.p0 < p_end) {
#pragma unroll
    for (int s = 0; s < PF; ++s) {
      if (cur.p0 < p_end) {
        stash(s, cur, b);   // tile b was last read two stripes ago; every wave has passed the barrier in between
        __syncthreads();
        prefetch(s, nxt.p0 < p_end ? nxt : cur);
        multiply(b);
        b ^= 1;
      }
      advance(cur, 1);
      advance(nxt, 1);
    }
  }
  // slab[m-block][ky][co][kx][CinP]
  float* my = p.slab + (((int64_t)blockIdx.x * 3 + ky) * g.Cout) * 3 * p.CinP;
#pragma unroll
  for (int j = 0; j < CW_MAXT; ++j) {
    const int ti = wave + 8 * j;
    if (ti < ntiles) {
      const int nt = ti / (3 * p.KT);
      const int rem = ti - nt * (3 * p.KT);
      const int kx = rem / p.KT, kt = rem - kx * p.KT;
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int co = nt * 32 + acc_row(v, h);
        if (co < g.Cout) my[((int64_t)co * 3 + kx) * p.CinP + kt * 32 + r] = acc[j][v];
      }
    }
  }
}

// dW[co][ci][ky][kx] = s * sum_m slab[m][ky][co][kx][ci];  dbias[co] = s * sum_m slab[m][pad][co][pad][ones_col]
__global__ void __launch_bounds__(256) conv_wgrad_reduce_kernel(const float* __restrict__ slab, int nm, ConvGeom g, int CinP,
                                                                int ones_col, float s, float* __restrict__ dW,
                                                                float* __restrict__ dbias) {
  const int ks = g.ks;
  const int64_t per_m = (int64_t)ks * g.Cout * ks * CinP;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= per_m) return;
  const int ci = (int)(i % CinP);
  int64_t rest = i / CinP;
  const int kx = (int)(rest % ks); rest /= ks;
  const int co = (int)(rest % g.Cout);
  const int ky = (int)(rest / g.Cout);
  const bool is_w = ci < g.Cin;
  const bool is_b = (ci == ones_col && ky == g.pad && kx == g.pad);
  if (!is_w && !is_b) return;
  float a = 0.f;
  for (int m0 = 0; m0 < nm; m0 += 16) {   // 16 loads in flight, summed in the same fixed order
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = (m0 + u < nm) ? slab[(int64_t)(m0 + u) * per_m + i] : 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) a += v[u];
  }
  if (is_w) {
    if (dW) dW[(((int64_t)co * g.Cin + ci) * ks + ky) * ks + kx] = a * s;
  } else if (dbias) {
    dbias[co] = a * s;
  }
}

}  // namespace

template <typename T>
int conv_fwd_mfma(const T* X, int64_t ldx, int in_act, const float* Wc, const float* bias, const T* R, int64_t ldr,
                  T* Y, int64_t ldy, const ConvGeom& g, float s, hipStream_t st) {
  if (mfma_disabled() || !rows_ok<T>(X, ldx) || g.Cin < 8) return RDST_ENOTSUP;
  // fp32: the weights of all taps stay in LDS and a wave holds the contracted channels of its pixels as <= 16 k-steps of 8:
  // more than 128 input channels (the 150 -> 60 fusion conv of an RDSTB) run as launches over equal slices of the input
  // channels, every launch after the first adding to the output in place (its residual operand)
  const int nsl = sizeof(T) == 4 ? (g.Cin + 127) / 128 : 1;
  if (nsl > 1 && (g.Cin % nsl != 0 || g.r != 1)) return RDST_ENOTSUP;
  const int cs = g.Cin / nsl;
  for (int i = 0; i < nsl; ++i) {
    ConvArgs<T> p{};
    p.A = X + i * cs; p.lda = ldx; p.CA = cs; p.Wc = Wc + (int64_t)i * cs * g.ks * g.ks;
    p.bias = i == 0 ? bias : nullptr; p.R = i == 0 ? R : Y; p.ldr = i == 0 ? ldr : ldy; p.Y = Y; p.ldy = ldy;
    p.in_act = in_act; p.g = g; p.Nout = g.Cout; p.s = s;
    const int rc = launch_conv<T, CMODE_FWD>(p, st, "conv_fwd_mfma");
    if (rc == RDST_ENOTSUP && i > 0) return rdst_fail(RDST_EINVAL, "conv_fwd_mfma: slice %d of %d declined after slice 0 ran", i, nsl);
    if (rc) return rc;
  }
  return 0;
}

size_t conv_mfma_scratch_bytes(const ConvGeom& g) {
  // un-shuffled dY (bf16 or fp32) + wgrad slab
  const size_t unsh = g.r > 1 ? (size_t)g.pixels() * g.Cout * 4 : 0;
  const int CinP = ((g.Cin + 1 + 31) / 32) * 32;
  const size_t slab = (size_t)128 * g.ks * g.Cout * g.ks * CinP * sizeof(float);
  return unsh + slab + 256;
}

template <typename T>
const T* plain_dy(const T* dY, int64_t lddy, const ConvGeom& g, void* scratch, int64_t& ld_out, hipStream_t st, int& rc) {
  rc = 0;
  if (g.r == 1) { ld_out = lddy; return dY; }
  T* tmp = reinterpret_cast<T*>(scratch);
  const int64_t tot = g.pixels() * g.Cout;
  if (sizeof(T) == 2 && g.r == 2 && g.Cout % 8 == 0 && (lddy % 2) == 0 && ((uintptr_t)dY & 3) == 0) {
    const int64_t n8 = g.pixels() * (g.Cout / 8);
    hipLaunchKernelGGL(unshuffle2_bf16_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, st, (const bf16*)dY, lddy,
                       (bf16*)tmp, g);
  } else if (sizeof(T) == 4 && g.r == 2 && g.Cout % 4 == 0) {
    const int64_t n4 = g.pixels() * (g.Cout / 4);
    hipLaunchKernelGGL(unshuffle2_f32_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, (const float*)dY, lddy,
                       (float*)tmp, g);
  } else {
    hipLaunchKernelGGL((unshuffle_kernel<T>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, dY, lddy, tmp, g);
  }
  rc = rdst_launch_status("unshuffle");
  ld_out = g.Cout;
  return tmp;
}

template <typename T>
int conv_dgrad_mfma(const T* X, int64_t ldx, int in_act, const float* Wc, const T* dYp, int64_t lddyp, T* dX,
                    int64_t lddx, const T* acc, int64_t ldacc, const ConvGeom& g, float s, hipStream_t st) {
  // dYp must already be plain rows (B*H*W, Cout)
  if (mfma_disabled() || !rows_ok<T>(dYp, lddyp) || g.Cout < 8) return RDST_ENOTSUP;
  // fp32, more than 128 output channels (the 60 -> 240 convs of the upsampler): launches over slices of the output
  // channels, the gradient is linear in dY; every launch after the first accumulates onto dX in place
  // (slices of <= 64 channels where the image width allows the row-stripe kernel: its nine-tap weight image of a 120-channel slice
  // does not fit the LDS and the launch fell back to the generic kernel with fragment-shaped global loads: 631 us per launch)
  const bool rows_form = g.ks == 3 && g.pad == 1 && g.W % 32 == 0;
  const int nsl = sizeof(T) != 4 ? 1 : (rows_form && g.Cout > 192 && g.Cout % 4 == 0) ? 4 : (g.Cout + 127) / 128;
  if (nsl > 1 && g.Cout % nsl != 0) return RDST_ENOTSUP;
  const int cs = g.Cout / nsl;
  for (int i = 0; i < nsl; ++i) {
    ConvArgs<T> p{};
    p.A = dYp + i * cs; p.lda = lddyp; p.CA = cs; p.Wc = Wc + (int64_t)i * cs * g.Cin * g.ks * g.ks;
    p.Y = dX; p.ldy = lddx; p.Xa = X; p.ldxa = ldx;
    p.in_act = in_act; p.Acc = i == 0 ? acc : dX; p.ldacc = i == 0 ? ldacc : lddx; p.g = g; p.Nout = g.Cin; p.s = s;
    const int rc = launch_conv<T, CMODE_DGRAD>(p, st, "conv_dgrad_mfma");
    if (rc == RDST_ENOTSUP && i > 0) return rdst_fail(RDST_EINVAL, "conv_dgrad_mfma: slice %d of %d declined after slice 0 ran", i, nsl);
    if (rc) return rc;
  }
  return 0;
}

template <typename T>
int conv_wgrad_mfma(const T* X, int64_t ldx, int in_act, const T* dYp, int64_t lddyp, float* dW, float* dbias,
                    float* slab, const ConvGeom& g, float s, hipStream_t st) {
  if (mfma_disabled() || !rows_ok<T>(X, ldx) || !rows_ok<T>(dYp, lddyp)) return RDST_ENOTSUP;
  ConvWgradArgs<T> p{};
  p.X = X; p.ldx = ldx; p.in_act = in_act; p.dY = dYp; p.lddy = lddyp; p.slab = slab; p.g = g;
  p.NT = (g.Cout + 31) / 32;
  p.KT = (g.Cin + 1 + 31) / 32;   // room for the ones column
  p.CinP = p.KT * 32;
  p.ones_col = g.Cin;
  if (p.NT * g.ks * p.KT > 8 * CW_MAXT) return RDST_ENOTSUP;
  // RDST_F32X3: the pipelined kernel's split form
  bool split = false;
  if constexpr (sizeof(T) == 4)
    split = rdst_split() && g.ks == 3 && g.pad == 1 && g.W % CW_STRIPE == 0;
  auto stride = [split](int elems) {   // (split mode: two bf16 planes in the bytes of the fp32 row, read like bf16 rows)
    const int b = elems * (int)sizeof(T);
    if (sizeof(T) == 4 && !split) return b;
    return b <= 64 ? 64 : ((b - 64 + 255) / 256) * 256 + 64;
  };
  p.ldn = stride(p.NT * 32);
  p.ldk = stride(p.KT * 32);
  const int64_t P = g.pixels();
  {  // the pipelined kernel: 3x3 / pad 1, rows of 32-pixel stripes, packs loaded whole (dword-aligned rows)
    constexpr int HP = Mma<T>::HP;
    const bool small_n = g.Cout < HP;
    bool long_stripes = p.NT <= 2 && p.KT <= 2 && g.W % 128 == 0;
    if (long_stripes && (size_t)2 * (128 * p.ldn + 130 * p.ldk) > 160 * 1024) long_stripes = false;   // (the split mode's padded rows)
    const int SLr = long_stripes ? 128 : CW_STRIPE;
    const size_t smem2 = (size_t)2 * (SLr * p.ldn + (SLr + 2) * p.ldk);
    const bool ok = g.ks == 3 && g.pad == 1 && g.W % CW_STRIPE == 0 && p.NT <= 8 && p.KT <= 8 && g.Cin >= HP &&
                    !((uintptr_t)X & 3) && (ldx * sizeof(T)) % 4 == 0 && (g.Cin * sizeof(T)) % 4 == 0 &&
                    (small_n ? CW_STRIPE * g.Cout <= 512
                             : (!((uintptr_t)dYp & 3) && (lddyp * sizeof(T)) % 4 == 0 && (g.Cout * sizeof(T)) % 4 == 0)) &&
                    smem2 <= 160 * 1024;
    if (ok) {
      int64_t nm = 85;   // x 3 kernel rows = 255 workgroups: one per CU
      if (nm > (P + SLr - 1) / SLr) nm = (P + SLr - 1) / SLr;
      p.pix_per_wg = (((P + nm - 1) / nm + SLr - 1) / SLr) * SLr;
      nm = (P + p.pix_per_wg - 1) / p.pix_per_wg;
      constexpr int PF = sizeof(T) == 2 ? 2 : 1;
#define RDST_CR_LAUNCH(XF)                                                                                            \
      {                                                                                                              \
        auto kern = long_stripes ? conv_wgrad_rows_kernel<T, PF, XF, 128> : conv_wgrad_rows_kernel<T, PF, XF, 32>;  \
        if constexpr (sizeof(T) == 4)                                                                                \
          if (split) kern = long_stripes ? conv_wgrad_rows_kernel<T, PF, XF, 128, true> : conv_wgrad_rows_kernel<T, PF, XF, 32, true>; \
        if (smem2 > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem2); \
        hipLaunchKernelGGL(kern, dim3((unsigned)nm, 3u), dim3(512), smem2, st, p);                                   \
      }
      if (in_act) RDST_CR_LAUNCH(3) else RDST_CR_LAUNCH(0)
#undef RDST_CR_LAUNCH
      if (int rc = rdst_launch_status("conv_wgrad_rows")) return rc;
      const int64_t per_m = (int64_t)g.ks * g.Cout * g.ks * p.CinP;
      hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((unsigned)((per_m + 255) / 256)), dim3(256), 0, st, slab, (int)nm, g,
                         p.CinP, p.ones_col, s, dW, dbias);
      return rdst_launch_status("conv_wgrad_reduce");
    }
  }
  const size_t smem = (size_t)CW_STRIPE * (p.ldn + (size_t)g.ks * p.ldk);
  if (smem > 160 * 1024) return RDST_ENOTSUP;
  int64_t nm = (P + CW_STRIPE - 1) / CW_STRIPE;
  const int64_t cap = 128;
  if (nm > cap) nm = cap;
  p.pix_per_wg = (((P + nm - 1) / nm + CW_STRIPE - 1) / CW_STRIPE) * CW_STRIPE;
  nm = (P + p.pix_per_wg - 1) / p.pix_per_wg;
  auto kern = conv_wgrad_mfma_kernel<T>;
  if (smem > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL(kern, dim3((unsigned)nm, (unsigned)g.ks), dim3(512), smem, st, p);
  if (int rc = rdst_launch_status("conv_wgrad_mfma")) return rc;
  const int64_t per_m = (int64_t)g.ks * g.Cout * g.ks * p.CinP;
  hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((unsigned)((per_m + 255) / 256)), dim3(256), 0, st, slab, (int)nm, g,
                     p.CinP, p.ones_col, s, dW, dbias);
  return rdst_launch_status("conv_wgrad_reduce");
}

#define INST(T)                                                                                                       \
  template int conv_fwd_mfma<T>(const T*, int64_t, int, const float*, const float*, const T*, int64_t, T*, int64_t,  \
                                const ConvGeom&, float, hipStream_t);                                                \
  template int conv_dgrad_mfma<T>(const T*, int64_t, int, const float*, const T*, int64_t, T*, int64_t, const T*,    \
                                  int64_t, const ConvGeom&, float, hipStream_t);                                     \
  template int conv_wgrad_mfma<T>(const T*, int64_t, int, const T*, int64_t, float*, float*, float*, const ConvGeom&, \
                                  float, hipStream_t);                                                               \
  template const T* plain_dy<T>(const T*, int64_t, const ConvGeom&, void*, int64_t&, hipStream_t, int&);
INST(float)
INST(bf16)
