// Batched fixed-order reductions of per-workgroup partial slabs (weight gradients, d(table)).
//
// Every backward op that splits its contraction over workgroups leaves `nwg` partial slabs and needs (1) their sum in
// a fixed order (deterministic gradients) and, behind a LayerNorm, (2) a finish that needs whole columns of that sum.
// Launched per op these are 5-13 us kernels with microseconds of work: 312 launches, 2.3 ms of a 25 ms step.  Between
// rdst_reduce_batch_begin() and rdst_reduce_batch_end() the ops RECORD their reductions instead of launching them and
// _end() runs all of them as two launches (every sum, then every LayerNorm finish): a Swin block's backward needs 2
// instead of 6.  The slabs and the G scratch must stay alive until _end() returns (they are the ops' workspaces).
#pragma once
#include "common.h"

namespace rbatch {
enum SumMap {
  MAP_COPY = 0,      // out[i] = sum                                  (G for a LayerNorm finish)
  MAP_LINEAR = 1,    // i = n * Kx + k: k < K -> dW[n*K + k] = s * sum, k == K -> dbias[n] = s * sum
  MAP_MLP = 2,       // i < hid*(C+1) -> G[i]; then [hid+1][C]: dW2[c*hid + j] / db2[c]
  MAP_DTABLE = 3,    // i = h * T + t -> dtable[t * heads + h]
  MAP_T = 4,         // element (j, c) of a [a+1][b] matrix: j < a -> out[c * a + j] (a transposed weight), j == a -> out2[c]
};
// Slab element formats: plain fp32 [tot], or G4 = bf16 in groups of 4 ROWS: element (n, c) of a [rows][W] matrix lives at
// ((n >> 2) * W + c) * 4 + (n & 3) (in bf16 units), i.e. a lane that owns column c of an MFMA accumulator tile dumps the
// 4 consecutive rows of a register group as ONE 8-byte store.  The partial sums of one workgroup (512 tokens of 131072)
// are rounded to bf16 there; the sum over the workgroups runs in fp32 (error ~2^-9 / sqrt(#workgroups) of a partial:
// far below the bf16 activations' own noise) and halves the slab traffic, the largest avoidable HBM stream of a step.
struct SumJob {
  const float* slab; int nwg; int64_t stride; int tot; int map;
  int g4;                                  // 0: fp32 slab, stride / tot in floats;  1: G4 bf16 slab: rows = a2, W = b2,
  int a2, b2;                              //    stride in 8-byte groups, tot = ceil(rows / 4) * W groups
  float* out; float* out2; float* out3;   // MAP_LINEAR: dW, dbias;  MAP_MLP: G, dW2, db2;  MAP_DTABLE: dtable
  int a, b;                               // MAP_LINEAR: K, Kx;  MAP_MLP: C, hid;  MAP_DTABLE: heads, T
  float s;
};
struct FinJob {                            // dW = s (gamma G + beta db), dbias = s db, dgamma = s sum_n W G, dbeta = s sum_n W db
  const float* G; const float* Wt; const float* gamma; const float* beta;
  int N, K, Kx; float s;
  float* dW; float* dbias; float* dgamma; float* dbeta;
};
// record-or-launch: inside a batch the job is queued (returns 0), otherwise it is launched on `st` right away
int sum(const SumJob& j, hipStream_t st);
int finish(const FinJob& j, hipStream_t st);
}  // namespace rbatch
