// rdst_pack_batch: the bf16 fragment images of MANY layers' weights in a handful of launches (48 layers each) instead
// of one ~5 us pack kernel in front of every forward op (120 Linear + 11 conv ops in an RDST-E1 step).  The host side
// (rdst_amd/ops.py: PackPlan) runs it once at the start of a network forward and hands every op its slice of the arena
// with workspace_bytes = RDST_PREPACKED.
#include "pack.h"

namespace {
constexpr int PB_MAX = 48;   // jobs per launch (the table travels by value in the kernel arguments: 48 x 64 + 200 bytes < 4 KB)
struct PackBatch { rdst_pack_job j[PB_MAX]; int first[PB_MAX + 1]; int n; };
static_assert(sizeof(PackBatch) <= 4000, "kernel arguments");

__global__ void __launch_bounds__(256) pack_batch_kernel(const PackBatch bt) {
  int k = 0;
#pragma unroll
  for (int q = 1; q < PB_MAX; ++q) k += (q < bt.n && (int)blockIdx.x >= bt.first[q]) ? 1 : 0;
  const rdst_pack_job& J = bt.j[k];
  const int bid = (int)blockIdx.x - bt.first[k];
  if (J.kind == RDST_PACK_LINEAR) {
    const int nt = (J.N + 31) / 32, ks = (J.K + 15) / 16;
    bf16* wp = reinterpret_cast<bf16*>(J.out);
    float* sb = reinterpret_cast<float*>(reinterpret_cast<char*>(J.out) + (size_t)nt * ks * 1024);
    lin3_pack_block(bid, J.W, J.gamma, J.beta, J.bias, wp, sb, J.N, J.K, ks, nt, J.s);
  } else if (J.kind == RDST_PACK_LINEAR_X3) {
    const int nt = (J.N + 31) / 32, ks = (J.K + 15) / 16;
    uint32_t* wp = reinterpret_cast<uint32_t*>(J.out);
    float* bp = reinterpret_cast<float*>(reinterpret_cast<char*>(J.out) + (size_t)nt * ks * 2048);
    lin3x_pack_block(bid, J.W, J.gamma, J.beta, J.bias, wp, bp, J.N, J.K, ks, nt, J.s);
  } else if (J.kind == RDST_PACK_CONV3_FWD_X3) {
    conv3x_pack_block(bid, J.W, reinterpret_cast<uint32_t*>(J.out), J.K, J.N, J.K, J.N, (J.K + 15) / 16, (J.N + 31) / 32, PK_FWD, J.s);
  } else if (J.kind == RDST_PACK_LINEAR_SEC3) {
    const int nt = lin3sec_tiles(J.K, J.N, 3), ks = (J.K + 15) / 16;
    bf16* wp = reinterpret_cast<bf16*>(J.out);
    float* sb = reinterpret_cast<float*>(reinterpret_cast<char*>(J.out) + (size_t)nt * ks * 1024);
    lin3sec_pack_block(bid, J.W, J.gamma, J.beta, J.bias, wp, sb, J.N, J.K, 3, J.s);
  } else {
    // conv forward image: K = Cin, N = Cout
    const int ks = (J.K + 15) / 16, ct = (J.N + 31) / 32;
    conv3_pack_block(bid, J.W, reinterpret_cast<bf16*>(J.out), J.K, J.N, J.K, J.N, ks, ct, PK_FWD, J.s);
  }
}
}  // namespace

extern "C" int rdst_pack_batch(const rdst_pack_job* jobs, int njobs, void* stream) {
  if (njobs < 0 || (njobs > 0 && !jobs)) return rdst_fail(RDST_EINVAL, "rdst_pack_batch: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  for (int base = 0; base < njobs; base += PB_MAX) {
    PackBatch bt{};
    bt.n = njobs - base < PB_MAX ? njobs - base : PB_MAX;
    int blocks = 0;
    for (int q = 0; q < bt.n; ++q) {
      const rdst_pack_job& J = jobs[base + q];
      if (!J.W || !J.out || J.N <= 0 || J.K <= 0 || ((uintptr_t)J.out & 15) ||
          (J.kind != RDST_PACK_LINEAR && J.kind != RDST_PACK_CONV3_FWD && J.kind != RDST_PACK_LINEAR_SEC3 && J.kind != RDST_PACK_LINEAR_X3 &&
           J.kind != RDST_PACK_CONV3_FWD_X3) ||
          (J.kind == RDST_PACK_LINEAR_SEC3 && J.N % 3))
        return rdst_fail(RDST_EINVAL, "rdst_pack_batch: job %d is malformed", base + q);
      bt.j[q] = J;
      bt.first[q] = blocks;
      blocks += J.kind == RDST_PACK_LINEAR ? lin3_pack_blocks(J.K, J.N)
                : J.kind == RDST_PACK_LINEAR_X3 ? lin3x_pack_blocks(J.K, J.N)
                : J.kind == RDST_PACK_CONV3_FWD_X3 ? conv3x_pack_blocks(J.K, J.N)
                : J.kind == RDST_PACK_LINEAR_SEC3 ? lin3sec_pack_blocks(J.K, J.N, 3) : conv3_pack_blocks(J.K, J.N);
    }
    bt.first[bt.n] = blocks;
    hipLaunchKernelGGL(pack_batch_kernel, dim3((unsigned)blocks), dim3(256), 0, st, bt);
    if (int rc = rdst_launch_status("pack_batch")) return rc;
  }
  return 0;
}
