// Do matrix-core (MFMA) and vector-ALU instructions overlap on one SIMD of gfx950?  Two experiments, fp32 (32x32x2) and bf16
// (32x32x16) MFMAs:
//  (1) one workgroup of 8 waves per CU (waves w and w + 4 share a SIMD): waves 0-3 run a chain-free stream of MFMAs, waves 4-7 a
//      stream of independent v_fma_f32; timed: MFMA waves alone, VALU waves alone, both together;
//  (2) ONE wave per SIMD: per step one MFMA (four independent accumulator chains) followed by NV independent v_fma_f32.
// Measured on MI355X (profiles/r04_ubench_mfma_valu.txt): (1) "both" = the SUM of the two (fp32: 962 + 330 -> 1253 us; bf16:
// 525 + 332 -> 835 us; the VALU wave's own clock = MFMA stream + its alone time: it is starved while the MFMA stream runs);
// (2) vector instructions are never free beside MFMAs: the 64 / 33-cycle MFMA step becomes 92 / 52 with 4 v_fma behind every
// MFMA, 101 / 59 with 8, 121 / 79 with 16, 161 with 32 (fp32) — ~5 cycles each for the first few, ~2.5 after that — with
// four independent accumulator chains and with ONE dependent chain alike (its own gap is 0 / 2 cycles).
// An MFMA holds the SIMD's vector issue port for all of its passes; vector instructions of ANY wave of that SIMD cost their
// issue time on top.  Per SIMD: time >= sum(MFMA passes x 4 cycles) + sum(vector issue cycles): a kernel reaches the matrix peak
// only without vector work, and there is no "hiding the softmax under the MFMAs" on this chip.
// build: hipcc --offload-arch=gfx950 -O3 -w tools/ubench/mfma_valu_overlap.hip -o tools/ubench/mfma_valu_overlap.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>   // 0: v_mfma_f32_32x32x2_f32, 1: v_mfma_f32_32x32x16_bf16
__global__ void __launch_bounds__(512) k(int iters, int do_mfma, int do_valu, float* out, long long* cyc) {
  const int wave = threadIdx.x >> 6;
  const bool mf = wave < 4;
  float r = 0.f;
  const long long t0 = clock64();
  if (mf) {
    if (do_mfma) {
      f32x16 a[4];
      for (int q = 0; q < 4; ++q)
        for (int v = 0; v < 16; ++v) a[q][v] = 0.f;
      const float x = (float)threadIdx.x * 1e-3f, y = 1.0f + x;
      bf16x8 xb, yb;
      for (int e = 0; e < 8; ++e) { xb[e] = (__bf16)x; yb[e] = (__bf16)y; }
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int q = 0; q < 4; ++q) {   // four independent accumulators: no dependent back-to-back MFMAs
            if (KIND == 0) a[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a[q], 0, 0, 0);
            else a[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb, yb, a[q], 0, 0, 0);
          }
      }
      for (int q = 0; q < 4; ++q) r += a[q][0];
    }
  } else if (do_valu) {
    float f[16];
    for (int v = 0; v < 16; ++v) f[v] = (float)(threadIdx.x + v);
    const float m = 1.0000001f, c = 1e-7f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int v = 0; v < 16; ++v) f[v] = __builtin_fmaf(f[v], m, c);   // 128 independent-enough v_fma_f32 per iteration
    }
    for (int v = 0; v < 16; ++v) r += f[v];
  }
  const long long t1 = clock64();
  out[blockIdx.x * 512 + threadIdx.x] = r;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

// the SAME wave: per step one MFMA (of four independent chains) followed by NV independent v_fma_f32; one wave per SIMD
template <int KIND, int NV, int NCH>
__global__ void __launch_bounds__(256) ks(int iters, float* out, long long* cyc) {
  f32x16 a[4];
  for (int q = 0; q < 4; ++q)
    for (int v = 0; v < 16; ++v) a[q][v] = 0.f;
  const float x = (float)threadIdx.x * 1e-3f, y = 1.0f + x;
  bf16x8 xb, yb;
  for (int e = 0; e < 8; ++e) { xb[e] = (__bf16)x; yb[e] = (__bf16)y; }
  float f[16];
  for (int v = 0; v < 16; ++v) f[v] = (float)(threadIdx.x + v);
  const float m = 1.0000001f, c = 1e-7f;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const int q = q4 % NCH;   // NCH = 1: ONE dependent chain (every MFMA waits for the previous result)
      if (KIND == 0) a[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a[q], 0, 0, 0);
      else a[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb, yb, a[q], 0, 0, 0);
#pragma unroll
      for (int v = 0; v < NV; ++v) f[v & 15] = __builtin_fmaf(f[v & 15], m, c);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const long long t1 = clock64();
  float r = 0.f;
  for (int q = 0; q < 4; ++q) r += a[q][0];
  for (int v = 0; v < 16; ++v) r += f[v];
  out[blockIdx.x * 256 + threadIdx.x] = r;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND, int NV, int NCH = 4>
void run_same(const char* name) {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
  const int iters = 4000;
  ks<KIND, NV, NCH><<<256, 256>>>(iters, out, cyc);
  ks<KIND, NV, NCH><<<256, 256>>>(iters, out, cyc);
  hipDeviceSynchronize();
  long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-28s same wave, %d chain(s), %2d v_fma per MFMA: %7.1f cycles per MFMA step\n", name, NCH, NV, (double)h / (4.0 * iters));
}

template <int KIND>
void run(const char* name) {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
  const int iters = 2000;
  for (int mode = 1; mode <= 3; ++mode) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND><<<256, 512>>>(iters, mode & 1, (mode >> 1) & 1, out, cyc);   // warm-up
    hipEventRecord(e0);
    k<KIND><<<256, 512>>>(iters, mode & 1, (mode >> 1) & 1, out, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    printf("%-28s %-10s %8.1f us   wave clocks: mfma wave %lld, valu wave %lld\n", name,
           mode == 1 ? "MFMA only" : mode == 2 ? "VALU only" : "both", ms * 1e3, h[0], h[4]);
  }
}

int main() {
  run<0>("v_mfma_f32_32x32x2_f32");
  run<1>("v_mfma_f32_32x32x16_bf16");
  run_same<0, 0>("v_mfma_f32_32x32x2_f32"); run_same<0, 8>("v_mfma_f32_32x32x2_f32"); run_same<0, 16>("v_mfma_f32_32x32x2_f32");
  run_same<0, 32>("v_mfma_f32_32x32x2_f32");
  run_same<1, 0>("v_mfma_f32_32x32x16_bf16"); run_same<1, 4>("v_mfma_f32_32x32x16_bf16"); run_same<1, 8>("v_mfma_f32_32x32x16_bf16");
  run_same<1, 16>("v_mfma_f32_32x32x16_bf16");
  run_same<0, 0, 1>("v_mfma_f32_32x32x2_f32"); run_same<0, 4, 1>("v_mfma_f32_32x32x2_f32"); run_same<0, 8, 1>("v_mfma_f32_32x32x2_f32");
  run_same<1, 0, 1>("v_mfma_f32_32x32x16_bf16"); run_same<1, 4, 1>("v_mfma_f32_32x32x16_bf16"); run_same<1, 8, 1>("v_mfma_f32_32x32x16_bf16");
  run_same<1, 16, 1>("v_mfma_f32_32x32x16_bf16");
  return 0;
}
