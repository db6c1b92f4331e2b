// How deep can ONE wave queue LDS-DMA (buffer_load_dwordx4 ... lds), and what does that make of a CU's streaming rate?
// The phase stamps of lin3x / lnlin3x (profiles/r06_lbx_stamps.txt, r06_lin3x_stamps.txt) show the waves that ISSUE a tile's DMA blocked
// for about as long as the tile takes to stream: 100-190 cycles per 1 KB instruction with 8-12 waves issuing, 2.2 k cycles per
// instruction when two waves issued a whole tile alone.  This file measures it directly, one workgroup per CU on all 256 CUs (so HBM
// sees the whole chip), W waves per workgroup, every wave issues N pieces of 1 KB back to back (cold addresses, 64 B/lane-group
// contiguous), then waits:
//   issue  = cycles until the last instruction has left the wave, per instruction
//   total  = cycles until vmcnt(0), per instruction            (1 KB / total = bytes per cycle and wave)
// beside the same bytes as global_load_dwordx4 into registers (16 in flight per wave) + ds_write_b128.
// build: hipcc --offload-arch=gfx950 -O3 -w tools/ubench/lds_dma_depth.hip -o tools/ubench/lds_dma_depth.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef uint32_t u32x4s_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int N>
__global__ void __launch_bounds__(1024) dma_kernel(const char* src, long long bytes, long long* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), nw = blockDim.x >> 6;
  u32x4s_t rs;
  rs.x = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)src);
  rs.y = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)src >> 32) & 0xffffu);
  rs.z = __builtin_amdgcn_readfirstlane((uint32_t)bytes);
  rs.w = 0x00020000u;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  // piece p of this wave: 1 KB at a cold place of the buffer (workgroup, wave and piece spread far apart)
  const uint32_t base = (uint32_t)((((long long)blockIdx.x * nw + wave) * N) * 1024 % (bytes - 1024 * 64)) + lane * 16;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)((wave * 4 + (i & 3)) * 1024));
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(base + i * 1024), "s"(dst), "s"(rs) : "memory");
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long t2 = __builtin_amdgcn_s_memtime();
  if (lane == 0) { out[(blockIdx.x * nw + wave) * 2] = t1 - t0; out[(blockIdx.x * nw + wave) * 2 + 1] = t2 - t0; }
}

template <int N>
__global__ void __launch_bounds__(1024) reg_kernel(const char* src, long long bytes, long long* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), nw = blockDim.x >> 6;
  const char* p = src + ((((long long)blockIdx.x * nw + wave) * N) * 1024 % (bytes - 1024 * 64)) + lane * 16;
  u32x4 v[N];
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
  for (int i = 0; i < N; ++i) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[i]) : "v"(p + i * 1024) : "memory");
  const long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = 0; i < N; ++i) {
    asm volatile("" : "+v"(v[i]));
    *reinterpret_cast<u32x4*>(smem + (wave * 4 + (i & 3)) * 1024 + lane * 16) = v[i];
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const long long t2 = __builtin_amdgcn_s_memtime();
  if (lane == 0) { out[(blockIdx.x * nw + wave) * 2] = t1 - t0; out[(blockIdx.x * nw + wave) * 2 + 1] = t2 - t0; }
}

template <int N>
void run(int W, const char* src, long long bytes, long long* dout) {
  const int grid = 256;
  long long* h = (long long*)malloc(sizeof(long long) * grid * W * 2);
  for (int kind = 0; kind < 2; ++kind) {
    for (int rep = 0; rep < 2; ++rep) {
      if (kind == 0) hipLaunchKernelGGL(dma_kernel<N>, dim3(grid), dim3(64 * W), 64 * 1024, 0, src, bytes, dout);
      else hipLaunchKernelGGL(reg_kernel<N>, dim3(grid), dim3(64 * W), 64 * 1024, 0, src, bytes, dout);
      hipDeviceSynchronize();
    }
    hipMemcpy(h, dout, sizeof(long long) * grid * W * 2, hipMemcpyDeviceToHost);
    double a = 0, b = 0;
    for (int i = 0; i < grid * W; ++i) { a += h[2 * i]; b += h[2 * i + 1]; }
    a /= grid * W; b /= grid * W;
    // s_memtime ticks at 100 MHz on this part: report ticks and ns
    printf("%-9s W=%2d N=%2d: issue %7.1f ticks/instr  total %7.1f ticks/instr  -> %6.2f KB per 1000 ticks and wave, %7.2f per CU\n",
           kind == 0 ? "lds-dma" : "registers", W, N, a / N, b / N, 1000.0 * N / b, 1000.0 * N * W / b);
  }
  free(h);
}

int main() {
  const long long bytes = 1ll << 30;
  char* src; long long* dout;
  hipMalloc(&src, bytes); hipMemset(src, 1, bytes);
  hipMalloc(&dout, sizeof(long long) * 256 * 16 * 2);
  hipFuncSetAttribute((const void*)dma_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  for (int W : {1, 2, 4, 8, 12, 16}) {
    run<4>(W, src, bytes, dout);
    run<16>(W, src, bytes, dout);
  }
  return 0;
}
