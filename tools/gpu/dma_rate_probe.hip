// LDS-DMA issue / completion rate probe (gfx950): how fast does ONE wave's stream of global_load_lds_dwordx4 retire, and
// what does the access pattern cost?  Each wave issues N instructions (1 KiB each) back to back, then waits vmcnt(0).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__device__ __forceinline__ void dma16(const float* sbase, unsigned voff, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}
// pattern 0: lane-contiguous 1 KiB; 1: 13 row pieces of 80 bytes at a 268800-byte stride (channel planes), unaligned start (the
// forward kernel's shape); 2: the same pieces starting on 128-byte boundaries (one line each); 3: pattern 1 re-reading the SAME
// addresses every instruction (L1 hits after the first); 4: 64 separate 16-byte pieces in 64 different lines
// patterns 5 / 6 (round 5): the ring BACKWARD's gradient staging.  A hit's operand is grad_out[roi, c0 : c0 + 32, 7, 7]: ONE
// contiguous, 16-byte aligned block of 6272 bytes.  5 = how the ring kernel reads it today: 448 pieces (channel, bin row,
// piece q) at byte (c * 49 + r * 7 + (q ? 3 : 0)) * 4 — 4-byte aligned, overlapping pairs, so that a bin row sits in two
// aligned 16-byte LDS slots; 6 = the same block as a LINEAR copy of 392 aligned pieces.  ROI blocks are drawn from a
// 51 MB tensor (1024 ROIs x 256 channels), instruction i of a wave = 64 consecutive pieces of block (i / 7).
template <int PATTERN>
__global__ void __launch_bounds__(64) probe_bwd(const float* src, long long* ticks, int n, int span_kb, float* sink) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x;
  typedef __attribute__((address_space(3))) float* lds_fptr_t;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(reinterpret_cast<uintptr_t>((lds_fptr_t)lds)));
  const long long t0 = wall_clock64();
  for (int i = 0; i < n; ++i) {
    const unsigned roi = (blockIdx.x * 2654435761u + (i / 7) * 40503u) % 1024u;
    const float* base = src + (static_cast<size_t>(roi) * 8 + (blockIdx.x & 7)) * 1568;      // 6272-byte block
    const int p = (i % 7) * 64 + lane;
    unsigned voff;
    if (PATTERN == 5) { const int c = p / 14, rem = p % 14, r = rem / 2, q = rem % 2; voff = (c * 49 + r * 7 + (q ? 3 : 0)) * 4; }
    else voff = min(p, 391) * 16;
    dma16(base, voff, lds0 + (i % span_kb) * 1024);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long t1 = wall_clock64();
  if (lane == 0) ticks[blockIdx.x] = t1 - t0;
  if (sink) sink[blockIdx.x * 64 + lane] = lds[lane];
}

template <int PATTERN>
__global__ void __launch_bounds__(64) probe(const float* src, long long* ticks, int n, int span_kb, float* sink) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x;
  typedef __attribute__((address_space(3))) float* lds_fptr_t;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(reinterpret_cast<uintptr_t>((lds_fptr_t)lds)));
  unsigned voff;
  if (PATTERN == 0) voff = lane * 16;
  else if (PATTERN == 4) voff = lane * 268800u + 12u;
  // 7 / 8 (round 5) separate the two things pattern 1 -> 2 changed at once: 7 = every lane's 16 bytes ALIGNED but the 80-byte
  // piece straddles a line boundary (starts 64 bytes into a line); 8 = lanes misaligned by 12 bytes as in 1, but the piece
  // stays inside one line (12 + 80 <= 128 with rows on 128-byte boundaries)
  else { const int ch = lane / 5, v = lane % 5; voff = ch * 268800u + v * 16u + (PATTERN == 2 ? 0u : (PATTERN == 7 ? 64u : 12u)); }
  const size_t wg_stride = (PATTERN == 0 ? 1024 : (PATTERN == 4 ? 64 : 13) * 268800ull) / 4;
  const float* base = src + (static_cast<size_t>(blockIdx.x) * 7919 % (PATTERN == 4 ? 13 : 97)) * wg_stride;   // all reads stay inside the 1 GiB buffer
  const long long t0 = wall_clock64();
  for (int i = 0; i < n; ++i) {
    const float* p = base + static_cast<size_t>(PATTERN == 3 ? 0 : i) * (PATTERN == 0 ? 256 * 256 : ((PATTERN == 2 || PATTERN == 7 || PATTERN == 8) ? 352 : 336));   // a new row / block every instruction (pattern 2: 1408-byte rows keep the 128-byte alignment)
    dma16(p, voff, lds0 + (i % span_kb) * 1024);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long t1 = wall_clock64();
  if (lane == 0) ticks[blockIdx.x] = t1 - t0;
  if (sink) sink[blockIdx.x * 64 + lane] = lds[lane];
}
int main() {
  const size_t bytes = 1ull << 30;
  float* src; hipMalloc(&src, bytes); hipMemset(src, 0, bytes);
  long long* ticks; hipMalloc(&ticks, 8 * 8192);
  std::vector<long long> h(8192);
  const int first = getenv("PROBE_FIRST") ? atoi(getenv("PROBE_FIRST")) : 0;
  for (int pattern = first; pattern < 9; ++pattern)
    for (int waves_per_cu : {1, 4, 8, 16})
      for (int n : {32, 128}) {
        const int grid = 256 * waves_per_cu;
        for (int rep = 0; rep < 2; ++rep) {
          if (pattern == 0) hipLaunchKernelGGL(probe<0>, dim3(grid), dim3(64), 16 * 1024, 0, src, ticks, n, 16, nullptr);
          else if (pattern == 1) hipLaunchKernelGGL(probe<1>, dim3(grid), dim3(64), 16 * 1024, 0, src, ticks, n, 16, nullptr);
          else if (pattern == 2) hipLaunchKernelGGL(probe<2>, dim3(grid), dim3(64), 16 * 1024, 0, src, ticks, n, 16, nullptr);
          else if (pattern == 3) hipLaunchKernelGGL(probe<3>, dim3(grid), dim3(64), 16 * 1024, 0, src, ticks, n, 16, nullptr);
          else if (pattern == 7) hipLaunchKernelGGL(probe<7>, dim3(grid), dim3(64), 16 * 1024, 0, src, ticks, n, 16, nullptr);
          else if (pattern == 8) hipLaunchKernelGGL(probe<8>, dim3(grid), dim3(64), 16 * 1024, 0, src, ticks, n, 16, nullptr);
          else if (pattern == 5) hipLaunchKernelGGL(probe_bwd<5>, dim3(grid), dim3(64), 16 * 1024, 0, src, ticks, n, 16, nullptr);
          else if (pattern == 6) hipLaunchKernelGGL(probe_bwd<6>, dim3(grid), dim3(64), 16 * 1024, 0, src, ticks, n, 16, nullptr);
          else hipLaunchKernelGGL(probe<4>, dim3(grid), dim3(64), 16 * 1024, 0, src, ticks, n, 16, nullptr);
          hipDeviceSynchronize();
        }
        hipMemcpy(h.data(), ticks, 8 * grid, hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < grid; ++i) s += h[i];
        const double us = s / grid / 100.0;
        printf("pattern %d waves/CU %d n %3d: %.2f us per wave = %.0f ns per instruction, %.1f KB/us per CU\n", pattern, waves_per_cu, n, us,
               us * 1000 / n, n * waves_per_cu / us);
      }
  return 0;
}
