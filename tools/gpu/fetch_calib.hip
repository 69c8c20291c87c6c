// FETCH_SIZE calibration (gfx950, rocprofv3): kernels that read a > L3 buffer EXACTLY ONCE in a known pattern, so that the
// number of 128-byte L2 lines they must fill is known by construction (MI355X_MICROARCH.md "HBM": FETCH_SIZE is calibrated
// for wide coalesced reads only — x2 — and must be calibrated on a known byte count for any other access pattern).
//   wide        16 bytes per lane, lanes contiguous: every line of the buffer once            lines = bytes / 128
//   rowpiece    the ROIAlign forward's shape: of every 1344-byte map row (336 floats) ONE 80-byte piece (5 lanes x 16
//               bytes) starting 12 bytes into the row; rows alternate between 128-byte alignment and +64, so even rows
//               touch 1 line (12..92) and odd rows 2 (76..156)                              lines = 1.5 * rows
//   rowpiece_lds  the same addresses through global_load_lds_dwordx4 (LDS-DMA), as the kernel issues them
// Prints the expected line bytes per kernel; run once per counter set under rocprofv3 --pmc (tools/gpu/r05j_fetch_calib.sh).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) wide(const float4* src, size_t n4, float* sink) {
  float acc = 0.f;
  for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < n4; i += static_cast<size_t>(gridDim.x) * 256) {
    const float4 v = src[i]; acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 12345.678f) sink[0] = acc;
}
// 64 lanes = 12 rows x 5 pieces (4 lanes idle); a wave takes rows [12 k, 12 k + 12)
__global__ void __launch_bounds__(64) rowpiece(const char* src, size_t rows, float* sink) {
  const int lane = threadIdx.x, r = lane / 5, v = lane % 5;
  float acc = 0.f;
  for (size_t r0 = static_cast<size_t>(blockIdx.x) * 12; r0 < rows; r0 += static_cast<size_t>(gridDim.x) * 12) {
    if (lane < 60 && r0 + r < rows) {
      const float* p = reinterpret_cast<const float*>(src + (r0 + r) * 1344 + 12 + v * 16);
      acc += p[0] + p[1] + p[2] + p[3];
    }
  }
  if (acc == 12345.678f) sink[0] = acc;
}
__global__ void __launch_bounds__(64) rowpiece_lds(const char* src, size_t rows, float* sink) {
  extern __shared__ float lds[];
  typedef __attribute__((address_space(3))) float* lds_fptr_t;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(reinterpret_cast<uintptr_t>((lds_fptr_t)lds)));
  const int lane = threadIdx.x, r = lane / 5, v = lane % 5;
  int it = 0;
  for (size_t r0 = static_cast<size_t>(blockIdx.x) * 12; r0 < rows; r0 += static_cast<size_t>(gridDim.x) * 12, ++it) {
    const bool on = lane < 60 && r0 + r < rows;
    const char* base = src + r0 * 1344;
    const unsigned voff = on ? static_cast<unsigned>(r * 1344 + 12 + v * 16) : 0u;
    unsigned keep;
    const unsigned dst = lds0 + (it & 7) * 1024;
    if (on || true)   // inactive lanes re-read the wave's first piece (an L1 hit): the instruction stays wave-wide
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(on ? voff : 12u), "s"(base), "s"(dst) : "memory");
    if ((it & 7) == 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (sink && lds[lane] == 12345.678f) sink[0] = lds[lane];
}
int main() {
  const size_t bytes = 768ull << 20;                      // 3 x the 256 MiB Infinity Cache
  char* src; hipMalloc(&src, bytes); hipMemset(src, 0, bytes);
  float* sink; hipMalloc(&sink, 64);
  const size_t rows = bytes / 1344;
  hipDeviceSynchronize();
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(wide, dim3(4096), dim3(256), 0, 0, reinterpret_cast<const float4*>(src), bytes / 16, sink);
    hipLaunchKernelGGL(rowpiece, dim3(8192), dim3(64), 0, 0, src, rows, sink);
    hipLaunchKernelGGL(rowpiece_lds, dim3(8192), dim3(64), 8192, 0, src, rows, sink);
  }
  hipDeviceSynchronize();
  printf("expected 128-byte line fills per launch: wide %.1f MB | rowpiece, rowpiece_lds %.1f MB (rows %zu x 1.5 lines; useful bytes %.1f MB)\n",
         bytes / 1e6, rows * 1.5 * 128 / 1e6, rows, rows * 80 / 1e6);
  return 0;
}
