// detops_emu_shims.h — host-emulation stand-ins for every device primitive of maskrcnn-benchmark_amd/csrc/detops_common.h
// (same names, same contracts; see the comments there).  Test infrastructure: only tests/emu builds with DETOPS_CPU_EMU.
#pragma once
#include "hip_cpu_emu.h"

#define DETOPS_DYNAMIC_LDS(T, name) T* name = reinterpret_cast<T*>(emu::dynamic_lds())

#define DETOPS_STAT(name, n) emu::stat(name, n)

#define DETOPS_OCCUPANCY(out, kernel, block, lds) ((out) = 4)

#define DETOPS_LDS_BARRIER() __syncthreads()

#define DETOPS_PIN4(a, b, c, d) ((void)0)

#define DETOPS_WAVE_SYNC() ((void)__ballot(1))
#define DETOPS_KEEP_TOGETHER3(a, b, c) ((void)0)
#define DETOPS_PIN6(a, b, c, d, e, f) ((void)0)

__device__ __forceinline__ void glds16(const float* g, float* lds_wave_base) {
  float* d = lds_wave_base + 4 * (threadIdx.x & 63);
  d[0] = g[0]; d[1] = g[1]; d[2] = g[2]; d[3] = g[3];
}
__device__ __forceinline__ void glds16_async(bool active, const float* g, float* lds_wave_base) {
  emu::dma_issue(active, g, lds_wave_base + 4 * (threadIdx.x & 63));
}
__device__ __forceinline__ void glds16_async_so(bool active, const float* sbase, unsigned voff_bytes, float* lds_wave_base) {
  emu::dma_issue(active, reinterpret_cast<const float*>(reinterpret_cast<const char*>(sbase) + voff_bytes),
                 lds_wave_base + 4 * (threadIdx.x & 63));
}
#define DETOPS_VMCNT_WAIT(n) emu::dma_wait(n)

__device__ __forceinline__ void store_f4_wt(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

#define DETOPS_ACQUIRE_AGENT() ((void)0)

__device__ __forceinline__ void detops_release_agent() {}
__device__ __forceinline__ void flag_store(detops_u64* p, detops_u64 v) { *p = v; }
__device__ __forceinline__ void flag_store_relaxed(detops_u64* p, detops_u64 v) { *p = v; }
__device__ __forceinline__ detops_u64 flag_load(const detops_u64* p) { return *p; }
__device__ __forceinline__ int flag_load(const int* p) { return *p; }
__device__ __forceinline__ void flag_add(int* p, int v) { *p += v; }
__device__ __forceinline__ detops_u64 flag_peek(const detops_u64* p) { return *p; }
__device__ __forceinline__ int flag_peek(const int* p) { return *p; }
__device__ __forceinline__ void store_u64_wt(detops_u64* p, detops_u64 v) { *p = v; }
__device__ __forceinline__ void store_u32_wt(void* p, unsigned v) { *static_cast<unsigned*>(p) = v; }
__device__ __forceinline__ float load_f32_coherent(const float* p) { return *p; }
__device__ __forceinline__ bool spin_again(int& budget) {
  // workgroups run one after another here: a wait that is not satisfied never will be.  With a TEST budget (tuning
  // "nms_spin_budget": a few hundred polls, fault injection) the wait gives up like a starved one does on the device;
  // with the library's default budget it is a dispatch-order bug of the kernel: abort loudly.
  if (budget < (1 << 20)) return false;
  fprintf(stderr, "emu: a workgroup waits for a flag no earlier workgroup has set\n");
  abort();
}

__device__ __forceinline__ int detops_fetch_add_relaxed(int32_t* p, int v) { const int o = *p; *p = o + v; return o; }
__device__ __forceinline__ void detops_atomic_add2(__half* p, float v0, float v1) {
  p[0] = __float2half(__half2float(p[0]) + v0); p[1] = __float2half(__half2float(p[1]) + v1);
}
__device__ __forceinline__ void detops_atomic_add2(__hip_bfloat16* p, float v0, float v1) {
  p[0] = __float2bfloat16(__bfloat162float(p[0]) + v0); p[1] = __float2bfloat16(__bfloat162float(p[1]) + v1);
}
__device__ __forceinline__ long long detops_wall_clock() { return 0; }

__device__ __forceinline__ float detops_exp(float x) { return expf(x); }
__device__ __forceinline__ float detops_log(float x) { return logf(x); }
#define DETOPS_MFMA_32x32x16_F16(a, b, c) emu_mfma_f32_32x32x16<decltype(a), _Float16>(a, b, c)
#define DETOPS_MFMA_32x32x16_BF16(a, b, c) emu_mfma_f32_32x32x16<decltype(a), __bf16>(a, b, c)

template <typename K>
static inline int detops_resident_workgroups(K kernel, int block, size_t lds) {
  (void)kernel; (void)block; (void)lds;
  return -1;
}
