// hip_cpu_emu.h — TEST INFRASTRUCTURE ONLY.  A minimal single-process emulation of the HIP
// execution model, just large enough to compile the kernels of maskrcnn-benchmark_amd/csrc/*.hip
// as plain C++ and execute them on the host: the container that builds this repo has no GPU, and
// GPU minutes are scarce, so kernel LOGIC (index arithmetic, barrier placement, LDS layout,
// ballot compaction) is first checked here against the oracle; parity proper is still the
// `-m gpu` suite on the real device.  Nothing under maskrcnn-benchmark_amd/ loads this.
//
// Model: workgroups run one after another; the threads of a workgroup are ucontext fibers that
// run round-robin and switch only at __syncthreads() / wave collectives.  Thread 0 therefore
// races far ahead of the others between barriers — a missing barrier shows up as a wrong result,
// deterministically.  A wave is 64 consecutive threads; __ballot is a wave-wide rendezvous (every
// live lane of the wave must reach it — true for the kernels here, whose ballots sit in
// wave-uniform control flow).  A rendezvous that can never complete aborts with a message.
#pragma once
#include <setjmp.h>
#include <ucontext.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <vector>

#define DETOPS_CPU_EMU 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
#define __align__(n) alignas(n)

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(8) float2 { float x, y; };
struct alignas(8) int2 { int x, y; };
struct alignas(16) int4 { int x, y, z, w; };
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }

typedef void* hipStream_t;
typedef int hipError_t;
constexpr hipError_t hipSuccess = 0;
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }

namespace emu {
constexpr int kWaveSize = 64;
constexpr size_t kStackBytes = 128 * 1024;
constexpr size_t kDynLdsBytes = 160 * 1024;

struct Wave {
  int alive = 0, arrived = 0;
  unsigned gen = 0;
  unsigned long long preds = 0, result[2] = {0, 0};
  unsigned long long xchg[2][kWaveSize];   // lane values of a wave-wide exchange (shfl / readlane)
  unsigned long long live[2];              // which lanes deposited
};

struct Block {
  dim3 grid, block, bid;
  int nthreads = 0, alive = 0, arrived = 0;
  unsigned gen = 0;
  std::vector<Wave> waves;
};

struct Fiber {
  ucontext_t ctx;   // first entry only; later switches use _setjmp/_longjmp (no sigprocmask syscalls)
  jmp_buf jb;
  char* stack = nullptr;
  bool done = false, started = false;
  dim3 tid;
  int linear = 0;
};

inline Block*& cur_block() { static Block* b = nullptr; return b; }
inline Fiber*& cur_fiber() { static Fiber* f = nullptr; return f; }
inline jmp_buf& sched_jb() { static jmp_buf j; return j; }
inline unsigned long long& progress() { static unsigned long long p = 0; return p; }
inline void* dynamic_lds() { alignas(64) static unsigned char buf[kDynLdsBytes]; return buf; }
inline std::function<void()>& body() { static std::function<void()> f; return f; }

// named work counters (DETOPS_STAT in the kernels): e.g. FMA bodies executed per wave, batches, scans
inline std::map<std::string, double>& stats() { static std::map<std::string, double> m; return m; }
inline void stat(const char* name, double n) { stats()[name] += n; }

inline void yield() {
  if (!_setjmp(cur_fiber()->jb)) _longjmp(sched_jb(), 1);
}

inline void trampoline() {
  body()();
  Fiber* f = cur_fiber();
  Block* b = cur_block();
  f->done = true;
  --b->alive;
  --b->waves[f->linear / kWaveSize].alive;
  ++progress();
  _longjmp(sched_jb(), 1);
}

inline void syncthreads() {
  Block* b = cur_block();
  const unsigned gen = b->gen;
  ++b->arrived;
  ++progress();
  while (b->gen == gen) {
    if (b->arrived >= b->alive) { b->arrived = 0; ++b->gen; ++progress(); break; }
    yield();
  }
}

inline unsigned long long ballot(bool pred) {
  Block* b = cur_block();
  Fiber* f = cur_fiber();
  Wave& w = b->waves[f->linear / kWaveSize];
  const unsigned gen = w.gen;
  if (pred) w.preds |= 1ull << (f->linear % kWaveSize);
  ++w.arrived;
  ++progress();
  while (w.gen == gen) {
    if (w.arrived >= w.alive) {
      w.result[gen & 1] = w.preds;
      w.preds = 0; w.arrived = 0; ++w.gen; ++progress();
      break;
    }
    yield();
  }
  return w.result[gen & 1];
}

// Wave-wide exchange: every live lane deposits a 64-bit value, then reads any lane's deposit.
// Returns the generation's value table (valid until two exchanges later) and the deposit mask.
inline const unsigned long long* exchange(unsigned long long v, unsigned long long* live_mask) {
  Block* b = cur_block();
  Fiber* f = cur_fiber();
  Wave& w = b->waves[f->linear / kWaveSize];
  const unsigned gen = w.gen;
  const int lane = f->linear % kWaveSize;
  if (w.arrived == 0) w.live[gen & 1] = 0;
  w.xchg[gen & 1][lane] = v;
  w.live[gen & 1] |= 1ull << lane;
  ++w.arrived;
  ++progress();
  while (w.gen == gen) {
    if (w.arrived >= w.alive) { w.arrived = 0; w.preds = 0; ++w.gen; ++progress(); break; }
    yield();
  }
  if (live_mask) *live_mask = w.live[gen & 1];
  return w.xchg[gen & 1];
}
inline int lane_id() { return cur_fiber()->linear % kWaveSize; }

// Asynchronous LDS-DMA model (glds16_async / DETOPS_VMCNT_WAIT in detops_common.h): every thread keeps the queue
// of the copies it has issued (one entry per INSTRUCTION, a null entry for a lane that was masked off); a wait for
// "at most n outstanding" lands everything but the newest n entries, oldest first.  Data therefore reaches LDS only at
// a covering wait — a kernel that reads a slot too early sees stale bytes here exactly as it would on the device.
struct DmaOp { const float* src; float* dst; };
inline std::vector<std::vector<DmaOp>>& dma_queues() { static std::vector<std::vector<DmaOp>> q; return q; }
inline void dma_issue(bool active, const float* src, float* dst) {
  auto& q = dma_queues();
  const size_t t = static_cast<size_t>(cur_fiber()->linear);
  if (q.size() <= t) q.resize(t + 1);
  q[t].push_back(active ? DmaOp{src, dst} : DmaOp{nullptr, nullptr});
}
inline void dma_wait(int n) {
  auto& q = dma_queues();
  const size_t t = static_cast<size_t>(cur_fiber()->linear);
  if (q.size() <= t) return;
  auto& mine = q[t];
  while (static_cast<int>(mine.size()) > n) {
    if (mine.front().src) memcpy(mine.front().dst, mine.front().src, 16);
    mine.erase(mine.begin());
  }
}

inline void launch(const std::function<void()>& kernel_body, dim3 grid, dim3 block, size_t lds_bytes) {
  if (lds_bytes > kDynLdsBytes) { fprintf(stderr, "emu: dynamic LDS %zu too large\n", lds_bytes); abort(); }
  const int nthreads = static_cast<int>(block.x * block.y * block.z);
  static std::vector<Fiber> fibers;
  if (static_cast<int>(fibers.size()) < nthreads) {
    const size_t old = fibers.size();
    fibers.resize(nthreads);
    for (size_t i = old; i < fibers.size(); ++i) fibers[i].stack = static_cast<char*>(malloc(kStackBytes));
  }
  body() = kernel_body;
  Block blk;
  blk.grid = grid; blk.block = block; blk.nthreads = nthreads;
  cur_block() = &blk;
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        blk.bid = dim3(bx, by, bz);
        blk.alive = nthreads; blk.arrived = 0; blk.gen = 0;
        blk.waves.assign((nthreads + kWaveSize - 1) / kWaveSize, Wave());
        for (auto& dq : dma_queues()) dq.clear();
        for (int t = 0; t < nthreads; ++t) {
          Fiber& f = fibers[t];
          f.done = false;
          f.started = false;
          f.linear = t;
          f.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
          ++blk.waves[t / kWaveSize].alive;
          getcontext(&f.ctx);
          f.ctx.uc_stack.ss_sp = f.stack;
          f.ctx.uc_stack.ss_size = kStackBytes;
          f.ctx.uc_link = nullptr;
          makecontext(&f.ctx, reinterpret_cast<void (*)()>(trampoline), 0);
        }
        while (blk.alive > 0) {
          const unsigned long long before = progress();
          for (int t = 0; t < nthreads; ++t) {
            if (fibers[t].done) continue;
            cur_fiber() = &fibers[t];
            if (!_setjmp(sched_jb())) {
              if (!fibers[t].started) { fibers[t].started = true; setcontext(&fibers[t].ctx); }
              else _longjmp(fibers[t].jb, 1);
            }
          }
          if (progress() == before) {
            fprintf(stderr, "emu: deadlock in block (%u,%u,%u): a barrier/ballot not reached by all live threads\n", bx, by, bz);
            abort();
          }
        }
      }
  cur_block() = nullptr;
  cur_fiber() = nullptr;
}
}  // namespace emu

#define threadIdx (emu::cur_fiber()->tid)
#define blockIdx (emu::cur_block()->bid)
#define blockDim (emu::cur_block()->block)
#define gridDim (emu::cur_block()->grid)

static inline void __syncthreads() { emu::syncthreads(); }
static inline unsigned long long __ballot(int pred) { return emu::ballot(pred != 0); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
static inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }

template <typename T> static inline T __shfl_down(T v, int off) {
  static_assert(sizeof(T) <= 8, "");
  unsigned long long raw = 0, live = 0; memcpy(&raw, &v, sizeof(T));
  const unsigned long long* t = emu::exchange(raw, &live);
  const int src = emu::lane_id() + off;
  if (src < emu::kWaveSize && ((live >> src) & 1ull)) { T r; memcpy(&r, &t[src], sizeof(T)); return r; }
  return v;
}
template <typename T> static inline T __shfl_up(T v, int off) {
  static_assert(sizeof(T) <= 8, "");
  unsigned long long raw = 0, live = 0; memcpy(&raw, &v, sizeof(T));
  const unsigned long long* t = emu::exchange(raw, &live);
  const int src = emu::lane_id() - off;
  if (src >= 0 && ((live >> src) & 1ull)) { T r; memcpy(&r, &t[src], sizeof(T)); return r; }
  return v;
}
template <typename T> static inline T __shfl(T v, int src) {
  static_assert(sizeof(T) <= 8, "");
  unsigned long long raw = 0, live = 0; memcpy(&raw, &v, sizeof(T));
  const unsigned long long* t = emu::exchange(raw, &live);
  if (src >= 0 && src < emu::kWaveSize && ((live >> src) & 1ull)) { T r; memcpy(&r, &t[src], sizeof(T)); return r; }
  return v;
}
// v_mfma_f32_32x32x16_{f16,bf16}: D = A (32x16) * B (16x32) + C.  Fragment layout (the documented gfx950 one):
// lane l holds A[l & 31][8 * (l >> 5) + 0..7] and B[8 * (l >> 5) + 0..7][l & 31]; C/D element `reg` of lane l is
// row (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5), column l & 31.  All 64 lanes must execute it.
template <typename V8, typename E>
static inline __attribute__((ext_vector_type(16))) float emu_mfma_f32_32x32x16(V8 a, V8 b,
                                                                                 __attribute__((ext_vector_type(16))) float c) {
  static_assert(sizeof(V8) == 16, "");
  unsigned long long ta[2][emu::kWaveSize], tb[2][emu::kWaveSize], raw[2];
  memcpy(raw, &a, 16);
  for (int h = 0; h < 2; ++h) { const unsigned long long* t = emu::exchange(raw[h], nullptr); memcpy(ta[h], t, sizeof(ta[h])); }
  memcpy(raw, &b, 16);
  for (int h = 0; h < 2; ++h) { const unsigned long long* t = emu::exchange(raw[h], nullptr); memcpy(tb[h], t, sizeof(tb[h])); }
  const int lane = emu::lane_id(), j = lane & 31;
  for (int reg = 0; reg < 16; ++reg) {
    const int i = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
    float sum = c[reg];
    for (int k = 0; k < 16; ++k) {
      E ea[8], eb[8];
      unsigned long long pa[2] = {ta[0][i + 32 * (k / 8)], ta[1][i + 32 * (k / 8)]};
      unsigned long long pb[2] = {tb[0][j + 32 * (k / 8)], tb[1][j + 32 * (k / 8)]};
      memcpy(ea, pa, 16); memcpy(eb, pb, 16);
      sum += static_cast<float>(ea[k % 8]) * static_cast<float>(eb[k % 8]);
    }
    c[reg] = sum;
  }
  return c;
}
static inline unsigned emu_readlane(unsigned v, int lane) {
  const unsigned long long* t = emu::exchange(v, nullptr);
  return static_cast<unsigned>(t[lane]);
}
static inline unsigned emu_readfirstlane(unsigned v) {
  unsigned long long live = 0;
  const unsigned long long* t = emu::exchange(v, &live);
  return static_cast<unsigned>(t[__builtin_ctzll(live)]);
}
#define __builtin_amdgcn_readlane(v, l) emu_readlane(v, l)
#define __builtin_amdgcn_readfirstlane(v) emu_readfirstlane(v)
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
#define __builtin_amdgcn_s_setprio(p) ((void)0)

static inline unsigned long long atomicOr(unsigned long long* p, unsigned long long v) { const unsigned long long o = *p; *p = o | v; return o; }

// fibers never run concurrently: plain read-modify-write is atomic here
static inline float atomicAdd(float* p, float v) { const float o = *p; *p = o + v; return o; }
static inline double atomicAdd(double* p, double v) { const double o = *p; *p = o + v; return o; }
static inline int atomicAdd(int* p, int v) { const int o = *p; *p = o + v; return o; }
static inline int atomicMin(int* p, int v) { const int o = *p; *p = std::min(o, v); return o; }
static inline int atomicMax(int* p, int v) { const int o = *p; *p = std::max(o, v); return o; }

static inline unsigned atomicAdd(unsigned* p, unsigned v) { const unsigned o = *p; *p = o + v; return o; }
static inline unsigned atomicCAS(unsigned* p, unsigned cmp, unsigned val) { const unsigned o = *p; if (o == cmp) *p = val; return o; }
static inline int atomicCAS(int* p, int cmp, int val) { const int o = *p; if (o == cmp) *p = val; return o; }
static inline int atomicSub(int* p, int v) { const int o = *p; *p = o - v; return o; }

// 16-bit storage types (fp32 arithmetic everywhere in the kernels): IEEE half via _Float16,
// bfloat16 with round-to-nearest-even like __float2bfloat16
struct __half { _Float16 v; };
static inline float __half2float(__half h) { return static_cast<float>(h.v); }
static inline __half __float2half(float f) { __half h; h.v = static_cast<_Float16>(f); return h; }
static inline unsigned short __half_as_ushort(__half h) { unsigned short u; memcpy(&u, &h, 2); return u; }
static inline __half __ushort_as_half(unsigned short u) { __half h; memcpy(&h, &u, 2); return h; }
struct __hip_bfloat16 { unsigned short bits; };
static inline float __bfloat162float(__hip_bfloat16 b) { return __uint_as_float(static_cast<unsigned>(b.bits) << 16); }
static inline __hip_bfloat16 __float2bfloat16(float f) {
  unsigned u = __float_as_uint(f);
  __hip_bfloat16 b;
  if ((u & 0x7fffffffu) > 0x7f800000u) { b.bits = static_cast<unsigned short>((u >> 16) | 0x40); return b; }  // NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  b.bits = static_cast<unsigned short>(u >> 16);
  return b;
}

template <typename T> static inline T min(T a, T b) { return a < b ? a : b; }
template <typename T> static inline T max(T a, T b) { return a > b ? a : b; }

#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) \
  emu::launch([&]() { kernel(__VA_ARGS__); }, grid, block, lds)

extern "C" __attribute__((visibility("default"), used, weak)) int detops_emu_stats(char* buf, int cap, int reset) {
  std::string out;
  for (auto& kv : emu::stats()) out += kv.first + "=" + std::to_string(kv.second) + "\n";
  if (reset) emu::stats().clear();
  if (buf && cap > 0) { strncpy(buf, out.c_str(), cap - 1); buf[cap - 1] = 0; }
  return static_cast<int>(out.size());
}
