// nms.hip — device-only greedy NMS for gfx950 (MI355X): single and batched/segmented.
//
// Replaces _C.nms (reference csrc/nms.h:10-28).  The CONTRACT is the reference CPU kernel
// (csrc/cpu/nms_cpu.cpp:5-65): boxes visited in descending score order, box j suppressed iff
// IoU(i,j) >= threshold for an earlier kept box i, "+1" pixel convention, result = ascending
// ORIGINAL indices (int64).  The reference CUDA path (csrc/cuda/nms.cu) uses `>` instead of `>=`,
// copies the n x n/64 bitmask to the host and scans it there; here everything stays on the
// device and in stream order (no host sync, graph-capturable):
//
//   1. sort      one workgroup per segment: 64-bit keys (~orderable(score) << 32 | index) sorted
//                ascending by an LDS bitonic network (n <= 8192) — score descending, ties by
//                ascending index (stable, like the CPU sort); gathers boxes + areas in order.
//                (n > 8192: one hipCUB radix sort of the same keys per segment.)
//   2. mask      64x64 tiles, one wavefront per tile, upper triangle only: lane r holds row box r,
//                the 64 column boxes sit in LDS; bit c of word (row, colblock) = IoU >= thr.
//                IoU arithmetic follows nms_cpu.cpp:49-60 operation by operation with FP
//                contraction off and IEEE division -> bit-exact decisions.
//   3. scan      one workgroup (8 waves) per segment walks the 64-row blocks: wave 0 resolves the
//                diagonal tile serially over the *alive* rows only (s_ff1 + v_readlane), then all
//                waves OR the kept rows' mask words into the pending `removed` words (LDS) for
//                the later column blocks; finally kept sorted positions are scattered to a bitset
//                in ORIGINAL index space and compacted with a workgroup prefix sum -> ascending
//                original indices, exactly at::nonzero(suppressed == 0) (nms_cpu.cpp:64).
//
// n <= 4096 per segment (every RPN / box-head call of the detector): the three stages are ONE launch,
// nms_fused_kernel — workgroups [0,S) sort, the next S*ceil(nb(nb+1)/16) compute mask tiles (a wave per tile; they wait for
// their segment's "sorted" token), the last S run the scan chain as the tiles of each row block are counted in.
// Larger problems keep the three launches.
#include <atomic>
#include <random>

#include "detops_devlib.h"

namespace {

using u64 = unsigned long long;

constexpr int kSortLdsMax = 8192;   // bitonic-in-LDS capacity (64 KiB of keys)
constexpr int kScanMaxN = 65536;    // removed-words / bitset capacity of the scan kernel
constexpr int kScanThreads = 512;
constexpr int kMaxWords = kScanMaxN / 64;  // 1024 column blocks

// Diagnostic timeline of ONE fused launch (tuning nms_debug & 4; tools/gpu/nms_probe.py): 10 ns wall-clock ticks of segment 0's
// workgroups.  [0] first sort start, [1] token published, [2] scan start, [3] token seen by the scan, [4] chain end, [5] kernel end of
// the scan workgroup; [128 + 3 t ..] = tile t start / go / counted.
constexpr int kNmsTimelineWords = 128 + 3 * 2112;
__device__ long long g_nms_timeline[kNmsTimelineWords];
#define NMS_TL(cond, idx) do { if ((scan_first & 16) && (cond)) g_nms_timeline[(idx)] = detops_wall_clock(); } while (0)

__device__ __forceinline__ u64 make_key(float score, unsigned idx) {
  if (score == 0.f) score = 0.f;  // -0.0 and +0.0 compare equal on the CPU path
  unsigned b = __float_as_uint(score);
  const unsigned asc = (b & 0x80000000u) ? ~b : (b | 0x80000000u);  // ascending-orderable
  return (static_cast<u64>(~asc) << 32) | idx;                      // descending score, then index
}

struct SegView {
  int begin, n;
};

__device__ __forceinline__ SegView seg_view(const int32_t* __restrict__ seg_offsets, int n_single,
                                            int s) {
  if (!seg_offsets) return SegView{0, n_single};
  const int b = seg_offsets[s];
  return SegView{b, seg_offsets[s + 1] - b};
}

// Workspace carve (per segment region of `stride` rows / `nbmax` words per row).
struct Work {
  float4* boxes;  // [S * stride] sorted boxes
  float* areas;   // [S * stride]
  int32_t* order; // [S * stride] sorted position -> local original index
  u64* mask;      // [S * mrows * nbmax]; mrows = stride rounded up to whole 64-row blocks (the scan prefetches whole blocks)
  int stride, nbmax, mrows;
};

// ---------------------------------------------------------------------------- 1. sort + gather
// Bitonic sort of one segment's keys in LDS + gather of the boxes in sorted order.  The compare-exchange pairs of a
// stage with j <= 64 stay inside the 128-key chunk the SAME wave handled in the previous stage (pair t covers keys
// 2*(t - t%j) + t%j and + j; 64 consecutive t's = keys [2*t0, 2*t0 + 128)), so those stages need no workgroup
// barrier: 56 of the 66 stages at n = 2048.
// Workgroup-wide AND of a per-thread predicate (two barriers; the flag is a workgroup-shared word).
__device__ __forceinline__ bool block_all(bool ok) {
  __shared__ int s_all;
  if (threadIdx.x == 0) s_all = 1;
  __syncthreads();
  if (!ok) s_all = 0;            // every writer stores the same value
  __syncthreads();
  return s_all != 0;
}

// ALREADY SORTED?  The detector's callers hand over top-k output (modeling/rpn/inference.py: `topk(sorted=True)`, then the
// fused decode keeps that order): scores descend and equal scores sit in position order, which is exactly ascending keys.
// The keys are unique, so the sorted sequence is unique: skipping the network on such input changes nothing in the result
// and takes the 15-25 us of the 66-stage network off the head of every segment's critical path.  `nms_no_presorted` (tuning)
// switches the test off for A/B runs.
template <bool WT>   // WT: the sorted rows are consumed by other workgroups of the same launch (write-through stores)
__device__ __forceinline__ void sort_and_gather(u64* keys, const float* __restrict__ boxes,
                                                const float* __restrict__ scores, SegView sv, int npad, const Work& w, int s,
                                                bool check_sorted = true) {
  const int n = sv.n;
  // smallest power of two >= n (uniform), bounded by npad (the host-side capacity)
  int np = 2;
  while (np < n) np <<= 1;
  if (np > npad) np = npad;  // cannot happen when the caller honoured max_n
  for (int i = threadIdx.x; i < np; i += blockDim.x)
    keys[i] = (i < n) ? make_key(scores[sv.begin + i], static_cast<unsigned>(i)) : ~0ull;
  __syncthreads();
  bool sorted = check_sorted;
  if (sorted) {
    for (int i = threadIdx.x; i + 1 < np; i += blockDim.x) sorted = sorted && keys[i] <= keys[i + 1];
    sorted = block_all(sorted);
  }
  for (int k = 2; k <= np && !sorted; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = threadIdx.x; t < (np >> 1); t += blockDim.x) {
        // t-th compare-exchange pair of this stage: i has bit j clear, partner = i | j
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const int p = i | j;
        const u64 a = keys[i], b = keys[p];
        const bool up = (i & k) == 0;
        if ((a > b) == up) { keys[i] = b; keys[p] = a; }
      }
      const int jn = j > 1 ? (j >> 1) : k;   // the next stage's distance (k: first stage of the next merge)
      if (j > kWave || jn > kWave) __syncthreads();
      else DETOPS_WAVE_SYNC();
    }
  }
  __syncthreads();
  float4* ob = w.boxes + static_cast<size_t>(s) * w.stride;
  float* oa = w.areas + static_cast<size_t>(s) * w.stride;
  int32_t* oo = w.order + static_cast<size_t>(s) * w.stride;
  const float4* ib = reinterpret_cast<const float4*>(boxes) + sv.begin;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
#pragma clang fp contract(off)
    const int src = static_cast<int>(keys[i] & 0xffffffffu);
    const float4 b = ib[src];
    const float area = (b.z - b.x + 1.f) * (b.w - b.y + 1.f);  // nms_cpu.cpp:22
    if (WT) {
      store_f4_wt(reinterpret_cast<float*>(ob + i), b);
      store_u32_wt(oa + i, __float_as_uint(area));
      store_u32_wt(oo + i, static_cast<unsigned>(src));
    } else {
      ob[i] = b;
      oa[i] = area;
      oo[i] = src;
    }
  }
}

// The same network for n <= 4 * blockDim keys with the keys in REGISTERS: thread t holds keys 4t .. 4t+3.  Partner
// distance j = 1, 2: inside the thread (21 of the 66 stages of a 2048-key sort); j = 4 .. 128: the same slot of lane
// (t ^ j/4) of the same wave, fetched with a lane shuffle (39 stages, no LDS memory, no barrier); j >= 256: another
// wave — the four keys go through LDS once (6 stages).  Element e, partner e ^ j: it keeps the smaller key iff
// ((e & j) == 0) == ((e & k) == 0).  `xch` is LDS for 4 * blockDim keys.
__device__ __forceinline__ u64 shfl_u64(u64 v, int src_lane) {
  const unsigned lo = __shfl(static_cast<unsigned>(v), src_lane);
  const unsigned hi = __shfl(static_cast<unsigned>(v >> 32), src_lane);
  return (static_cast<u64>(hi) << 32) | lo;
}

template <bool WT>
__device__ __forceinline__ void sort_and_gather_reg(u64* xch, const float* __restrict__ boxes,
                                                    const float* __restrict__ scores, SegView sv, const Work& w, int s,
                                                    bool check_sorted = true) {
  const int n = sv.n, tid = threadIdx.x, lane = tid & (kWave - 1);
  const int np = 4 * static_cast<int>(blockDim.x);
  u64 key[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int e = 4 * tid + r;
    key[r] = e < n ? make_key(scores[sv.begin + e], static_cast<unsigned>(e)) : ~0ull;
  }
  bool sorted = check_sorted;
  if (sorted) {
    // thread t's four keys ascend and its last one does not exceed thread t + 1's first (lane 63: through LDS)
    const int wave = tid / kWave, nw = static_cast<int>(blockDim.x) / kWave;
    if (lane == 0) xch[wave] = key[0];
    u64 next0 = shfl_u64(key[0], (lane + 1) & (kWave - 1));
    __syncthreads();
    if (lane == kWave - 1) next0 = wave + 1 < nw ? xch[wave + 1] : ~0ull;
    sorted = block_all(key[0] <= key[1] && key[1] <= key[2] && key[2] <= key[3] && key[3] <= next0);   // (its barriers also release xch)
  }
  for (int k = 2; k <= np && !sorted; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j < 4) {                                  // partner in this thread: slots (0,1),(2,3) for j = 1, (0,2),(1,3) for j = 2
        auto cx = [&](u64& a, u64& b, int r) {      // compare-exchange of slots r < r ^ j (constant register indices)
          const bool up = ((4 * tid + r) & k) == 0;
          const bool sw = (a > b) == up;
          const u64 lo = sw ? b : a, hi = sw ? a : b;
          a = lo; b = hi;
        };
        if (j == 1) { cx(key[0], key[1], 0); cx(key[2], key[3], 2); }
        else        { cx(key[0], key[2], 0); cx(key[1], key[3], 1); }
        continue;
      }
      u64 other[4];
      const int dt = j >> 2;                        // partner thread = tid ^ dt, same slot
      if (dt < kWave) {
#pragma unroll
        for (int r = 0; r < 4; ++r) other[r] = shfl_u64(key[r], lane ^ dt);
      } else {
        __syncthreads();                            // previous readers of xch are done
#pragma unroll
        for (int r = 0; r < 4; ++r) xch[4 * tid + r] = key[r];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) other[r] = xch[4 * (tid ^ dt) + r];
      }
      const bool lower = (tid & dt) == 0;           // (e & j) == 0 for all four slots
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool up = ((4 * tid + r) & k) == 0;
        const u64 a = key[r], b = other[r];
        const u64 mn = a < b ? a : b, mx = a < b ? b : a;
        key[r] = (lower == up) ? mn : mx;
      }
    }
  }
  float4* ob = w.boxes + static_cast<size_t>(s) * w.stride;
  float* oa = w.areas + static_cast<size_t>(s) * w.stride;
  int32_t* oo = w.order + static_cast<size_t>(s) * w.stride;
  const float4* ib = reinterpret_cast<const float4*>(boxes) + sv.begin;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
#pragma clang fp contract(off)
    const int i = 4 * tid + r;
    if (i >= n) continue;
    const int src = static_cast<int>(key[r] & 0xffffffffu);
    const float4 b = ib[src];
    const float area = (b.z - b.x + 1.f) * (b.w - b.y + 1.f);  // nms_cpu.cpp:22
    if (WT) {
      store_f4_wt(reinterpret_cast<float*>(ob + i), b);
      store_u32_wt(oa + i, __float_as_uint(area));
      store_u32_wt(oo + i, static_cast<unsigned>(src));
    } else {
      ob[i] = b;
      oa[i] = area;
      oo[i] = src;
    }
  }
}

__global__ void __launch_bounds__(1024)
nms_sort_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                const int32_t* __restrict__ seg_offsets, int n_single, int npad, Work w) {
  DETOPS_DYNAMIC_LDS(unsigned char, smem_raw);
  const int s = blockIdx.x;
  sort_and_gather<false>(reinterpret_cast<u64*>(smem_raw), boxes, scores, seg_view(seg_offsets, n_single, s), npad, w, s);
}

// large-n path (n > 8192, any number of segments): per-segment key rows padded to `stride` with
// ~0 keys -> one hipcub radix sort per segment (host loop; the reference's non-FPN configs have one
// 12000-candidate problem per image) -> gather.  blockIdx.y = segment.
__global__ void nms_make_keys_kernel(const float* __restrict__ scores, const int32_t* __restrict__ seg_offsets,
                                     int n_single, int stride, u64* __restrict__ keys) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int s = blockIdx.y;
  if (i >= stride) return;
  const SegView sv = seg_view(seg_offsets, n_single, s);
  keys[static_cast<size_t>(s) * stride + i] =
      (i < sv.n) ? make_key(scores[sv.begin + i], static_cast<unsigned>(i)) : ~0ull;
}

__global__ void nms_gather_kernel(const float* __restrict__ boxes, const u64* __restrict__ keys,
                                  const int32_t* __restrict__ seg_offsets, int n_single, Work w) {
#pragma clang fp contract(off)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int s = blockIdx.y;
  const SegView sv = seg_view(seg_offsets, n_single, s);
  if (i >= sv.n) return;
  const size_t o = static_cast<size_t>(s) * w.stride + i;
  const int src = static_cast<int>(keys[o] & 0xffffffffu);
  const float4 b = (reinterpret_cast<const float4*>(boxes) + sv.begin)[src];
  w.boxes[o] = b;
  w.areas[o] = (b.z - b.x + 1.f) * (b.w - b.y + 1.f);
  w.order[o] = src;
}

// ---------------------------------------------------------------------------- 2. IoU bitmask
// One wave = one 64x64 tile: lane r holds row box r, the 64 column boxes are broadcast lane by lane (v_readlane ->
// scalar operands; no LDS, no barrier).  Every lane runs the whole loop (rows / columns past n carry zeros and are
// masked off), so the broadcasts stay wave-wide.
__device__ __forceinline__ float lane_bcast(float v, int c) {
  return __int_as_float(static_cast<int>(__builtin_amdgcn_readlane(static_cast<unsigned>(__float_as_int(v)), c)));
}

template <bool WT, int TRIP = kWave>   // WT: the word is published to another workgroup of the same launch (write-through store); TRIP = 0: timing ablation
__device__ __forceinline__ void mask_tile(const float4* __restrict__ sb, const float* __restrict__ sa, u64* __restrict__ mask,
                                          int n, int nb, int rb, int cb, float thr) {
  const int lane = threadIdx.x & (kWave - 1);
  const int ncol = min(kWave, n - cb * kWave);
  const int col = cb * kWave + lane, row = rb * kWave + lane;
  float4 cbx = make_float4(0.f, 0.f, 0.f, 0.f);
  float car = 0.f;
  float4 a;
  float iarea;
  const int rrow = min(row, n - 1);
  if (WT) {   // the sorted rows were published by the sort workgroup of this launch
    const float* fb = reinterpret_cast<const float*>(sb);
    if (lane < ncol) {
      cbx = make_float4(load_f32_coherent(fb + 4 * col), load_f32_coherent(fb + 4 * col + 1),
                        load_f32_coherent(fb + 4 * col + 2), load_f32_coherent(fb + 4 * col + 3));
      car = load_f32_coherent(sa + col);
    }
    a = make_float4(load_f32_coherent(fb + 4 * rrow), load_f32_coherent(fb + 4 * rrow + 1),
                    load_f32_coherent(fb + 4 * rrow + 2), load_f32_coherent(fb + 4 * rrow + 3));
    iarea = load_f32_coherent(sa + rrow);
  } else {
    if (lane < ncol) { cbx = sb[col]; car = sa[col]; }
    a = sb[rrow];
    iarea = sa[rrow];
  }
  u64 bits = 0;
  // Diagonal tile: the FULL symmetric relation of the row (every c != lane).  The comparison is symmetric bit for
  // bit (max / min / fp add commute, `iarea + carea - inter` sees the same two addends), so the bits below the
  // diagonal are "rows j < lane that suppress lane" — the transposed view the scan's fixed-point resolve needs.
#pragma unroll 8
  for (int c = 0; c < TRIP; ++c) {
#pragma clang fp contract(off)
    const float bx = lane_bcast(cbx.x, c), by = lane_bcast(cbx.y, c), bz = lane_bcast(cbx.z, c), bw = lane_bcast(cbx.w, c);
    const float ca = lane_bcast(car, c);
    const float xx1 = fmaxf(a.x, bx), yy1 = fmaxf(a.y, by);
    const float xx2 = fminf(a.z, bz), yy2 = fminf(a.w, bw);
    const float ww = fmaxf(0.f, xx2 - xx1 + 1.f);
    const float hh = fmaxf(0.f, yy2 - yy1 + 1.f);
    const float inter = ww * hh;
    bool sup;
    if (inter > 0.f || thr <= 0.f) {
      const float ovr = inter / (iarea + ca - inter);  // IEEE fp32 division
      sup = ovr >= thr;                                // nms_cpu.cpp:60
    } else {
      sup = false;  // inter == 0 -> ovr is +-0 (or NaN for a zero union): never >= a positive thr
    }
    if (sup && !(rb == cb && c == lane)) bits |= 1ull << c;
  }
  if (ncol < kWave) bits &= (1ull << ncol) - 1;
  if (row < n) {
    if (WT) store_u64_wt(mask + static_cast<size_t>(row) * nb + cb, bits);
    else mask[static_cast<size_t>(row) * nb + cb] = bits;
  }
}

__global__ void __launch_bounds__(kWave)
nms_mask_kernel(const int32_t* __restrict__ seg_offsets, int n_single, float thr, Work w) {
  const int cb = blockIdx.x, rb = blockIdx.y, s = blockIdx.z;
  if (cb < rb) return;  // upper triangle only
  const int n = seg_view(seg_offsets, n_single, s).n;
  if (rb * kWave >= n || cb * kWave >= n) return;
  mask_tile<false>(w.boxes + static_cast<size_t>(s) * w.stride, w.areas + static_cast<size_t>(s) * w.stride,
                   w.mask + static_cast<size_t>(s) * w.mrows * w.nbmax, n, (n + kWave - 1) / kWave, rb, cb, thr);
}

// ---------------------------------------------------------------------------- 3. scan + compaction
__device__ __forceinline__ u64 readlane64(u64 v, int lane) {
  const unsigned lo = __builtin_amdgcn_readlane(static_cast<unsigned>(v), lane);
  const unsigned hi = __builtin_amdgcn_readlane(static_cast<unsigned>(v >> 32), lane);
  return (static_cast<u64>(hi) << 32) | lo;
}

__device__ __forceinline__ u64 uniform64(u64 v) {  // value known to be wave-uniform -> SGPRs
  const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v));
  const unsigned hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v >> 32));
  return (static_cast<u64>(hi) << 32) | lo;
}

// The greedy chain of one segment with n <= 4096, walked by ONE wave with the pending "removed" words in registers.
// A row of the mask has nb <= 64 words; with nb <= 32 (16) a wave-wide load covers RPI = 2 (4) rows at once: lane =
// (sub, col) holds column word `col` of row t*RPI + sub in iteration t, and accumulates the removed bits of the rows
// = sub (mod RPI) — the RPI partial words of a column are OR-ed when the chain reaches that column block.  Per row
// block: resolve the diagonal tile (ballot fixed point over the kept rows), then 64/RPI iterations of
// `removed |= row & -(kept bit)` with the same registers refilled from the NEXT block (rows / columns outside the
// segment pick up words nobody consumes: no per-load predication; the mask region is padded to whole blocks).
// `done` (fused launch): tile counters per row block; a block's rows are only requested once its nb - rb tiles are in.
// Returns false when a wait for the tile counters ran out of its polling budget (fused launch only): the caller then
// reports the segment as failed instead of publishing a keep set built from unpublished rows.
template <int RPI>
__device__ __forceinline__ bool scan_chain(const u64* __restrict__ mask, int n, int nb, const int* done, u64* keptw, int spin_budget = kSpinBudget) {
  constexpr int CW = kWave / RPI, NIT = kWave / RPI;
  const int lane = threadIdx.x & (kWave - 1);
  const int sub = lane / CW, col = lane % CW;
  const int ccol = min(col, nb - 1);
  u64 ready = done ? 0ull : ~0ull;
  int budget = spin_budget;
  bool timed_out = false;
  auto ensure = [&](int rbx) {                       // uniform: the tiles of row block rbx are in memory
    if ((ready >> rbx) & 1ull) return;
    for (;;) {
      const bool ok = lane < nb ? flag_peek(done + lane) >= nb - lane : true;
      ready = __ballot(ok);
      if ((ready >> rbx) & 1ull) break;
      if (!spin_again(budget)) { timed_out = true; break; }
    }
    DETOPS_ACQUIRE_AGENT();                          // the rows of every block counted complete are now loadable
  };
  u64 removed = 0;                                   // lane (sub, c): pending removed bits of column block c, rows = sub
  u64 sym_next = 0;                                  // lane i: symmetric diagonal-tile word of row rb*64 + i
  ensure(0);
  if (lane < n) sym_next = mask[static_cast<size_t>(lane) * nb];
  const u64 below_me = (1ull << lane) - 1;
  u64 v[NIT];
  {
    const u64* p = mask + static_cast<size_t>(sub) * nb + ccol;
#pragma unroll
    for (int t = 0; t < NIT; ++t) v[t] = p[static_cast<size_t>(t * RPI) * nb];
  }
  for (int rb = 0; rb < nb; ++rb) {
    const int nrow = min(kWave, n - rb * kWave);
    const u64 below = sym_next & below_me;           // rows j < lane of this block that suppress row lane
    sym_next = 0;
    const int rbn = min(rb + 1, nb - 1);             // last block: re-reads itself, result unused
    ensure(rbn);
    if (rb + 1 < nb && (rb + 1) * kWave + lane < n)
      sym_next = mask[static_cast<size_t>((rb + 1) * kWave + lane) * nb + rb + 1];
    u64 dead = 0;
#pragma unroll
    for (int q = 0; q < RPI; ++q) dead |= readlane64(removed, rb + q * CW);   // rb < nb <= CW
    if (nrow < kWave) dead |= ~0ull << nrow;
    // greedy choice inside the block = the unique fixed point of K = alive & {i : no kept j < i suppresses i};
    // the iteration settles rows of dependency depth <= t after t rounds (a handful for real boxes, 64 at worst)
    const u64 alive = ~dead;
    u64 kept = alive;
    for (int round = 0; round <= kWave; ++round) {
      const u64 next = __ballot((below & kept) == 0) & alive;
      if (next == kept) break;
      kept = next;
    }
    if (lane == 0) keptw[rb] = kept;
    const u64 mine = kept >> sub;                    // bit t*RPI: is row t*RPI + sub kept
    const u64* pn = mask + (static_cast<size_t>(rbn) * kWave + sub) * nb + ccol;
#pragma unroll
    for (int t = 0; t < NIT; ++t) {
      removed |= v[t] & (0ull - ((mine >> (t * RPI)) & 1ull));
      v[t] = pn[static_cast<size_t>(t * RPI) * nb];
    }
  }
  return !timed_out;
}

// WIDE = false: the caller guarantees nb <= 32 (the one-row-per-load form and its 128 prefetch registers are left out)
template <bool WIDE>
__device__ __forceinline__ bool scan_chain_any(const u64* __restrict__ mask, int n, int nb, const int* done, u64* keptw,
                                               int spin_budget = kSpinBudget) {
  if (nb <= 16) return scan_chain<4>(mask, n, nb, done, keptw, spin_budget);
  if (nb <= 32 || !WIDE) return scan_chain<2>(mask, n, nb, done, keptw, spin_budget);
  return scan_chain<1>(mask, n, nb, done, keptw, spin_budget);
}

// kept sorted positions -> ascending ORIGINAL indices (at::nonzero(suppressed == 0), nms_cpu.cpp:64), the dense 0/1
// mask and the count.  Whole workgroup (kScanThreads); `flags` zeroed by the caller, `keptw` complete, barrier passed.
__device__ __forceinline__ void compact_keep(const u64* keptw, u64* flags, int* wsum, const int32_t* __restrict__ order,
                                             SegView sv, int s, int64_t* __restrict__ keep,
                                             int32_t* __restrict__ num_keep, uint8_t* __restrict__ keep_mask) {
  const int n = sv.n, nb = (n + kWave - 1) / kWave;
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  constexpr int kWaves = kScanThreads / kWave;
  // kept sorted positions -> bitset over original (segment-local) indices
  for (int p = tid; p < n; p += kScanThreads) {
    if ((keptw[p >> 6] >> (p & 63)) & 1ull) {
      const int o = order[p];
      atomicOr(&flags[o >> 6], 1ull << (o & 63));
    }
  }
  __syncthreads();
  if (keep_mask) {  // dense 0/1 form (original index space) for fixed-shape, sync-free callers
    uint8_t* km = keep_mask + sv.begin;
    for (int p = tid; p < n; p += kScanThreads) km[p] = static_cast<uint8_t>((flags[p >> 6] >> (p & 63)) & 1ull);
  }
  // workgroup exclusive prefix sum over popcounts of the flag words
  const int wpt = (nb + kScanThreads - 1) / kScanThreads;  // words per thread (<= 2)
  const int w0 = tid * wpt;
  int local = 0;
  for (int j = 0; j < wpt; ++j)
    if (w0 + j < nb) local += __popcll(flags[w0 + j]);
  int incl = local;
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const int t = __shfl_up(incl, off);
    if (lane >= off) incl += t;
  }
  if (lane == kWave - 1) wsum[wave] = incl;
  __syncthreads();
  int base = 0, total = 0;
  for (int j = 0; j < kWaves; ++j) {
    const int v = wsum[j];
    if (j < wave) base += v;
    total += v;
  }
  int pos = base + incl - local;
  int64_t* kout = keep + sv.begin;
  for (int j = 0; keep && j < wpt; ++j) {
    if (w0 + j >= nb) break;
    u64 f = flags[w0 + j];
    while (f) {
      const int b = __builtin_ctzll(f);
      f &= f - 1;
      kout[pos++] = static_cast<int64_t>((w0 + j) * 64 + b);
    }
  }
  if (tid == 0) num_keep[s] = total;
}

__global__ void __launch_bounds__(kScanThreads)
nms_scan_kernel(const int32_t* __restrict__ seg_offsets, int n_single, Work w,
                int64_t* __restrict__ keep, int32_t* __restrict__ num_keep,
                uint8_t* __restrict__ keep_mask) {
  __shared__ u64 remv[kMaxWords];    // pending "removed" bits per column block (sorted positions)
  __shared__ u64 keptw[kMaxWords];   // kept bits per row block (sorted positions)
  __shared__ u64 flags[kMaxWords];   // kept bits in ORIGINAL index space
  __shared__ int wsum[kScanThreads / kWave];
  __shared__ u64 s_kept;

  const int s = blockIdx.x;
  const SegView sv = seg_view(seg_offsets, n_single, s);
  const int n = sv.n;
  const int nb = (n + kWave - 1) / kWave;
  const u64* mask = w.mask + static_cast<size_t>(s) * w.mrows * w.nbmax;
  const int32_t* order = w.order + static_cast<size_t>(s) * w.stride;
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  constexpr int kWaves = kScanThreads / kWave;

  for (int i = tid; i < nb; i += kScanThreads) { remv[i] = 0; flags[i] = 0; }
  __syncthreads();

  // n <= 4096 (nb <= 64 column words): ONE wave walks the whole chain (scan_chain above), no LDS, no barrier.  The
  // shared-memory form below costs two workgroup barriers and a dependent global round trip per row block.
  const bool one_wave = nb <= kWave;
  if (one_wave) {
    if (wave == 0 && nb > 0) scan_chain_any<true>(mask, n, nb, nullptr, keptw);
    __syncthreads();
  }

  // wave 0 prefetches the next diagonal tile while the other waves' pushes are in flight
  u64 diag_next = 0;
  if (!one_wave && wave == 0 && lane < n) diag_next = mask[static_cast<size_t>(lane) * nb];
  for (int rb = 0; !one_wave && rb < nb; ++rb) {
    const int nrow = min(kWave, n - rb * kWave);
    if (wave == 0) {
      // diagonal tile: lane i holds the bits (> i) that row i suppresses inside this block (the stored word also
      // carries the symmetric bits below the diagonal)
      const u64 diag = diag_next & ~(((1ull << lane) - 1) | (1ull << lane));
      diag_next = 0;
      if (rb + 1 < nb && (rb + 1) * kWave + lane < n)
        diag_next = mask[static_cast<size_t>((rb + 1) * kWave + lane) * nb + rb + 1];
      u64 dead = uniform64(remv[rb]);
      if (nrow < kWave) dead |= ~0ull << nrow;
      u64 kept = 0;
      u64 alive = ~dead;
      while (alive) {  // uniform: one iteration per KEPT row
        const int i = __builtin_ctzll(alive);
        kept |= 1ull << i;
        const u64 d = readlane64(diag, i);
        alive &= ~(d | (1ull << i));
      }
      if (lane == 0) { s_kept = kept; keptw[rb] = kept; }
    }
    __syncthreads();
    const u64 kept = uniform64(s_kept);
    // push: removed[c] |= OR over kept rows i of mask[rb*64+i][c], for c > rb.
    // Work item = (kept row, 64-column group); waves take items round-robin.
    const int ncolw = nb - (rb + 1);
    if (ncolw > 0 && kept) {
      const int groups = (ncolw + kWave - 1) / kWave;
      for (int g = 0; g < groups; ++g) {
        const int c = rb + 1 + g * kWave + lane;
        u64 acc = 0;
        // this wave's share of the kept rows: ranks wave, wave+kWaves, ...
        u64 rest = kept;
        int rank = 0;
        u64 v0 = 0, v1 = 0, v2 = 0, v3 = 0;
        int pend = 0;
        while (rest) {
          const int i = __builtin_ctzll(rest);
          rest &= rest - 1;
          if ((rank++ % kWaves) != wave) continue;
          u64 v = 0;
          if (c < nb) v = mask[static_cast<size_t>(rb * kWave + i) * nb + c];
          // keep up to 4 loads in flight before consuming
          if (pend == 0) v0 = v; else if (pend == 1) v1 = v; else if (pend == 2) v2 = v; else v3 = v;
          if (++pend == 4) { acc |= v0 | v1 | v2 | v3; pend = 0; v0 = v1 = v2 = v3 = 0; }
        }
        acc |= v0 | v1 | v2 | v3;
        if (c < nb && acc) atomicOr(&remv[c], acc);
      }
    }
    __syncthreads();
  }

  compact_keep(keptw, flags, wsum, order, sv, s, keep, num_keep, keep_mask);
}

// ---------------------------------------------------------------------------- fused single launch (n <= 4096)
constexpr int kFusedMaxN = 4096;
constexpr int kFusedWaves = kScanThreads / kWave;   // tiles per tile workgroup

struct FusedCtrl {        // one per segment, in the caller's workspace (arbitrary previous content)
  u64 token;              // == this launch's token once the segment's sorted boxes / areas / order are in memory
  int32_t done[kWave];    // finished mask tiles per row block; zeroed by the sort workgroup BEFORE the token
  int32_t error;          // != 0: a wait of this launch ran out of its polling budget (zeroed by the sort workgroup
                          // BEFORE the token, raised by tile workgroups, read by the scan workgroup after its chain)
  int32_t pad[15];
};

// blockIdx.x:  [0, S)                 sort + gather of segment s, then publish the token
//              S * G                  mask tiles: chunk * S + s -> 8 consecutive tiles of segment s's upper triangle, row
//                                     major (row block 0 of every segment first: the scans start while later rows
//                                     are still computed); G = ceil(nbmax (nbmax + 1) / 16)
//              S                      scan chain + compaction of segment s
// scan_first = 0: tiles, then scans — every wait is for a workgroup with a LOWER index (detops_common.h, "flags between
// workgroups"); the scans are dispatched last, i.e. once most tile workgroups have retired.  scan_first = 1 (the host
// passes it when 2S workgroups are a small part of what the device holds at once): the scan workgroups sit right
// behind the sorts and consume row blocks while the tiles are still being produced; they wait for later-indexed
// workgroups, which is safe because those only need the S sort workgroups (earlier, never waiting) and one free slot.
// The token is unique per launch (64 bits: random per process + call counter, never 0), so a stale or never-written
// control block cannot match; the scan workgroup clears it at the end, so a replay of the same launch (captured
// graph: same kernel arguments) starts from a word that does not match either.
template <bool WIDE>   // false: max_n <= 2048 (every call the detector makes): ~half the registers, two workgroups per CU
__global__ void __launch_bounds__(kScanThreads)
nms_fused_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                 const int32_t* __restrict__ seg_offsets, int n_single, int npad, float thr, Work w,
                 FusedCtrl* __restrict__ ctrl, u64 token, u64 publish_token, int spin_budget, int S, int G, int scan_first, int64_t* __restrict__ keep,
                 int32_t* __restrict__ num_keep, uint8_t* __restrict__ keep_mask) {
  DETOPS_DYNAMIC_LDS(unsigned char, smem_raw);
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  const int T = S * G;                                            // G tile workgroups per segment
  int bid = blockIdx.x;                                           // role index: sorts, tiles, scans
  const bool check_sorted = !(scan_first & 2);                    // bit 1: tuning nms_no_presorted (A/B)
  if ((scan_first & 1) && bid >= S) bid = bid < 2 * S ? bid + T : bid - S;
  if (bid < S) {                                                  // ---- sort
    const int s = bid;
    NMS_TL(s == 0 && tid == 0, 0);
    if (tid < kWave) store_u32_wt(&ctrl[s].done[tid], 0u);
    if (tid == kWave) store_u32_wt(&ctrl[s].error, 0u);
    const SegView sv = seg_view(seg_offsets, n_single, s);
    if (sv.n <= 4 * kScanThreads)     // every call of the detector (n <= 2048): keys in registers
      sort_and_gather_reg<true>(reinterpret_cast<u64*>(smem_raw), boxes, scores, sv, w, s, check_sorted);
    else
      sort_and_gather<true>(reinterpret_cast<u64*>(smem_raw), boxes, scores, sv, npad, w, s, check_sorted);
    DETOPS_VMCNT_WAIT(0);            // this wave's write-through stores (sorted rows, the zeroed counters) are in memory
    __syncthreads();
    NMS_TL(s == 0 && tid == 0, 1);
    if (tid == 0) flag_store_relaxed(&ctrl[s].token, publish_token);   // everything it guards is already in memory (== token, except under the fault-injection switch)
    return;
  }
  if (bid < S + T) {                                              // ---- mask tiles, one per wave
    // G workgroups per segment, each 8 consecutive tiles of the segment's upper triangle in row-major order (row block
    // 0 first); workgroup index = chunk * S + segment, so all segments advance together
    const int idx = bid - S;
    const int s = idx % S, chunk = idx / S;
    const int n = seg_view(seg_offsets, n_single, s).n;
    const int nb = (n + kWave - 1) / kWave;
    const int ntiles = nb * (nb + 1) / 2;
    int t = chunk * kFusedWaves + wave;                           // wave-uniform
    if (t >= ntiles) return;                                      // (wave 0 holds the smallest t: it leaves last)
    int rb = 0;
    while (t >= nb - rb) { t -= nb - rb; ++rb; }                  // row lengths nb, nb - 1, ..., 1
    const int cb = rb + t;
    int* s_ok = reinterpret_cast<int*>(smem_raw);
    const int tl_tile = chunk * kFusedWaves + wave;
    NMS_TL(s == 0 && lane == 0 && tl_tile < 2112, 128 + 3 * tl_tile);
    if (wave == 0) {
      int budget = spin_budget;
      bool seen = true;
      while (flag_peek(&ctrl[s].token) != token) {
        if (!spin_again(budget)) { seen = false; break; }
      }
      if (lane == 0) s_ok[0] = seen ? 1 : 0;
    }
    __syncthreads();                 // waves without a tile have left; the barrier counts the remaining ones
    if (!s_ok[0]) {
      // the segment's sorted rows never showed up within the polling budget: this tile is NOT computed and NOT counted,
      // so the scan workgroup cannot complete its chain either and reports the segment as failed (num_keep = -1)
      if (tid == 0) flag_add(&ctrl[s].error, 1);
      return;
    }
    NMS_TL(s == 0 && lane == 0 && tl_tile < 2112, 129 + 3 * tl_tile);
    if (scan_first & 4)              // timing ablation (tuning nms_debug = 1): everything but the IoU arithmetic — wrong results
      mask_tile<true, 0>(w.boxes + static_cast<size_t>(s) * w.stride, w.areas + static_cast<size_t>(s) * w.stride,
                         w.mask + static_cast<size_t>(s) * w.mrows * w.nbmax, n, nb, rb, cb, thr);
    else
      mask_tile<true>(w.boxes + static_cast<size_t>(s) * w.stride, w.areas + static_cast<size_t>(s) * w.stride,
                      w.mask + static_cast<size_t>(s) * w.mrows * w.nbmax, n, nb, rb, cb, thr);
    DETOPS_VMCNT_WAIT(0);            // the 64 write-through stores of this wave have reached memory
    if (lane == 0) flag_add(&ctrl[s].done[rb], 1);
    NMS_TL(s == 0 && lane == 0 && tl_tile < 2112, 130 + 3 * tl_tile);
    return;
  }
  const int s = bid - S - T;                                      // ---- scan
  const SegView sv = seg_view(seg_offsets, n_single, s);
  const int n = sv.n, nb = (n + kWave - 1) / kWave;
  u64* keptw = reinterpret_cast<u64*>(smem_raw);
  u64* flags = keptw + kWave;
  int* wsum = reinterpret_cast<int*>(flags + kWave);
  int* s_fail = wsum + kScanThreads / kWave;
  if (tid < kWave) flags[tid] = 0;
  NMS_TL(s == 0 && tid == 0, 2);
  const u64* seg_mask = w.mask + static_cast<size_t>(s) * w.mrows * w.nbmax;
  if (wave == 0) {
    int budget = spin_budget;
    bool ok = true;
    while (flag_peek(&ctrl[s].token) != token) {   // the counters below are valid from here on
      if (!spin_again(budget)) { ok = false; break; }
    }
    DETOPS_ACQUIRE_AGENT();
    NMS_TL(s == 0 && lane == 0, 3);
    if (ok && nb > 0) ok = scan_chain_any<WIDE>(seg_mask, n, nb, ctrl[s].done, keptw, spin_budget);
    if (lane == 0) s_fail[0] = ok ? 0 : 1;
  }
  __syncthreads();
  NMS_TL(s == 0 && tid == 0, 4);
  if (wave == 0) {
    bool ok = s_fail[0] == 0;
    if (ok && flag_peek(&ctrl[s].error) != 0) ok = false;
    // a wait ran out of its budget (a producer workgroup was starved for seconds — another process or stream holding
    // the CUs): publish NOTHING for this segment and say so (num_keep[s] = -1, all-zero keep mask) instead of a keep
    // set built from rows that were never written
    if (!ok) keptw[lane] = 0ull;
    DETOPS_WAVE_SYNC();          // every lane of this wave has read s_fail / the token
    if (lane == 0) s_fail[0] = ok ? 0 : 1;
    // every tile of the segment is counted in, so nobody reads the token any more: clear it.  A captured graph
    // replays this launch with the SAME token value — the cleared word is what makes the replay wait again.
    if (lane == 0) flag_store_relaxed(&ctrl[s].token, 0ull);
  }
  __syncthreads();
  compact_keep(keptw, flags, wsum, w.order + static_cast<size_t>(s) * w.stride, sv, s, keep, num_keep, keep_mask);
  if (tid == 0 && s_fail[0]) num_keep[s] = -1;     // (same thread that wrote the count in compact_keep)
  NMS_TL(s == 0 && tid == 0, 5);
}

// ---------------------------------------------------------------------------- repair of failed segments
// Launched right behind nms_fused_kernel on the same stream, S workgroups.  Workgroup s looks at num_keep[s]: anything
// but the failure marker (-1) -> it leaves at once (the normal case: ~2 us for the whole launch).  A failed segment is
// redone HERE by this one workgroup alone — sort, every mask tile (a wave per tile, 8 at a time), scan chain, compaction
// — with no wait for any other workgroup, so it cannot time out; slow (one CU), but the reference never drops a
// segment (csrc/cuda/nms.cu:70-131) and neither does this path.  The fused launch has retired by now (stream order): its
// late writers cannot race with this one.  `status` (optional): a sticky device word, incremented once per repaired
// segment — the trainer reads it at its logging interval.
__global__ void __launch_bounds__(kScanThreads)
nms_repair_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                  const int32_t* __restrict__ seg_offsets, int n_single, int npad, float thr, Work w,
                  int64_t* __restrict__ keep, int32_t* __restrict__ num_keep, uint8_t* __restrict__ keep_mask,
                  int32_t* __restrict__ status) {
  DETOPS_DYNAMIC_LDS(unsigned char, smem_raw);
  const int s = blockIdx.x;
  if (num_keep[s] != -1) return;                                  // uniform over the workgroup
  const int tid = threadIdx.x, wave = tid / kWave;
  u64* keys = reinterpret_cast<u64*>(smem_raw);
  u64* keptw = keys + npad;
  u64* flags = keptw + kWave;
  int* wsum = reinterpret_cast<int*>(flags + kWave);
  const SegView sv = seg_view(seg_offsets, n_single, s);
  const int n = sv.n, nb = (n + kWave - 1) / kWave;
  if (tid < kWave) { flags[tid] = 0; keptw[tid] = 0; }
  sort_and_gather<true>(keys, boxes, scores, sv, npad, w, s);     // write-through: the other waves read the rows below
  DETOPS_VMCNT_WAIT(0);
  __syncthreads();
  const int ntiles = nb * (nb + 1) / 2;
  u64* mask = w.mask + static_cast<size_t>(s) * w.mrows * w.nbmax;
  for (int t0 = wave; t0 < ntiles; t0 += kFusedWaves) {
    int t = t0, rb = 0;
    while (t >= nb - rb) { t -= nb - rb; ++rb; }
    mask_tile<true>(w.boxes + static_cast<size_t>(s) * w.stride, w.areas + static_cast<size_t>(s) * w.stride, mask, n, nb, rb,
                    rb + t, thr);
  }
  DETOPS_VMCNT_WAIT(0);
  __syncthreads();
  if (wave == 0) {
    DETOPS_ACQUIRE_AGENT();
    if (nb > 0) scan_chain_any<true>(mask, n, nb, nullptr, keptw);
  }
  __syncthreads();
  compact_keep(keptw, flags, wsum, w.order + static_cast<size_t>(s) * w.stride, sv, s, keep, num_keep, keep_mask);
  if (tid == 0 && status) atomicAdd(status, 1);
}

// ---------------------------------------------------------------------------- host side
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct Layout {
  size_t off_boxes, off_areas, off_order, off_mask, off_ctrl, off_keys, off_keys2, off_cub, total;
  size_t cub_bytes;
};

Layout make_layout(int S, int max_n, bool big) {
  Layout l{};
  const size_t rows = static_cast<size_t>(S) * max_n;
  const size_t nbmax = static_cast<size_t>((max_n + kWave - 1) / kWave);
  size_t o = 0;
  l.off_boxes = o; o = align_up(o + rows * sizeof(float4), 256);
  l.off_areas = o; o = align_up(o + rows * sizeof(float), 256);
  l.off_order = o; o = align_up(o + rows * sizeof(int32_t), 256);
  l.off_mask = o;  o = align_up(o + static_cast<size_t>(S) * nbmax * kWave * nbmax * sizeof(u64), 256);
  l.off_ctrl = o;  o = align_up(o + static_cast<size_t>(S) * sizeof(FusedCtrl), 256);
  if (big) {
    l.off_keys = o;  o = align_up(o + rows * sizeof(u64), 256);
    l.off_keys2 = o; o = align_up(o + rows * sizeof(u64), 256);
    size_t cub = detops_sort_u64_bytes(max_n);
    if (cub == 0) cub = rows * sizeof(u64) * 2 + (1u << 20);   // size query failed (e.g. no device visible): a safe upper bound
    l.cub_bytes = cub;
    l.off_cub = o; o = align_up(o + cub, 256);
  }
  l.total = o;
  return l;
}

int run_nms(const float* boxes, const float* scores, const int32_t* seg_offsets, int S, int max_n,
            float thr, int64_t* keep, int32_t* num_keep, uint8_t* keep_mask, void* ws,
            size_t ws_bytes, hipStream_t st, int32_t* status = nullptr) {
  const bool big = max_n > kSortLdsMax;
  const Layout l = make_layout(S, max_n, big);
  if (ws_bytes < l.total || !ws) return DETOPS_EWORKSPACE;
  unsigned char* base_ptr = static_cast<unsigned char*>(ws);
  unsigned char* base = base_ptr;
  Work w;
  w.boxes = reinterpret_cast<float4*>(base + l.off_boxes);
  w.areas = reinterpret_cast<float*>(base + l.off_areas);
  w.order = reinterpret_cast<int32_t*>(base + l.off_order);
  w.mask = reinterpret_cast<u64*>(base + l.off_mask);
  w.stride = max_n;
  w.nbmax = (max_n + kWave - 1) / kWave;
  w.mrows = w.nbmax * kWave;

  if (max_n <= kFusedMaxN && detops_tuning().nms_fused != 2) {
    int npad = 2;
    while (npad < max_n) npad <<= 1;
    static const u64 base = (static_cast<u64>(std::random_device{}()) << 32) ^ 0x9e3779b97f4a7c15ull;
    static std::atomic<u64> calls{0};
    const u64 token = (base + calls.fetch_add(1)) | 1ull;   // never 0: the scan workgroups clear the word when done
    const int G = (w.nbmax * (w.nbmax + 1) / 2 + kFusedWaves - 1) / kFusedWaves;   // tile workgroups per segment
    const size_t lds = std::max<size_t>(npad * sizeof(u64), 4 * kScanThreads * sizeof(u64));   // LDS sort keys / the register sort's exchange buffer
    auto kernel = w.nbmax <= 32 ? nms_fused_kernel<false> : nms_fused_kernel<true>;
    int scan_first = 0;
    static int resident[2] = {0, 0};   // workgroups the device holds at once, per kernel variant (first device seen)
    int& cap = resident[w.nbmax <= 32 ? 0 : 1];
    if (cap == 0) cap = detops_resident_workgroups(kernel, kScanThreads, 32 * 1024);   // -1: unknown / host emulation
    scan_first = (cap > 0 && 8 * S <= cap && detops_tuning().nms_fused != 3) ? 1 : 0;
    hipLaunchKernelGGL(kernel, dim3(2 * S + S * G), dim3(kScanThreads), lds, st, boxes, scores,
                       seg_offsets, max_n, npad, thr, w, reinterpret_cast<FusedCtrl*>(base_ptr + l.off_ctrl), token,
                       detops_tuning().nms_fault == 1 ? (token ^ 2ull) : token,   // fault injection (tests): the token never shows up
                       detops_tuning().nms_spin_budget > 0 ? detops_tuning().nms_spin_budget : kSpinBudget, S, G,
                       scan_first | (detops_tuning().nms_no_presorted ? 2 : 0) | ((detops_tuning().nms_debug & 1) ? 4 : 0) | ((detops_tuning().nms_debug & 4) ? 16 : 0), keep, num_keep, keep_mask);
    if (detops_tuning().nms_no_repair != 1)    // failed segments (num_keep = -1) are redone, each by one workgroup alone
      hipLaunchKernelGGL(nms_repair_kernel, dim3(S), dim3(kScanThreads), npad * sizeof(u64) + 2 * kWave * sizeof(u64) + 64,
                         st, boxes, scores, seg_offsets, max_n, npad, thr, w, keep, num_keep, keep_mask, status);
    return launch_status();
  }

  if (!big) {
    int npad = 2;
    while (npad < max_n) npad <<= 1;
    const int threads = max(kWave, min(1024, npad / 2));
    hipLaunchKernelGGL(nms_sort_kernel, dim3(S), dim3(threads), npad * sizeof(u64), st, boxes,
                       scores, seg_offsets, max_n, npad, w);
  } else {
    u64* keys = reinterpret_cast<u64*>(base + l.off_keys);
    u64* keys2 = reinterpret_cast<u64*>(base + l.off_keys2);
    hipLaunchKernelGGL(nms_make_keys_kernel, dim3((max_n + 255) / 256, S), dim3(256), 0, st, scores,
                       seg_offsets, max_n, max_n, keys);
    for (int sg = 0; sg < S; ++sg) {
      u64* k1 = keys + static_cast<size_t>(sg) * max_n;
      u64* k2 = keys2 + static_cast<size_t>(sg) * max_n;
      {
        const int rc = detops_sort_u64(base + l.off_cub, l.cub_bytes, k1, k2, max_n, st);
        if (rc) return rc;
      }
    }
    hipLaunchKernelGGL(nms_gather_kernel, dim3((max_n + 255) / 256, S), dim3(256), 0, st, boxes, keys2,
                       seg_offsets, max_n, w);
  }
  int rc = launch_status();
  if (rc) return rc;
  const int nbm = w.nbmax;
  hipLaunchKernelGGL(nms_mask_kernel, dim3(nbm, nbm, S), dim3(kWave), 0, st, seg_offsets, max_n, thr,
                     w);
  hipLaunchKernelGGL(nms_scan_kernel, dim3(S), dim3(kScanThreads), 0, st, seg_offsets, max_n, w, keep,
                     num_keep, keep_mask);
  return launch_status();
}

}  // namespace

DETOPS_API size_t detops_nms_workspace_bytes(int n) {
  if (n <= 0) return 256;
  return make_layout(1, n, n > kSortLdsMax).total;
}

DETOPS_API int detops_nms_f32(const float* boxes, const float* scores, int n, float threshold,
                              int64_t* keep, int32_t* num_keep, void* workspace,
                              size_t workspace_bytes, detops_stream_t stream) {
  if (n < 0 || !num_keep) return DETOPS_EINVAL;
  hipStream_t st = as_stream(stream);
  if (n == 0) return static_cast<int>(hipMemsetAsync(num_keep, 0, sizeof(int32_t), st));
  if (!boxes || !scores || !keep) return DETOPS_EINVAL;
  if (n > kScanMaxN) return DETOPS_EUNSUPPORTED;
  return run_nms(boxes, scores, nullptr, 1, n, threshold, keep, num_keep, nullptr, workspace,
                 workspace_bytes, st);
}

DETOPS_API size_t detops_nms_batched_workspace_bytes(int num_segments, int max_n) {
  if (num_segments <= 0 || max_n <= 0) return 256;
  return make_layout(num_segments, max_n, max_n > kSortLdsMax).total;
}

DETOPS_API int detops_nms_batched_f32(const float* boxes, const float* scores,
                                      const int32_t* seg_offsets, int num_segments, int max_n,
                                      float threshold, int64_t* keep, int32_t* num_keep,
                                      void* workspace, size_t workspace_bytes,
                                      detops_stream_t stream) {
  if (num_segments < 0 || max_n < 0) return DETOPS_EINVAL;
  if (num_segments == 0) return 0;
  if (!seg_offsets || !num_keep) return DETOPS_EINVAL;
  hipStream_t st = as_stream(stream);
  if (max_n == 0)
    return static_cast<int>(hipMemsetAsync(num_keep, 0, sizeof(int32_t) * num_segments, st));
  if (!boxes || !scores || !keep) return DETOPS_EINVAL;
  if (max_n > kScanMaxN) return DETOPS_EUNSUPPORTED;
  return run_nms(boxes, scores, seg_offsets, num_segments, max_n, threshold, keep, num_keep,
                 nullptr, workspace, workspace_bytes, st);
}

DETOPS_API int detops_nms_batched_mask_f32(const float* boxes, const float* scores,
                                           const int32_t* seg_offsets, int num_segments, int max_n,
                                           float threshold, uint8_t* keep_mask, int32_t* num_keep,
                                           void* workspace, size_t workspace_bytes,
                                           detops_stream_t stream) {
  if (num_segments < 0 || max_n < 0) return DETOPS_EINVAL;
  if (num_segments == 0) return 0;
  if (!seg_offsets || !num_keep) return DETOPS_EINVAL;
  hipStream_t st = as_stream(stream);
  if (max_n == 0)
    return static_cast<int>(hipMemsetAsync(num_keep, 0, sizeof(int32_t) * num_segments, st));
  if (!boxes || !scores || !keep_mask) return DETOPS_EINVAL;
  if (max_n > kScanMaxN) return DETOPS_EUNSUPPORTED;
  return run_nms(boxes, scores, seg_offsets, num_segments, max_n, threshold, nullptr, num_keep,
                 keep_mask, workspace, workspace_bytes, st);
}


/* The segmented problem with BOTH result forms optional (keep and / or keep_mask) and a sticky status word: *status is
 * incremented once for every segment the single launch reported as failed and the repair launch redid (include/detops.h). */
DETOPS_API int detops_nms_batched_status_f32(const float* boxes, const float* scores, const int32_t* seg_offsets,
                                             int num_segments, int max_n, float threshold, int64_t* keep,
                                             uint8_t* keep_mask, int32_t* num_keep, int32_t* status, void* workspace,
                                             size_t workspace_bytes, detops_stream_t stream) {
  if (num_segments < 0 || max_n < 0) return DETOPS_EINVAL;
  if (num_segments == 0) return 0;
  if (!seg_offsets || !num_keep) return DETOPS_EINVAL;
  hipStream_t st = as_stream(stream);
  if (max_n == 0)
    return static_cast<int>(hipMemsetAsync(num_keep, 0, sizeof(int32_t) * num_segments, st));
  if (!boxes || !scores || (!keep && !keep_mask)) return DETOPS_EINVAL;
  if (max_n > kScanMaxN) return DETOPS_EUNSUPPORTED;
  return run_nms(boxes, scores, seg_offsets, num_segments, max_n, threshold, keep, num_keep, keep_mask, workspace,
                 workspace_bytes, st, status);
}

// diagnosis: the timeline words of the last fused launch that ran with tuning nms_debug & 4 (see g_nms_timeline)
DETOPS_API int detops_debug_nms_timeline(int64_t* host_out, int n) {
  if (!host_out || n < 0 || n > kNmsTimelineWords) return DETOPS_EINVAL;
#ifdef DETOPS_CPU_EMU
  for (int i = 0; i < n; ++i) host_out[i] = g_nms_timeline[i];
#else
  DETOPS_HIP_TRY(hipDeviceSynchronize());
  DETOPS_HIP_TRY(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_nms_timeline), sizeof(long long) * static_cast<size_t>(n)));
#endif
  return 0;
}
