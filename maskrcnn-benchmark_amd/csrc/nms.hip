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
#ifndef DETOPS_CPU_EMU
#include <hipcub/hipcub.hpp>
#else
#include <algorithm>
#endif

#include "detops_common.h"

namespace {

using u64 = unsigned long long;

constexpr int kSortLdsMax = 8192;   // bitonic-in-LDS capacity (64 KiB of keys)
constexpr int kScanMaxN = 65536;    // removed-words / bitset capacity of the scan kernel
constexpr int kScanThreads = 512;
constexpr int kMaxWords = kScanMaxN / 64;  // 1024 column blocks

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
  u64* mask;      // [S * stride * nbmax]
  int stride, nbmax;
};

// ---------------------------------------------------------------------------- 1. sort + gather
__global__ void __launch_bounds__(1024)
nms_sort_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                const int32_t* __restrict__ seg_offsets, int n_single, int npad, Work w) {
  DETOPS_DYNAMIC_LDS(unsigned char, smem_raw);
  u64* keys = reinterpret_cast<u64*>(smem_raw);
  const int s = blockIdx.x;
  const SegView sv = seg_view(seg_offsets, n_single, s);
  const int n = sv.n;
  // smallest power of two >= n (uniform), bounded by npad (the host-side capacity)
  int np = 2;
  while (np < n) np <<= 1;
  if (np > npad) np = npad;  // cannot happen when the caller honoured max_n
  for (int i = threadIdx.x; i < np; i += blockDim.x)
    keys[i] = (i < n) ? make_key(scores[sv.begin + i], static_cast<unsigned>(i)) : ~0ull;
  __syncthreads();
  for (int k = 2; k <= np; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = threadIdx.x; t < (np >> 1); t += blockDim.x) {
        // t-th compare-exchange pair of this stage: i has bit j clear, partner = i | j
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const int p = i | j;
        const u64 a = keys[i], b = keys[p];
        const bool up = (i & k) == 0;
        if ((a > b) == up) { keys[i] = b; keys[p] = a; }
      }
      __syncthreads();
    }
  }
  float4* ob = w.boxes + static_cast<size_t>(s) * w.stride;
  float* oa = w.areas + static_cast<size_t>(s) * w.stride;
  int32_t* oo = w.order + static_cast<size_t>(s) * w.stride;
  const float4* ib = reinterpret_cast<const float4*>(boxes) + sv.begin;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
#pragma clang fp contract(off)
    const int src = static_cast<int>(keys[i] & 0xffffffffu);
    const float4 b = ib[src];
    ob[i] = b;
    oa[i] = (b.z - b.x + 1.f) * (b.w - b.y + 1.f);  // nms_cpu.cpp:22
    oo[i] = src;
  }
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
__global__ void __launch_bounds__(kWave)
nms_mask_kernel(const int32_t* __restrict__ seg_offsets, int n_single, float thr, Work w) {
  const int cb = blockIdx.x, rb = blockIdx.y, s = blockIdx.z;
  if (cb < rb) return;  // upper triangle only
  const int n = seg_view(seg_offsets, n_single, s).n;
  if (rb * kWave >= n || cb * kWave >= n) return;
  const int nb = (n + kWave - 1) / kWave;  // words per row for THIS segment
  const float4* sb = w.boxes + static_cast<size_t>(s) * w.stride;
  const float* sa = w.areas + static_cast<size_t>(s) * w.stride;
  u64* mask = w.mask + static_cast<size_t>(s) * w.stride * w.nbmax;

  __shared__ float4 cbox[kWave];
  __shared__ float carea[kWave];
  const int lane = threadIdx.x;
  const int ncol = min(kWave, n - cb * kWave);
  if (lane < ncol) {
    cbox[lane] = sb[cb * kWave + lane];
    carea[lane] = sa[cb * kWave + lane];
  }
  __syncthreads();
  const int row = rb * kWave + lane;
  if (row >= n) return;
  const float4 a = sb[row];
  const float iarea = sa[row];
  u64 bits = 0;
  // Diagonal tile: the FULL symmetric relation of the row (every c != lane).  The comparison is symmetric bit for
  // bit (max / min / fp add commute, `iarea + carea - inter` sees the same two addends), so the bits below the
  // diagonal are "rows j < lane that suppress lane" — the transposed view the scan's fixed-point resolve needs.
  for (int c = 0; c < ncol; ++c) {
#pragma clang fp contract(off)
    if (rb == cb && c == lane) continue;
    const float4 b = cbox[c];
    const float xx1 = fmaxf(a.x, b.x), yy1 = fmaxf(a.y, b.y);
    const float xx2 = fminf(a.z, b.z), yy2 = fminf(a.w, b.w);
    const float ww = fmaxf(0.f, xx2 - xx1 + 1.f);
    const float hh = fmaxf(0.f, yy2 - yy1 + 1.f);
    const float inter = ww * hh;
    bool sup;
    if (inter > 0.f || thr <= 0.f) {
      const float ovr = inter / (iarea + carea[c] - inter);  // IEEE fp32 division
      sup = ovr >= thr;                                      // nms_cpu.cpp:60
    } else {
      sup = false;  // inter == 0 -> ovr is +-0 (or NaN for a zero union): never >= a positive thr
    }
    if (sup) bits |= 1ull << c;
  }
  mask[static_cast<size_t>(row) * nb + cb] = bits;
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
  const u64* mask = w.mask + static_cast<size_t>(s) * w.stride * w.nbmax;
  const int32_t* order = w.order + static_cast<size_t>(s) * w.stride;
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  constexpr int kWaves = kScanThreads / kWave;

  for (int i = tid; i < nb; i += kScanThreads) { remv[i] = 0; flags[i] = 0; }
  __syncthreads();

  // n <= 4096 (nb <= 64 column words): ONE wave walks the whole chain with the pending-removed words in registers,
  // lane c = column block c.  Per row block: resolve the diagonal tile (SALU chain over the kept rows), then OR the
  // kept rows' mask rows into the lanes — one coalesced load per kept row, all of a block's loads in flight
  // together, no LDS, no barrier.  The shared-memory form below costs two workgroup barriers and a dependent
  // global round trip per row block (32 blocks at n = 2000: 189 us for the 10 RPN segments of a training step).
  const bool one_wave = nb <= kWave;
  if (one_wave) {
    if (wave == 0) {
      u64 removed = 0;                                   // lane c: pending removed bits of column block c
      u64 sym_next = 0;                                  // lane i: symmetric diagonal-tile word of row rb*64 + i
      if (lane < n) sym_next = mask[static_cast<size_t>(lane) * nb];
      const u64 below_me = (1ull << lane) - 1;
      // A block's 64 mask rows, lane = column word.  Loads are unconditional (row / column clamped into the
      // segment): lanes <= rb or >= nb pick up words nobody reads again (their `removed` is never consulted), rows
      // >= nrow are never kept — no per-load predication.  Row u of the NEXT block is requested right after row u of
      // this block has been consumed, into the same registers: the loads fly during the OR pass and the next
      // resolve instead of stalling the chain once per block.
      const int ccol = min(lane, nb - 1);
      u64 v[kWave];
      {
        const int nrow0 = min(kWave, n);
#pragma unroll
        for (int u = 0; u < kWave; ++u) v[u] = mask[static_cast<size_t>(min(u, nrow0 - 1)) * nb + ccol];
      }
      for (int rb = 0; rb < nb; ++rb) {
        const int nrow = min(kWave, n - rb * kWave);
        const u64 below = sym_next & below_me;           // rows j < lane of this block that suppress row lane
        sym_next = 0;
        if (rb + 1 < nb && (rb + 1) * kWave + lane < n)
          sym_next = mask[static_cast<size_t>((rb + 1) * kWave + lane) * nb + rb + 1];
        u64 dead = readlane64(removed, rb);
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
        const int rbn = min(rb + 1, nb - 1);             // last block: re-reads itself, result unused
        const int nrown = min(kWave, n - rbn * kWave);
        const u64* blkn = mask + static_cast<size_t>(rbn) * kWave * nb;   // uniform
#pragma unroll
        for (int u = 0; u < kWave; ++u) {
          if ((kept >> u) & 1ull) removed |= v[u];       // uniform condition
          v[u] = blkn[static_cast<size_t>(min(u, nrown - 1)) * nb + ccol];
        }
      }
    }
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

// ---------------------------------------------------------------------------- host side
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct Layout {
  size_t off_boxes, off_areas, off_order, off_mask, off_keys, off_keys2, off_cub, total;
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
  l.off_mask = o;  o = align_up(o + rows * nbmax * sizeof(u64), 256);
  if (big) {
    l.off_keys = o;  o = align_up(o + rows * sizeof(u64), 256);
    l.off_keys2 = o; o = align_up(o + rows * sizeof(u64), 256);
    size_t cub = 0;
#ifndef DETOPS_CPU_EMU
    const hipError_t qe = hipcub::DeviceRadixSort::SortKeys(
        nullptr, cub, static_cast<u64*>(nullptr), static_cast<u64*>(nullptr), max_n);
    // size query failed (e.g. no device visible): use a safe upper bound
    if (qe != hipSuccess || cub == 0) cub = rows * sizeof(u64) * 2 + (1u << 20);
#else
    cub = 256;
#endif
    l.cub_bytes = cub;
    l.off_cub = o; o = align_up(o + cub, 256);
  }
  l.total = o;
  return l;
}

int run_nms(const float* boxes, const float* scores, const int32_t* seg_offsets, int S, int max_n,
            float thr, int64_t* keep, int32_t* num_keep, uint8_t* keep_mask, void* ws,
            size_t ws_bytes, hipStream_t st) {
  const bool big = max_n > kSortLdsMax;
  const Layout l = make_layout(S, max_n, big);
  if (ws_bytes < l.total || !ws) return DETOPS_EWORKSPACE;
  unsigned char* base = static_cast<unsigned char*>(ws);
  Work w;
  w.boxes = reinterpret_cast<float4*>(base + l.off_boxes);
  w.areas = reinterpret_cast<float*>(base + l.off_areas);
  w.order = reinterpret_cast<int32_t*>(base + l.off_order);
  w.mask = reinterpret_cast<u64*>(base + l.off_mask);
  w.stride = max_n;
  w.nbmax = (max_n + kWave - 1) / kWave;

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
#ifndef DETOPS_CPU_EMU
      size_t cub = l.cub_bytes;
      DETOPS_HIP_TRY(hipcub::DeviceRadixSort::SortKeys(base + l.off_cub, cub, k1, k2, max_n, 0, 64, st));
#else
      std::copy(k1, k1 + max_n, k2);
      std::sort(k2, k2 + max_n);
#endif
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
