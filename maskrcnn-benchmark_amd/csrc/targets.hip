// targets.hip — training-target assignment on the device, gfx950.
//
//   detops_match_boxes_f32      IoU(gt, boxes) + Matcher, fused: the [M, K] quality matrix of the reference
//                               (structures/boxlist_ops.py:53-89 -> modeling/matcher.py:42-112) is never
//                               written — 268,569 RPN anchors x M ground-truth boxes per image are evaluated
//                               in registers, twice when the low-quality rule is on (per-gt maximum first).
//   detops_sample_labels        BalancedPositiveNegativeSampler (modeling/balanced_positive_negative_sampler.py
//                               :19-68) without sorting the anchors: candidates whose hashed key falls under a
//                               threshold sized for ~8x the quota survive a single filtering pass, one
//                               workgroup per image sorts the few thousand survivors and keeps the k smallest.
//   detops_mask_targets         crop + bilinear resize of the matched ground-truth mask to M x M per positive
//                               ROI (roi_heads/mask_head/loss.py:11-42, structures/segmentation_mask.py:118-158)
//                               in the operation order of the CPU kernel the reference runs it on.
//
// Everything is stream-ordered and sync-free; decisions (IoU thresholds, equality with the per-gt maximum,
// integer-mask truncation) are evaluated with FP contraction off and IEEE division so that they are the
// decisions the reference's CPU arithmetic takes.
#include <algorithm>

#include <cmath>

#include "detops_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kGtChunk = 256;   // ground-truth boxes staged in LDS per pass

// IoU with the reference's "+1" convention and evaluation order (boxlist_ops.py:72-88):
//   area = (x2 - x1 + 1) * (y2 - y1 + 1); wh = clamp(rb - lt + 1, 0); inter / (area_a + area_b - inter)
__device__ __forceinline__ float iou_ref(const float4 g, float area_g, const float4 b, float area_b) {
#pragma clang fp contract(off)
  const float ltx = fmaxf(g.x, b.x), lty = fmaxf(g.y, b.y);
  const float rbx = fminf(g.z, b.z), rby = fminf(g.w, b.w);
  const float w = fmaxf(rbx - ltx + 1.f, 0.f), h = fmaxf(rby - lty + 1.f, 0.f);
  const float inter = w * h;
  return inter / (area_g + area_b - inter);
}

__device__ __forceinline__ float box_area(const float4 b) {
#pragma clang fp contract(off)
  return (b.z - b.x + 1.f) * (b.w - b.y + 1.f);
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) v = fmaxf(v, __shfl_down(v, off));
  return v;
}

// PASS = 0: per box the best ground truth (first index among ties, like torch.max on the CPU) and its IoU;
//           with `best_gt` != NULL also the per-gt maximum over the boxes: a wave that overlaps a gt at all reduces its
//           lanes' best overlap (four boxes per lane) and raises the gt's word in LDS (max on the non-negative float's
//           bit pattern); the workgroup publishes each raised word with ONE filtered global atomicMax.
// PASS = 1: the final Matcher output, low-quality rule included (needs the completed best_gt of pass 0).
// grid = (ceil(K / (256 * kMatchPer)), N); gt [N, M, 4], valid [N, M], boxes [N or 1, K, 4].
constexpr int kMatchPer = 4;   // boxes per lane: one LDS read of a gt row serves four IoUs

template <int PASS>
__global__ void __launch_bounds__(kBlock)
match_kernel(const float* __restrict__ gt, const uint8_t* __restrict__ valid, const float* __restrict__ boxes,
             int boxes_batched, int M, int K, float high, float low, int allow_lq,
             int32_t* __restrict__ best_gt /* [N, M] float bits */, int64_t* __restrict__ matched /* [N, K] */) {
  __shared__ float4 s_gt[kGtChunk];
  __shared__ float s_area[kGtChunk];
  __shared__ int s_ok[kGtChunk];
  __shared__ int s_best[kGtChunk];
  const int n = blockIdx.y;
  const int k0 = blockIdx.x * (kBlock * kMatchPer) + threadIdx.x;
  const bool track = PASS == 0 && best_gt != nullptr;
  float4 b[kMatchPer];
  float area_b[kMatchPer], best[kMatchPer];
  int arg[kMatchPer];
  bool live[kMatchPer], lq[kMatchPer];
#pragma unroll
  for (int j = 0; j < kMatchPer; ++j) {
    const int k = k0 + j * kBlock;
    live[j] = k < K;
    b[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live[j]) b[j] = reinterpret_cast<const float4*>(boxes)[(boxes_batched ? static_cast<size_t>(n) * K : 0) + k];
    area_b[j] = box_area(b[j]);
    best[j] = -2.f;   // below every quality (invalid rows hold -1)
    arg[j] = 0;
    lq[j] = false;
  }
  for (int m0 = 0; m0 < M; m0 += kGtChunk) {
    const int mc = min(kGtChunk, M - m0);
    __syncthreads();
    for (int t = threadIdx.x; t < mc; t += kBlock) {
      const float4 g = reinterpret_cast<const float4*>(gt)[static_cast<size_t>(n) * M + m0 + t];
      s_gt[t] = g;
      s_area[t] = box_area(g);
      s_ok[t] = valid[static_cast<size_t>(n) * M + m0 + t] != 0;
      s_best[t] = (PASS == 1 && allow_lq) ? best_gt[static_cast<size_t>(n) * M + m0 + t] : 0;
    }
    __syncthreads();
    for (int t = 0; t < mc; ++t) {
      const float4 g = s_gt[t];
      const float ag = s_area[t];
      const bool ok = s_ok[t] != 0;
      const int sb = (PASS == 1) ? s_best[t] : 0;
      float qmax = 0.f;                                        // this lane's best overlap with gt t over its boxes
#pragma unroll
      for (int j = 0; j < kMatchPer; ++j) {
        // (a wave-uniform shortcut around the IEEE division for pairs that do not overlap was measured: 20.6 -> 23.7 us)
        const float q = ok ? iou_ref(g, ag, b[j], area_b[j]) : -1.f;
        if (q > best[j]) { best[j] = q; arg[j] = m0 + t; }   // strict: the first maximum wins
        if (track && live[j]) qmax = fmaxf(qmax, q);
        if (PASS == 1 && allow_lq) lq[j] = lq[j] || (ok && __float_as_int(q) == sb && q >= 0.f);
      }
      // most (wave, gt) pairs do not overlap at all (a wave holds 256 neighbouring anchors): those cost one ballot; the
      // others one wave reduction and ONE LDS max by lane 0 (64 lanes raising the same LDS word would serialise)
      if (track && __ballot(qmax > 0.f) != 0ull) {
        const float wm = wave_max(qmax);
        if ((threadIdx.x & (kWave - 1)) == 0) atomicMax(&s_best[t], __float_as_int(wm));
      }
    }
    if (track) {
      __syncthreads();
      for (int t = threadIdx.x; t < mc; t += kBlock) {
        const int v = s_best[t];
        int32_t* slot = &best_gt[static_cast<size_t>(n) * M + m0 + t];
        if (v > 0 && v > *slot) atomicMax(slot, v);   // (a stale read only costs a redundant atomic)
      }
    }
  }
  if (PASS == 0 && allow_lq) return;   // pass 1 writes the result
#pragma unroll
  for (int j = 0; j < kMatchPer; ++j) {
    if (!live[j]) continue;
    int64_t out = arg[j];
    if (best[j] < low) out = -1;                       // Matcher.BELOW_LOW_THRESHOLD
    else if (best[j] < high) out = -2;                 // Matcher.BETWEEN_THRESHOLDS
    if (lq[j]) out = arg[j];                           // low-quality matches keep their arg-max ground truth
    matched[static_cast<size_t>(n) * K + k0 + j * kBlock] = out;
  }
}

// ------------------------------------------------------------------------------------------ sampler
// 32-bit key of element (row, i) under `seed`: two rounds of a multiply-xorshift mixer (the finaliser of
// MurmurHash3 — public domain), enough for an unbiased-looking random subset.
__device__ __forceinline__ unsigned sample_key(unsigned long long seed, int row, int i) {
  unsigned h = static_cast<unsigned>(seed) ^ (static_cast<unsigned>(row) * 0x9E3779B1u);
  h ^= static_cast<unsigned>(i) * 0x85EBCA6Bu + static_cast<unsigned>(seed >> 32);
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  h += static_cast<unsigned>(i);
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}

// label class of an element: 1 positive (label >= 1), 0 negative (label == 0), -1 ignored
template <typename T>
__device__ __forceinline__ int label_class(T v) { return v >= static_cast<T>(1) ? 1 : (v == static_cast<T>(0) ? 0 : -1); }

// per (row, block): candidates of each class -> partial[row][class][block] (no atomics, no zeroed counters); the
// filter and finish kernels add a row's partials in block order.  Block (0, row) also clears the row's survivor counters.
template <typename T>
__global__ void __launch_bounds__(kBlock)
sampler_count_kernel(const T* __restrict__ labels, int n, int32_t* __restrict__ partial /* [N, 2, gridDim.x] */,
                     int32_t* __restrict__ nsurv /* [N, 2] */) {
  const int row = blockIdx.y;
  int c0 = 0, c1 = 0;
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const int c = label_class(labels[static_cast<size_t>(row) * n + i]);
    c0 += c == 0;
    c1 += c == 1;
  }
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) { c0 += __shfl_down(c0, off); c1 += __shfl_down(c1, off); }
  __shared__ int s_c[2][kBlock / kWave];
  if ((threadIdx.x & (kWave - 1)) == 0) { s_c[0][threadIdx.x / kWave] = c0; s_c[1][threadIdx.x / kWave] = c1; }
  __syncthreads();
  if (threadIdx.x < 2) {
    int t = 0;
    for (int w = 0; w < kBlock / kWave; ++w) t += s_c[threadIdx.x][w];
    partial[(static_cast<size_t>(row) * 2 + threadIdx.x) * gridDim.x + blockIdx.x] = t;
    if (blockIdx.x == 0) nsurv[row * 2 + threadIdx.x] = 0;
  }
}

// a row's class totals from the count kernel's partials (whole workgroup; result broadcast through LDS)
__device__ __forceinline__ void sampler_totals(const int32_t* __restrict__ partial, int row, int nblocks, int* s_tot, int& n_neg, int& n_pos) {
  if (threadIdx.x < 2 * kWave) {            // wave 0: negatives, wave 1: positives
    const int cls = threadIdx.x / kWave, lane = threadIdx.x & (kWave - 1);
    int t = 0;
    for (int i = lane; i < nblocks; i += kWave) t += partial[(static_cast<size_t>(row) * 2 + cls) * nblocks + i];
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) t += __shfl_down(t, off);
    if (lane == 0) s_tot[cls] = t;
  }
  __syncthreads();
  n_neg = s_tot[0]; n_pos = s_tot[1];
}

struct SamplerPlan {
  int n, B, max_pos, cap;   // cap = survivor capacity per (row, class)
  int count_blocks;         // grid.x of the count kernel (= partial sums per row and class)
  unsigned long long seed;
  const unsigned long long* seed_dev;   // optional device word mixed into the seed at run time (a captured HIP graph replays its
};                                      // launch arguments: the word is what changes from one replay to the next)

// key threshold of a class: survivors ~ Binomial(cand, f) with mean mu = quota + 8 sqrt(quota) + 32, i.e. more
// than 8 standard deviations above the quota (P[fewer than quota] < 1e-14) and far below cap = 16 x B; all
// candidates when there are at most 2 mu of them.  A short survivor list keeps the per-image sort small.
__device__ __forceinline__ unsigned class_threshold(int cand, int quota) {
  if (quota <= 0) return 0u;
  const double mu = quota + 8.0 * sqrt(static_cast<double>(quota)) + 32.0;
  if (cand <= 2.0 * mu) return 0xffffffffu;
  return static_cast<unsigned>(mu / static_cast<double>(cand) * 4294967295.0);
}

template <typename T>
__global__ void __launch_bounds__(kBlock)
sampler_filter_kernel(const T* __restrict__ labels, SamplerPlan P, const int32_t* __restrict__ partial,
                      int32_t* __restrict__ nsurv /* [N, 2] */, unsigned long long* __restrict__ surv /* [N, 2, cap] */,
                      uint8_t* __restrict__ pos_mask, uint8_t* __restrict__ neg_mask) {
  const int row = blockIdx.y;
  __shared__ int s_tot[2];
  int n_neg, n_pos;
  sampler_totals(partial, row, P.count_blocks, s_tot, n_neg, n_pos);
  const int k_pos = min(n_pos, P.max_pos);
  const int k_neg = min(min(n_neg, P.B), P.B - k_pos);
  const unsigned thr_pos = class_threshold(n_pos, k_pos), thr_neg = class_threshold(n_neg, k_neg);
  const unsigned long long seed = P.seed_dev ? (P.seed ^ (*P.seed_dev * 0x9E3779B97F4A7C15ull)) : P.seed;
  const int lane = threadIdx.x & (kWave - 1);
  const int span = gridDim.x * kBlock;
  for (int i0 = blockIdx.x * kBlock; i0 < P.n; i0 += span) {   // uniform trip count: the ballots need every lane
    const int i = i0 + threadIdx.x;
    const int c = (i < P.n) ? label_class(labels[static_cast<size_t>(row) * P.n + i]) : -1;
    if (i < P.n) {   // the masks start all-zero (this pass touches every element anyway: no fill launches); the finish
      pos_mask[static_cast<size_t>(row) * P.n + i] = 0;   // kernel — the next launch — sets the chosen ones
      neg_mask[static_cast<size_t>(row) * P.n + i] = 0;
    }
    const unsigned key = sample_key(seed, row, i);
    const unsigned thr = c == 1 ? thr_pos : thr_neg;
    const bool keep = c >= 0 && thr != 0u && key <= thr;
#pragma unroll
    for (int cls = 0; cls < 2; ++cls) {   // one atomic per wave and class; lanes take consecutive slots
      const unsigned long long m = __ballot(keep && c == cls);
      if (m == 0ull) continue;
      int base = 0;
      if (lane == 0) base = atomicAdd(&nsurv[row * 2 + cls], __popcll(m));
      base = __shfl(base, 0);
      if (keep && c == cls) {
        const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
        if (slot < P.cap)
          surv[(static_cast<size_t>(row) * 2 + cls) * P.cap + slot] = (static_cast<unsigned long long>(key) << 32) | static_cast<unsigned>(i);
      }
    }
  }
}

// one workgroup per (row, class): sort the class's survivors by (key, index) in LDS, keep the quota
__global__ void __launch_bounds__(1024)
sampler_finish_kernel(SamplerPlan P, const int32_t* __restrict__ partial, const int32_t* __restrict__ nsurv,
                      const unsigned long long* __restrict__ surv, uint8_t* __restrict__ pos_mask,
                      uint8_t* __restrict__ neg_mask, int64_t* __restrict__ idx /* [N, B] nullable */,
                      uint8_t* __restrict__ idx_valid /* [N, B] nullable */) {
  DETOPS_DYNAMIC_LDS(unsigned long long, keys);
  __shared__ int s_tot[2];
  const int row = blockIdx.x;
  const int c = 1 - static_cast<int>(blockIdx.y);   // y = 0: positives
  int n_neg, n_pos;
  sampler_totals(partial, row, P.count_blocks, s_tot, n_neg, n_pos);
  const int k_pos = min(n_pos, P.max_pos);
  const int k_neg = min(min(n_neg, P.B), P.B - k_pos);
  const int ns = min(nsurv[row * 2 + c], P.cap);
  const int k = min(c ? k_pos : k_neg, ns);
  int npad = 2;
  while (npad < ns) npad <<= 1;   // the bitonic network only spans the survivors actually present
  for (int i = threadIdx.x; i < npad; i += blockDim.x)
    keys[i] = (i < ns) ? surv[(static_cast<size_t>(row) * 2 + c) * P.cap + i] : ~0ull;
  __syncthreads();
  for (int kk = 2; kk <= npad; kk <<= 1) {
    for (int j = kk >> 1; j > 0; j >>= 1) {
      for (int t = threadIdx.x; t < (npad >> 1); t += blockDim.x) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const int p = i | j;
        const unsigned long long a = keys[i], b = keys[p];
        const bool up = (i & kk) == 0;
        if ((a > b) == up) { keys[i] = b; keys[p] = a; }
      }
      __syncthreads();
    }
  }
  uint8_t* mask = c ? pos_mask : neg_mask;
  const int base = c ? 0 : k_pos;   // fixed-length list: positives first, then negatives
  for (int i = threadIdx.x; i < k; i += blockDim.x) {
    const int e = static_cast<int>(keys[i] & 0xffffffffull);
    mask[static_cast<size_t>(row) * P.n + e] = 1;
    if (idx) { idx[static_cast<size_t>(row) * P.B + base + i] = e; idx_valid[static_cast<size_t>(row) * P.B + base + i] = 1; }
  }
  // the slots this class could not fill (every slot of the list is written by this kernel: no fill launches)
  const int want = c ? k_pos : (P.B - k_pos);
  for (int i = k + threadIdx.x; idx && i < want; i += blockDim.x) {
    idx[static_cast<size_t>(row) * P.B + base + i] = 0;
    idx_valid[static_cast<size_t>(row) * P.B + base + i] = 0;
  }
}

// ------------------------------------------------------------------------------------------ mask targets
struct AxisTap { int i0, i1; float l; };

// torch's bilinear resize (align_corners = False) of a crop [lo, lo + size) to M samples, sample d:
// src = fma(size / M, d + 0.5, -0.5) clamped at 0 (ONE rounding, as ATen's CPU kernel evaluates it),
// taps (floor(src), +1) clamped to the crop.
__device__ __forceinline__ AxisTap axis_tap(int lo, int size, int M, int d) {
#pragma clang fp contract(off)
  const float scale = static_cast<float>(size) / static_cast<float>(M);
  float src = fmaf(scale, static_cast<float>(d) + 0.5f, -0.5f);
  src = fmaxf(src, 0.f);
  int i0 = static_cast<int>(floorf(src));
  i0 = min(i0, size - 1);
  const int i1 = min(i0 + 1, size - 1);
  AxisTap t;
  t.l = src - static_cast<float>(i0);
  t.i0 = i0 + lo;
  t.i1 = i1 + lo;
  return t;
}

// one workgroup per ROI; masks [G, H, W] of MT (uint8 = integer masks: the interpolated value is truncated,
// structures/segmentation_mask.py `.type_as(masks)`; float = kept), out [P, M, M] fp32
// TRUNC: 0 = keep the float value, 1 = integer masks (truncate), 2 = bool masks (non-zero -> 1)
template <typename MT, int TRUNC>
__global__ void __launch_bounds__(kBlock)
mask_targets_kernel(const MT* __restrict__ masks, const int64_t* __restrict__ mask_index, const float* __restrict__ boxes,
                    int G, int H, int W, int M, float* __restrict__ out) {
#pragma clang fp contract(off)
  const int p = blockIdx.x;
  const float* bx = boxes + static_cast<size_t>(p) * 4;
  // box.round() (half to even) -> clamp to the image -> at least one pixel
  const int bx0 = static_cast<int>(rintf(bx[0])), by0 = static_cast<int>(rintf(bx[1]));
  const int bx1 = static_cast<int>(rintf(bx[2])), by1 = static_cast<int>(rintf(bx[3]));
  const int xmin = min(max(bx0, 0), W - 1), ymin = min(max(by0, 0), H - 1);
  const int xmax = max(min(max(bx1, 0), W), xmin + 1), ymax = max(min(max(by1, 0), H), ymin + 1);
  int64_t g = mask_index[p];
  g = g < 0 ? 0 : (g >= G ? G - 1 : g);
  const MT* m = masks + static_cast<size_t>(g) * H * W;
  for (int e = threadIdx.x; e < M * M; e += kBlock) {
    const int dy = e / M, dx = e - dy * M;
    const AxisTap ty = axis_tap(ymin, ymax - ymin, M, dy);
    const AxisTap tx = axis_tap(xmin, xmax - xmin, M, dx);
    const float hy = 1.0f - ty.l, hx = 1.0f - tx.l;
    const float p00 = static_cast<float>(m[static_cast<size_t>(ty.i0) * W + tx.i0]);
    const float p01 = static_cast<float>(m[static_cast<size_t>(ty.i0) * W + tx.i1]);
    const float p10 = static_cast<float>(m[static_cast<size_t>(ty.i1) * W + tx.i0]);
    const float p11 = static_cast<float>(m[static_cast<size_t>(ty.i1) * W + tx.i1]);
    // ((h0*w0)*p00 + (h0*w1)*p01) + (h1*w0)*p10 + (h1*w1)*p11, every product and sum rounded to fp32
    float v = (hy * hx) * p00 + (hy * tx.l) * p01;
    v = v + (ty.l * hx) * p10;
    v = v + (ty.l * tx.l) * p11;
    if (TRUNC == 1) v = static_cast<float>(static_cast<MT>(v));
    if (TRUNC == 2) v = (v != 0.f) ? 1.f : 0.f;
    out[static_cast<size_t>(p) * M * M + e] = v;
  }
}

inline size_t up256(size_t v) { return (v + 255) & ~static_cast<size_t>(255); }

constexpr int kSamplerBlocks = 2 * kNumCU;   // count / filter grid.x at most

struct SamplerLayout { size_t off_counts, off_nsurv, off_surv, total; int cap; };

SamplerLayout sampler_layout(int N, int B) {
  SamplerLayout l{};
  // survivors are at most 2 mu = 2 (k + 8 sqrt(k) + 32) per class (class_threshold), k <= B: 16 B covers that for
  // B >= 8; tiny batch sizes (unit tests) need the explicit bound or the filter would drop survivors in arrival order
  l.cap = std::max(16 * std::max(B, 1), 2 * (std::max(B, 1) + 8 * static_cast<int>(std::ceil(std::sqrt(static_cast<double>(std::max(B, 1))))) + 32) + 8);
  size_t o = 0;
  l.off_counts = o; o = up256(o + sizeof(int32_t) * 2 * N * kSamplerBlocks);   // the count kernel's partial sums
  l.off_nsurv = o;  o = up256(o + sizeof(int32_t) * 2 * N);
  l.off_surv = o;   o = up256(o + sizeof(unsigned long long) * 2 * static_cast<size_t>(N) * l.cap);
  l.total = o;
  return l;
}

template <typename T>
int run_sampler(const void* labels, int N, int n, int B, int max_pos, unsigned long long seed, const unsigned long long* seed_dev,
                uint8_t* pos_mask, uint8_t* neg_mask, int64_t* idx, uint8_t* idx_valid, void* ws, hipStream_t st) {
  const SamplerLayout l = sampler_layout(N, B);
  unsigned char* base = static_cast<unsigned char*>(ws);
  int32_t* counts = reinterpret_cast<int32_t*>(base + l.off_counts);
  int32_t* nsurv = reinterpret_cast<int32_t*>(base + l.off_nsurv);
  unsigned long long* surv = reinterpret_cast<unsigned long long*>(base + l.off_surv);
  // three launches, nothing to clear: the count kernel writes per-block partial sums and zeroes the survivor counters,
  // masks and lists are written in full by the filter / finish kernels
  const int bx = std::max(1, static_cast<int>(std::min<int64_t>(ceil_div64(n, kBlock), kSamplerBlocks)));
  const SamplerPlan P{n, B, max_pos, l.cap, bx, seed, seed_dev};
  const dim3 grid(static_cast<unsigned>(bx), static_cast<unsigned>(N));
  const T* lab = static_cast<const T*>(labels);
  hipLaunchKernelGGL(sampler_count_kernel<T>, grid, dim3(kBlock), 0, st, lab, n, counts, nsurv);
  hipLaunchKernelGGL(sampler_filter_kernel<T>, grid, dim3(kBlock), 0, st, lab, P, counts, nsurv, surv, pos_mask, neg_mask);
  int npad = 2;
  while (npad < l.cap) npad <<= 1;
  hipLaunchKernelGGL(sampler_finish_kernel, dim3(static_cast<unsigned>(N), 2), dim3(1024), npad * sizeof(unsigned long long), st,
                     P, counts, nsurv, surv, pos_mask, neg_mask, idx, idx_valid);
  return launch_status();
}


// ---------------------------------------------------------------------------------------------------------
// RPN loss (reference modeling/rpn/loss.py:92-127): objectness = BCE-with-logits over the sampled anchors,
// box = smooth-L1 (beta 1/9) between the regression output and BoxCoder.encode(matched gt, anchor) over the
// sampled positives, both divided by the number of sampled anchors.  The reference builds both from ~120
// elementwise / indexing launches over [N, 268,569] anchors (permute + cat of the per-level head outputs,
// encode, masks, two reductions, and the autograd mirror of each).  Here ONE launch reads the head outputs in
// the layout the heads wrote them ([N, A, H, W] / [N, 4A, H, W] per level), evaluates both losses per anchor,
// and writes d(loss sum)/d(logit) in the same layout; a second launch finishes the two fixed-order sums.  The
// backward is one scaling launch.  Anchor t = level offset + (y * W + x) * A + a, i.e. the order of
// AnchorGenerator.grid_anchors and of concat_box_prediction_layers (modeling/rpn/utils.py:9-45).
// ---------------------------------------------------------------------------------------------------------
struct RpnLevels {
  const float* obj[DETOPS_MAX_LEVELS];
  const float* box[DETOPS_MAX_LEVELS];
  float* gobj[DETOPS_MAX_LEVELS];
  float* gbox[DETOPS_MAX_LEVELS];
  int plane[DETOPS_MAX_LEVELS];       // H * W
  int first[DETOPS_MAX_LEVELS + 1];   // first anchor index of the level; first[num] = T
  int num;
};

constexpr int kRpnMaxBlocks = 2048;

__global__ void __launch_bounds__(kBlock)
rpn_loss_kernel(RpnLevels L, int A, const float* __restrict__ anchors, const int64_t* __restrict__ matched,
                const uint8_t* __restrict__ pos, const uint8_t* __restrict__ neg, const float* __restrict__ gt,
                int N, int M, int T, float beta, float wx, float wy, float ww, float wh,
                float* __restrict__ partial /* [gridDim.x][3] */) {
#pragma clang fp contract(off)
  float s_obj = 0.f, s_box = 0.f, s_cnt = 0.f;
  const int64_t total = static_cast<int64_t>(N) * T;
  const float inv_beta = 1.f / beta, half_beta = 0.5f * beta;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < total;
       i += static_cast<int64_t>(gridDim.x) * kBlock) {
    const int n = static_cast<int>(i / T);
    const int t = static_cast<int>(i - static_cast<int64_t>(n) * T);
    int l = 0;
#pragma unroll
    for (int k = 1; k < DETOPS_MAX_LEVELS; ++k)
      if (k < L.num && t >= L.first[k]) l = k;
    const float* obj = L.obj[0]; const float* box = L.box[0]; float* gobj = L.gobj[0]; float* gbox = L.gbox[0];
    int plane = L.plane[0], first = L.first[0];
#pragma unroll
    for (int k = 1; k < DETOPS_MAX_LEVELS; ++k)
      if (k == l) { obj = L.obj[k]; box = L.box[k]; gobj = L.gobj[k]; gbox = L.gbox[k]; plane = L.plane[k]; first = L.first[k]; }
    const int rem = t - first;
    const int loc = rem / A, a = rem - loc * A;
    const size_t o1 = (static_cast<size_t>(n) * A + a) * plane + loc;
    const size_t o4 = (static_cast<size_t>(n) * A + a) * 4 * plane + loc;
    const bool p = pos[i] != 0, q = neg[i] != 0;
    // objectness: binary_cross_entropy_with_logits(x, z), z = 1 for sampled positives, 0 for sampled negatives
    float g1 = 0.f;
    if (p || q) {
      const float x = obj[o1], z = p ? 1.f : 0.f;
      const float e = expf(-fabsf(x));
      s_obj += fmaxf(x, 0.f) - x * z + log1pf(e);
      const float sig = (x >= 0.f) ? 1.f / (1.f + e) : e / (1.f + e);
      g1 = sig - z;
      s_cnt += 1.f;
    }
    gobj[o1] = g1;
    float g4[4] = {0.f, 0.f, 0.f, 0.f};
    if (p) {
      const int64_t m = matched[i];
      const float4 gb = reinterpret_cast<const float4*>(gt)[static_cast<size_t>(n) * M + (m < 0 ? 0 : m)];
      const float4 an = reinterpret_cast<const float4*>(anchors)[t];
      // BoxCoder.encode (modeling/box_coder.py:27-51), "+1" widths
      const float ew = an.z - an.x + 1.f, eh = an.w - an.y + 1.f;
      const float ex = an.x + 0.5f * ew, ey = an.y + 0.5f * eh;
      const float gw = gb.z - gb.x + 1.f, gh = gb.w - gb.y + 1.f;
      const float gx = gb.x + 0.5f * gw, gy = gb.y + 0.5f * gh;
      const float tg[4] = {wx * (gx - ex) / ew, wy * (gy - ey) / eh, ww * logf(gw / ew), wh * logf(gh / eh)};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float d = box[o4 + static_cast<size_t>(k) * plane] - tg[k];
        const float ad = fabsf(d);
        if (ad < beta) { s_box += 0.5f * ad * ad * inv_beta; g4[k] = d * inv_beta; }
        else { s_box += ad - half_beta; g4[k] = (d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f); }
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) gbox[o4 + static_cast<size_t>(k) * plane] = g4[k];
  }
  __shared__ float red[3][kBlock / kWave];
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) {
    s_obj += __shfl_down(s_obj, off); s_box += __shfl_down(s_box, off); s_cnt += __shfl_down(s_cnt, off);
  }
  if ((threadIdx.x & (kWave - 1)) == 0) {
    red[0][threadIdx.x / kWave] = s_obj; red[1][threadIdx.x / kWave] = s_box; red[2][threadIdx.x / kWave] = s_cnt;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    float v = 0.f;
    for (int j = 0; j < kBlock / kWave; ++j) v += red[threadIdx.x][j];
    partial[blockIdx.x * 3 + threadIdx.x] = v;
  }
}

// fixed-order sum of the per-workgroup partials -> {objectness loss, box loss, 1 / max(#sampled, 1)}
__global__ void __launch_bounds__(kBlock)
rpn_loss_finish_kernel(const float* __restrict__ partial, int nblocks, float* __restrict__ out3) {
  __shared__ float red[3][kBlock / kWave];
  float v[3] = {0.f, 0.f, 0.f};
  for (int i = threadIdx.x; i < nblocks; i += kBlock) {
    v[0] += partial[i * 3]; v[1] += partial[i * 3 + 1]; v[2] += partial[i * 3 + 2];
  }
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) {
    v[0] += __shfl_down(v[0], off); v[1] += __shfl_down(v[1], off); v[2] += __shfl_down(v[2], off);
  }
  if ((threadIdx.x & (kWave - 1)) == 0)
    for (int k = 0; k < 3; ++k) red[k][threadIdx.x / kWave] = v[k];
  __syncthreads();
  if (threadIdx.x == 0) {
    float t[3] = {0.f, 0.f, 0.f};
    for (int j = 0; j < kBlock / kWave; ++j)
      for (int k = 0; k < 3; ++k) t[k] += red[k][j];
    const float inv = 1.f / fmaxf(t[2], 1.f);
    out3[0] = t[0] * inv;
    out3[1] = t[1] * inv;
    out3[2] = inv;
  }
}

// backward: grad *= upstream gradient of the loss * 1 / #sampled, in place, every level in one launch
__global__ void __launch_bounds__(kBlock)
rpn_loss_scale_kernel(RpnLevels L, int A, int N, const float* __restrict__ up_obj, const float* __restrict__ up_box,
                      const float* __restrict__ inv_count) {
  const float so = up_obj[0] * inv_count[0], sb = up_box[0] * inv_count[0];
  for (int l = 0; l < L.num; ++l) {
    const int64_t n1 = static_cast<int64_t>(N) * A * L.plane[l];
    float* go = L.gobj[l];
    float* gb = L.gbox[l];
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; i < 5 * n1;
         i += static_cast<int64_t>(gridDim.x) * kBlock) {
      if (i < n1) go[i] *= so; else gb[i - n1] *= sb;
    }
  }
}

// ------------------------------------------------------------------------------------------ RPN proposal decode
// One launch per feature level for the box path of RPNPostProcessor.forward_for_single_feature_map
// (modeling/rpn/inference.py:75-110): gather the selected anchors' deltas from the head output in ITS layout
// [N, 4A, H, W] (no permuted copy), BoxCoder.decode (modeling/box_coder.py:61-95), clip_to_image(remove_empty = False)
// (structures/bounding_box.py:202-213) and remove_small_boxes (structures/boxlist_ops.py:34-50) as a mask — ~35 elementwise
// launches per level as an ATen composition.  Same operations in the same order in fp32, contraction off.
//   idx [N, k]: positions in the (y, x, a) anchor order of permute_and_flatten (modeling/rpn/utils.py:9-13)
// writes  boxes_im [N][*][4] / scores_im [N][*] at the level's column offset of the image-major result (row strides
//         given), and the level's slice of the level-major NMS input: nms_boxes [N * k, 4], nms_scores [N * k] (boxes
//         failing min_size moved far away with score -1: they must not take part in NMS), ok [N * k]
__global__ void __launch_bounds__(kBlock)
rpn_decode_kernel(const float* __restrict__ reg, const int64_t* __restrict__ idx, const float* __restrict__ scores,
                  const float* __restrict__ anchors, const float* __restrict__ image_hw, int N, int A, int H, int W, int k,
                  float wx, float wy, float ww, float wh, float xform_clip, float min_size,
                  float* __restrict__ boxes_im, int64_t boxes_row_stride, float* __restrict__ scores_im,
                  int64_t scores_row_stride, float* __restrict__ nms_boxes, float* __restrict__ nms_scores,
                  uint8_t* __restrict__ ok_out) {
#pragma clang fp contract(off)
  const int64_t t = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (t >= static_cast<int64_t>(N) * k) return;
  const int n = static_cast<int>(t / k), j = static_cast<int>(t - static_cast<int64_t>(n) * k);
  int64_t i = idx[t];
  const int64_t last = static_cast<int64_t>(A) * H * W - 1;
  i = i < 0 ? 0 : (i > last ? last : i);          // a position outside the level would read outside the head output
  const int a = static_cast<int>(i % A);
  const int64_t pos = i / A;
  const int y = static_cast<int>(pos / W), x = static_cast<int>(pos - static_cast<int64_t>(y) * W);
  const size_t plane = static_cast<size_t>(H) * W;
  const float* r = reg + (static_cast<size_t>(n) * 4 * A + 4 * a) * plane + static_cast<size_t>(y) * W + x;
  const float4 an = reinterpret_cast<const float4*>(anchors)[i];
  const float w = an.z - an.x + 1.f, h = an.w - an.y + 1.f;
  const float cx = an.x + 0.5f * w, cy = an.y + 0.5f * h;
  const float dx = r[0] / wx, dy = r[plane] / wy;
  const float dw = fminf(r[2 * plane] / ww, xform_clip), dh = fminf(r[3 * plane] / wh, xform_clip);
  const float pcx = dx * w + cx, pcy = dy * h + cy;
  const float pw = expf(dw) * w, ph = expf(dh) * h;
  float x1 = pcx - 0.5f * pw, y1 = pcy - 0.5f * ph, x2 = pcx + 0.5f * pw - 1.f, y2 = pcy + 0.5f * ph - 1.f;
  const float hx = image_hw[2 * n + 1] - 1.f, hy = image_hw[2 * n] - 1.f;
  x1 = fminf(fmaxf(x1, 0.f), hx); y1 = fminf(fmaxf(y1, 0.f), hy);
  x2 = fminf(fmaxf(x2, 0.f), hx); y2 = fminf(fmaxf(y2, 0.f), hy);
  const float bw = x2 - x1 + 1.f, bh = y2 - y1 + 1.f;
  const bool ok = bw >= min_size && bh >= min_size;
  const float s = scores[t];
  reinterpret_cast<float4*>(boxes_im + static_cast<size_t>(n) * boxes_row_stride)[j] = make_float4(x1, y1, x2, y2);
  scores_im[static_cast<size_t>(n) * scores_row_stride + j] = s;
  reinterpret_cast<float4*>(nms_boxes)[t] = ok ? make_float4(x1, y1, x2, y2) : make_float4(-1e6f, -1e6f, -1e6f + 1.f, -1e6f + 1.f);
  nms_scores[t] = ok ? s : -1.f;
  ok_out[t] = ok ? 1 : 0;
}

// ------------------------------------------------------------------------------------------ labels / head targets
// Matcher output -> training labels in ONE launch (the where / eq / full_like chains of modeling/rpn/loss.py:108-118 and
// roi_heads/box_head/loss.py:56-72):  matched >= 0 -> the matched ground truth's class (1 without classes),
// BELOW_LOW_THRESHOLD (-1) -> 0, BETWEEN_THRESHOLDS (-2) -> -1, and -1 wherever `valid` (anchor visibility / proposal
// validity) is false.
template <typename T>
__global__ void __launch_bounds__(kBlock)
match_labels_kernel(const int64_t* __restrict__ matched, const int64_t* __restrict__ gt_labels,
                    const uint8_t* __restrict__ valid, int64_t total, int K, int M, T* __restrict__ out) {
  const int64_t t = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (t >= total) return;
  const int64_t m = matched[t];
  T v = static_cast<T>(-1);
  if (m >= 0) v = gt_labels ? (m < M ? static_cast<T>(gt_labels[(t / K) * M + m]) : static_cast<T>(-1)) : static_cast<T>(1);
  else if (m == -1) v = static_cast<T>(0);
  if (valid && !valid[t]) v = static_cast<T>(-1);
  out[t] = v;
}

// The sampled slots of the box head in ONE launch (roi_heads/box_head/loss.py:56-110 of the reference does this per
// image with ~25 indexing launches after encoding ALL proposals): slot (n, j) takes proposal i = idx[n][j] and gets its
// box, class label (as above; -1 for an unfilled slot), BoxCoder.encode(matched gt, box) (box_coder.py:27-51),
// matched index and objectness.
__global__ void __launch_bounds__(kBlock)
roi_head_targets_kernel(const float* __restrict__ boxes, const int64_t* __restrict__ matched, const float* __restrict__ gt,
                        const int64_t* __restrict__ gt_labels, const uint8_t* __restrict__ valid,
                        const int64_t* __restrict__ idx, const uint8_t* __restrict__ slot_valid,
                        const float* __restrict__ objectness, int N, int K, int M, int B, float wx, float wy, float ww,
                        float wh, float* __restrict__ out_boxes, int64_t* __restrict__ out_labels,
                        float* __restrict__ out_reg, int64_t* __restrict__ out_matched, float* __restrict__ out_obj) {
#pragma clang fp contract(off)
  const int64_t t = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (t >= static_cast<int64_t>(N) * B) return;
  const int n = static_cast<int>(t / B);
  int64_t i = idx[t];
  i = i < 0 ? 0 : (i >= K ? K - 1 : i);
  const size_t src = static_cast<size_t>(n) * K + i;
  const int64_t m = matched[src];
  const float4 b = reinterpret_cast<const float4*>(boxes)[src];
  const int64_t mg = m < 0 ? 0 : (m >= M ? M - 1 : m);     // the matched row (row 0 when there is none: value unused)
  const float4 g = reinterpret_cast<const float4*>(gt)[static_cast<size_t>(n) * M + mg];
  int64_t label = -1;
  if (slot_valid[t] && (!valid || valid[src])) label = m >= 0 ? (m < M ? gt_labels[static_cast<size_t>(n) * M + m] : -1) : (m == -1 ? 0 : -1);
  const float ew = b.z - b.x + 1.f, eh = b.w - b.y + 1.f;
  const float ex = b.x + 0.5f * ew, ey = b.y + 0.5f * eh;
  const float gw = g.z - g.x + 1.f, gh = g.w - g.y + 1.f;
  const float gx = g.x + 0.5f * gw, gy = g.y + 0.5f * gh;
  reinterpret_cast<float4*>(out_boxes)[t] = b;
  reinterpret_cast<float4*>(out_reg)[t] = make_float4(wx * (gx - ex) / ew, wy * (gy - ey) / eh, ww * logf(gw / ew), wh * logf(gh / eh));
  out_labels[t] = label;
  out_matched[t] = m;
  if (out_obj) out_obj[t] = objectness ? objectness[src] : 0.f;
}

}  // namespace

DETOPS_API size_t detops_match_boxes_workspace_bytes(int N, int M) {
  if (N <= 0 || M <= 0) return 256;
  return up256(sizeof(int32_t) * static_cast<size_t>(N) * M);
}

DETOPS_API int detops_match_boxes_f32(const float* gt_boxes, const uint8_t* gt_valid, const float* boxes,
                                      int boxes_batched, int N, int M, int K, float high_threshold,
                                      float low_threshold, int allow_low_quality_matches, int64_t* matched_idxs,
                                      void* workspace, size_t workspace_bytes, detops_stream_t stream) {
  if (N < 0 || M <= 0 || K < 0) return DETOPS_EINVAL;
  if (N == 0 || K == 0) return 0;
  if (!gt_boxes || !gt_valid || !boxes || !matched_idxs) return DETOPS_EINVAL;
  hipStream_t st = as_stream(stream);
  const dim3 grid(static_cast<unsigned>(ceil_div64(K, kBlock * kMatchPer)), static_cast<unsigned>(N));
  int32_t* best = nullptr;
  if (allow_low_quality_matches) {
    if (!workspace || workspace_bytes < detops_match_boxes_workspace_bytes(N, M)) return DETOPS_EWORKSPACE;
    best = static_cast<int32_t*>(workspace);
    DETOPS_HIP_TRY(hipMemsetAsync(best, 0, sizeof(int32_t) * static_cast<size_t>(N) * M, st));
  }
  hipLaunchKernelGGL(match_kernel<0>, grid, dim3(kBlock), 0, st, gt_boxes, gt_valid, boxes, boxes_batched, M, K,
                     high_threshold, low_threshold, allow_low_quality_matches, best, matched_idxs);
  if (allow_low_quality_matches)
    hipLaunchKernelGGL(match_kernel<1>, grid, dim3(kBlock), 0, st, gt_boxes, gt_valid, boxes, boxes_batched, M, K,
                       high_threshold, low_threshold, allow_low_quality_matches, best, matched_idxs);
  return launch_status();
}

DETOPS_API int detops_rpn_decode_f32(const float* box_regression, const int64_t* topk_idx, const float* topk_scores,
                                     const float* anchors, const float* image_hw, int N, int A, int H, int W, int k,
                                     float wx, float wy, float ww, float wh, float bbox_xform_clip, float min_size,
                                     float* boxes, int64_t boxes_row_stride, float* scores, int64_t scores_row_stride,
                                     float* nms_boxes, float* nms_scores, uint8_t* ok, detops_stream_t stream) {
  if (N < 0 || A <= 0 || H <= 0 || W <= 0 || k < 0) return DETOPS_EINVAL;
  if (N == 0 || k == 0) return 0;
  if (!box_regression || !topk_idx || !topk_scores || !anchors || !image_hw || !boxes || !scores || !nms_boxes ||
      !nms_scores || !ok || boxes_row_stride < 4 * static_cast<int64_t>(k) || scores_row_stride < k)
    return DETOPS_EINVAL;
  const int64_t total = static_cast<int64_t>(N) * k;
  hipLaunchKernelGGL(rpn_decode_kernel, dim3(static_cast<unsigned>(ceil_div64(total, kBlock))), dim3(kBlock), 0, as_stream(stream),
                     box_regression, topk_idx, topk_scores, anchors, image_hw, N, A, H, W, k, wx, wy, ww, wh, bbox_xform_clip,
                     min_size, boxes, boxes_row_stride, scores, scores_row_stride, nms_boxes, nms_scores, ok);
  return launch_status();
}

DETOPS_API int detops_match_labels(const int64_t* matched, const int64_t* gt_labels, const uint8_t* valid, int N, int K,
                                   int M, int out_dtype, void* out, detops_stream_t stream) {
  if (N < 0 || K < 0 || M < 0 || (out_dtype != 0 && out_dtype != 1)) return DETOPS_EINVAL;
  const int64_t total = static_cast<int64_t>(N) * K;
  if (total == 0) return 0;
  if (!matched || !out || (gt_labels && M == 0)) return DETOPS_EINVAL;
  const dim3 grid(static_cast<unsigned>(ceil_div64(total, kBlock)));
  if (out_dtype == 0)
    hipLaunchKernelGGL(match_labels_kernel<float>, grid, dim3(kBlock), 0, as_stream(stream), matched, gt_labels, valid, total, K,
                       M, static_cast<float*>(out));
  else
    hipLaunchKernelGGL(match_labels_kernel<int64_t>, grid, dim3(kBlock), 0, as_stream(stream), matched, gt_labels, valid, total,
                       K, M, static_cast<int64_t*>(out));
  return launch_status();
}

DETOPS_API int detops_roi_head_targets_f32(const float* boxes, const int64_t* matched, const float* gt_boxes,
                                           const int64_t* gt_labels, const uint8_t* valid, const int64_t* idx,
                                           const uint8_t* slot_valid, const float* objectness, int N, int K, int M, int B,
                                           float wx, float wy, float ww, float wh, float* out_boxes, int64_t* out_labels,
                                           float* out_regression_targets, int64_t* out_matched, float* out_objectness,
                                           detops_stream_t stream) {
  if (N < 0 || K < 0 || M < 0 || B < 0) return DETOPS_EINVAL;
  if (N == 0 || B == 0) return 0;
  if (K == 0 || M == 0 || !boxes || !matched || !gt_boxes || !gt_labels || !idx || !slot_valid || !out_boxes || !out_labels ||
      !out_regression_targets || !out_matched)
    return DETOPS_EINVAL;
  const int64_t total = static_cast<int64_t>(N) * B;
  hipLaunchKernelGGL(roi_head_targets_kernel, dim3(static_cast<unsigned>(ceil_div64(total, kBlock))), dim3(kBlock), 0,
                     as_stream(stream), boxes, matched, gt_boxes, gt_labels, valid, idx, slot_valid, objectness, N, K, M, B, wx, wy,
                     ww, wh, out_boxes, out_labels, out_regression_targets, out_matched, out_objectness);
  return launch_status();
}

DETOPS_API size_t detops_sample_labels_workspace_bytes(int N, int batch_size_per_image) {
  if (N <= 0) return 256;
  return sampler_layout(N, batch_size_per_image).total;
}

// seed_dev (optional): a device word mixed into `seed` when the kernels RUN — for callers that capture the launch into a HIP
// graph (engine/graph_step.py): the replayed launch arguments are frozen, the word is not.
DETOPS_API int detops_sample_labels_dseed(const void* labels, int label_dtype, int N, int n, int batch_size_per_image,
                                          int max_positives, uint64_t seed, const uint64_t* seed_dev, uint8_t* pos_mask,
                                          uint8_t* neg_mask, int64_t* sampled_idx, uint8_t* sampled_valid, void* workspace,
                                          size_t workspace_bytes, detops_stream_t stream) {
  if (N < 0 || n < 0 || batch_size_per_image <= 0 || max_positives < 0) return DETOPS_EINVAL;
  if (N == 0 || n == 0) return 0;
  if (!labels || !pos_mask || !neg_mask || (sampled_idx != nullptr) != (sampled_valid != nullptr)) return DETOPS_EINVAL;
  if (batch_size_per_image > 512) return DETOPS_EUNSUPPORTED;   // survivor sort: 16 x quota 64-bit keys in <= 64 KiB of LDS
  if (!workspace || workspace_bytes < detops_sample_labels_workspace_bytes(N, batch_size_per_image)) return DETOPS_EWORKSPACE;
  hipStream_t st = as_stream(stream);
  const unsigned long long* sd = reinterpret_cast<const unsigned long long*>(seed_dev);
  switch (label_dtype) {
    case DETOPS_LABEL_F32:
      return run_sampler<float>(labels, N, n, batch_size_per_image, max_positives, seed, sd, pos_mask, neg_mask, sampled_idx,
                                sampled_valid, workspace, st);
    case DETOPS_LABEL_I64:
      return run_sampler<int64_t>(labels, N, n, batch_size_per_image, max_positives, seed, sd, pos_mask, neg_mask,
                                  sampled_idx, sampled_valid, workspace, st);
    default:
      return DETOPS_EUNSUPPORTED;
  }
}

DETOPS_API int detops_sample_labels(const void* labels, int label_dtype, int N, int n, int batch_size_per_image,
                                    int max_positives, uint64_t seed, uint8_t* pos_mask, uint8_t* neg_mask,
                                    int64_t* sampled_idx, uint8_t* sampled_valid, void* workspace,
                                    size_t workspace_bytes, detops_stream_t stream) {
  return detops_sample_labels_dseed(labels, label_dtype, N, n, batch_size_per_image, max_positives, seed, nullptr, pos_mask,
                                    neg_mask, sampled_idx, sampled_valid, workspace, workspace_bytes, stream);
}

DETOPS_API int detops_mask_targets(const void* masks, int mask_dtype, const int64_t* mask_index, const float* boxes,
                                   int G, int H, int W, int P, int M, float* out, detops_stream_t stream) {
  if (G < 0 || H <= 0 || W <= 0 || P < 0 || M <= 0) return DETOPS_EINVAL;
  if (P == 0) return 0;
  if (G == 0 || !masks || !mask_index || !boxes || !out) return DETOPS_EINVAL;
  hipStream_t st = as_stream(stream);
  const dim3 grid(static_cast<unsigned>(P));
  if (mask_dtype == DETOPS_MASK_U8)
    hipLaunchKernelGGL((mask_targets_kernel<uint8_t, 1>), grid, dim3(kBlock), 0, st, static_cast<const uint8_t*>(masks),
                       mask_index, boxes, G, H, W, M, out);
  else if (mask_dtype == DETOPS_MASK_BOOL)
    hipLaunchKernelGGL((mask_targets_kernel<uint8_t, 2>), grid, dim3(kBlock), 0, st, static_cast<const uint8_t*>(masks),
                       mask_index, boxes, G, H, W, M, out);
  else if (mask_dtype == DETOPS_MASK_F32)
    hipLaunchKernelGGL((mask_targets_kernel<float, 0>), grid, dim3(kBlock), 0, st, static_cast<const float*>(masks),
                       mask_index, boxes, G, H, W, M, out);
  else
    return DETOPS_EUNSUPPORTED;
  return launch_status();
}

static int rpn_levels(RpnLevels& L, const float* const* obj, const float* const* box, float* const* gobj,
                      float* const* gbox, const int* H, const int* W, int num_levels, int A, int T, bool need_inputs) {
  if (num_levels < 1 || num_levels > DETOPS_MAX_LEVELS || A <= 0 || !gobj || !gbox || !H || !W) return DETOPS_EINVAL;
  if (need_inputs && (!obj || !box)) return DETOPS_EINVAL;
  L = RpnLevels{};
  L.num = num_levels;
  int64_t first = 0;
  for (int l = 0; l < num_levels; ++l) {
    if (H[l] <= 0 || W[l] <= 0 || !gobj[l] || !gbox[l] || (need_inputs && (!obj[l] || !box[l]))) return DETOPS_EINVAL;
    L.obj[l] = need_inputs ? obj[l] : nullptr; L.box[l] = need_inputs ? box[l] : nullptr;
    L.gobj[l] = gobj[l]; L.gbox[l] = gbox[l];
    L.plane[l] = H[l] * W[l];
    L.first[l] = static_cast<int>(first);
    first += static_cast<int64_t>(H[l]) * W[l] * A;
  }
  if (first != T) return DETOPS_EINVAL;   // the anchor list must be exactly the concatenation of the level grids
  for (int l = num_levels; l <= DETOPS_MAX_LEVELS; ++l) L.first[l] = T;
  return 0;
}

DETOPS_API size_t detops_rpn_loss_workspace_bytes(void) { return up256(sizeof(float) * 3 * kRpnMaxBlocks); }

DETOPS_API int detops_rpn_loss_f32(const float* const* objectness_host, const float* const* box_regression_host,
                                   const int* H_host, const int* W_host, int num_levels, int anchors_per_location,
                                   const float* anchors, const int64_t* matched_idxs, const uint8_t* pos_mask,
                                   const uint8_t* neg_mask, const float* gt_boxes, int N, int M, int T, float beta,
                                   const float* weights4_host, float* const* grad_objectness_host,
                                   float* const* grad_box_regression_host, float* losses3, void* workspace,
                                   size_t workspace_bytes, detops_stream_t stream) {
  if (N <= 0 || M <= 0 || T <= 0 || !(beta > 0.f) || !weights4_host) return DETOPS_EINVAL;
  if (!anchors || !matched_idxs || !pos_mask || !neg_mask || !gt_boxes || !losses3) return DETOPS_EINVAL;
  if (!workspace || workspace_bytes < detops_rpn_loss_workspace_bytes()) return DETOPS_EWORKSPACE;
  RpnLevels L;
  if (const int rc = rpn_levels(L, objectness_host, box_regression_host, grad_objectness_host, grad_box_regression_host,
                                H_host, W_host, num_levels, anchors_per_location, T, true))
    return rc;
  hipStream_t st = as_stream(stream);
  const int blocks = static_cast<int>(std::min<int64_t>(ceil_div64(static_cast<int64_t>(N) * T, kBlock), kRpnMaxBlocks));
  float* partial = static_cast<float*>(workspace);
  hipLaunchKernelGGL(rpn_loss_kernel, dim3(blocks), dim3(kBlock), 0, st, L, anchors_per_location, anchors, matched_idxs,
                     pos_mask, neg_mask, gt_boxes, N, M, T, beta, weights4_host[0], weights4_host[1], weights4_host[2],
                     weights4_host[3], partial);
  hipLaunchKernelGGL(rpn_loss_finish_kernel, dim3(1), dim3(kBlock), 0, st, partial, blocks, losses3);
  return launch_status();
}

DETOPS_API int detops_rpn_loss_backward_f32(float* const* grad_objectness_host, float* const* grad_box_regression_host,
                                            const int* H_host, const int* W_host, int num_levels,
                                            int anchors_per_location, int N, int T, const float* upstream_objectness,
                                            const float* upstream_box, const float* inv_count,
                                            detops_stream_t stream) {
  if (N <= 0 || T <= 0 || !upstream_objectness || !upstream_box || !inv_count) return DETOPS_EINVAL;
  RpnLevels L;
  if (const int rc = rpn_levels(L, nullptr, nullptr, grad_objectness_host, grad_box_regression_host, H_host, W_host,
                                num_levels, anchors_per_location, T, false))
    return rc;
  const int blocks = static_cast<int>(std::min<int64_t>(ceil_div64(static_cast<int64_t>(N) * T * 5, kBlock * 4), kRpnMaxBlocks));
  hipLaunchKernelGGL(rpn_loss_scale_kernel, dim3(std::max(blocks, 1)), dim3(kBlock), 0, as_stream(stream), L,
                     anchors_per_location, N, upstream_objectness, upstream_box, inv_count);
  return launch_status();
}
