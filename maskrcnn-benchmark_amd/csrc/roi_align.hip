// roi_align.hip — ROIAlign forward / backward for gfx950 (MI355X), fp32 NCHW.
//
// Replaces the reference's RoIAlignForward / RoIAlignBackwardFeature
// (maskrcnn_benchmark/csrc/cuda/ROIAlign_cuda.cu:64-122, :177-254) and its CPU forward
// (csrc/cpu/ROIAlign_cpu.cpp:113-219) behind detops_roi_align_{forward,backward}_f32 and the
// multi-level detops_roi_align_fpn_* entry points (include/detops.h).
//
// Design (not a translation of the reference's one-thread-per-output kernels):
//   * one workgroup per (ROI, channel chunk).  The sampling geometry of a ROI is SEPARABLE:
//     the y taps depend on (ph, iy) only and the x taps on (pw, ix) only, so wave 0.. build two
//     small axis tables in LDS — PH*gh + PW*gw entries of {low, high, frac, 1-frac} — once per
//     workgroup instead of re-deriving 4 indices + 4 weights per output element per channel
//     (the reference CUDA kernel redoes that C times; the CPU kernel keeps a PH*PW*gh*gw table).
//   * the per-sample arithmetic keeps the reference's evaluation order with FP contraction off
//     (w = hy*hx ...; val = w1*v1 + w2*v2 + w3*v3 + w4*v4; acc += val; acc /= count), so the
//     forward is bit-identical to the reference CPU kernel for finite inputs.
//   * lanes run over (channel, bin) with bin fastest: stores are fully coalesced (the output of a
//     (ROI, chunk) is one contiguous run), gathers hit a <= ~30x30 px patch per channel that
//     stays L1/L2 resident.
//   * forward fast path (fixed 1x1 / 2x2 sampling): the ROI footprint is staged in LDS, a thread
//     keeps its bin's sample offsets + weights in registers across the channel loop (see below).
//   * backward: pixel-owner gather (roi_align_bwd_gather_kernel) — the adjoint of the separable
//     sampling is two tiny per-axis matrices per ROI; a workgroup owns an 8 x 32 pixel tile of one
//     gradient map, every thread ONE pixel with its channel sums in registers: no atomics of any
//     kind, no zero-fill pass, bit-reproducible (213 us per launch; the LDS-scatter tile kernel it
//     replaced, kept below as the fallback for huge bin counts, took 1463 us).
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "detops_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kTabBig = 512;    // axis-table entries per axis kept in LDS (adaptive grids);
constexpr int kTabSmall = 32;   // fixed sampling_ratio: PH*sr, PW*sr <= 32 covers 7x7..14x14 @ sr 2

// 16-byte global load from a 4-byte-aligned address (gfx950 global_load_dwordx4 needs dword alignment only)
#ifdef DETOPS_CPU_EMU
__device__ __forceinline__ float4 load_float4_dword_aligned(const float* p) { return make_float4(p[0], p[1], p[2], p[3]); }
#else
typedef float float4_a4 __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ float4 load_float4_dword_aligned(const float* p) {
  const float4_a4 v = *reinterpret_cast<const float4_a4*>(p);
  return make_float4(v.x, v.y, v.z, v.w);
}
#endif

struct __align__(16) Tap {
  int lo, hi;   // y axis: pre-multiplied by W
  float l, h;   // frac, 1-frac (both 0 for a sample outside the map)
};

struct Level {
  const float* in;  // forward: feature map; backward: unused
  float* gin;       // backward: gradient map
  int H, W;
  float scale;
};

struct Levels {
  Level lv[DETOPS_MAX_LEVELS];
  int num;
  // LevelMapper parameters (only read when num > 1 and levels == nullptr)
  int k_min, k_max;
  float s0, lvl0, eps;
};

// One axis sample, reference order of operations (ROIAlign_cpu.cpp:36-92 / ROIAlign_cuda.cu:15-49).
__device__ __forceinline__ Tap axis_entry(float start, float bin, int p, int i, int grid, int size,
                                          int premul) {
#pragma clang fp contract(off)
  Tap t;
  float c = start + p * bin + static_cast<float>(i + .5f) * bin / static_cast<float>(grid);
  if (c < -1.0f || c > static_cast<float>(size)) {
    t.lo = 0; t.hi = 0; t.l = 0.f; t.h = 0.f;
    return t;
  }
  if (c <= 0.f) c = 0.f;
  int lo = static_cast<int>(c), hi;
  if (lo >= size - 1) {
    hi = lo = size - 1;
    c = static_cast<float>(lo);
  } else {
    hi = lo + 1;
  }
  const float l = c - static_cast<float>(lo);
  t.lo = lo * premul;
  t.hi = hi * premul;
  t.l = l;
  t.h = static_cast<float>(1. - static_cast<double>(l));  // `T hy = 1. - ly` (double literal)
  return t;
}

struct RoiGeom {
  int b;
  float start_w, start_h, bin_w, bin_h;
  int gh, gw;
  float count;
};

__device__ __forceinline__ RoiGeom roi_geometry(const float* __restrict__ roi, float scale, int PH,
                                                int PW, int sr) {
#pragma clang fp contract(off)
  RoiGeom g;
  g.b = static_cast<int>(roi[0]);
  g.start_w = roi[1] * scale;
  g.start_h = roi[2] * scale;
  const float end_w = roi[3] * scale;
  const float end_h = roi[4] * scale;
  const float rw = fmaxf(end_w - g.start_w, 1.f);
  const float rh = fmaxf(end_h - g.start_h, 1.f);
  g.bin_h = rh / static_cast<float>(PH);
  g.bin_w = rw / static_cast<float>(PW);
  g.gh = (sr > 0) ? sr : static_cast<int>(ceilf(rh / PH));
  g.gw = (sr > 0) ? sr : static_cast<int>(ceilf(rw / PW));
  g.count = static_cast<float>(g.gh * g.gw);
  return g;
}

// LevelMapper (reference modeling/poolers.py:33-42) on device, fp32 like torch.
__device__ __forceinline__ int fpn_level(const float* __restrict__ roi, const Levels& L) {
#pragma clang fp contract(off)
  const float area = (roi[3] - roi[1] + 1.f) * (roi[4] - roi[2] + 1.f);
  const float s = sqrtf(area);
  float t = floorf(L.lvl0 + log2f(s / L.s0 + L.eps));
  t = fminf(fmaxf(t, static_cast<float>(L.k_min)), static_cast<float>(L.k_max));
  return static_cast<int>(t) - L.k_min;
}

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
template <int PH_, int PW_, int kTabCap>
__global__ void __launch_bounds__(kBlock)
roi_align_fwd_kernel(Levels L, const float* __restrict__ rois, const int32_t* __restrict__ levels_in,
                     int32_t* __restrict__ levels_out, float* __restrict__ out, int C, int K,
                     int PHr, int PWr, int sr, int CT, int chunks) {
  const int PH = PH_ ? PH_ : PHr;
  const int PW = PW_ ? PW_ : PWr;
  const int bins = PH * PW;
  __shared__ Tap tabY[kTabCap];
  __shared__ Tap tabX[kTabCap];

  const int bid = blockIdx.x;
  const int k = bid / chunks;
  const int chunk = bid - k * chunks;
  const float* roi = rois + static_cast<size_t>(k) * 5;
  int lvl = 0;
  if (L.num > 1) lvl = levels_in ? levels_in[k] : fpn_level(roi, L);
  if (levels_out && chunk == 0 && threadIdx.x == 0) levels_out[k] = lvl;
  // wave-uniform select (num <= 8): avoids dynamic indexing of the kernarg struct
  const float* in = L.lv[0].in; int H = L.lv[0].H, W = L.lv[0].W; float scale = L.lv[0].scale;
#pragma unroll
  for (int i = 1; i < DETOPS_MAX_LEVELS; ++i)
    if (i == lvl) { in = L.lv[i].in; H = L.lv[i].H; W = L.lv[i].W; scale = L.lv[i].scale; }

  const RoiGeom g = roi_geometry(roi, scale, PH, PW, sr);
  const int ny = PH * g.gh, nx = PW * g.gw;
  const bool use_tab = (ny <= kTabCap) && (nx <= kTabCap);
  if (use_tab) {
    for (int t = threadIdx.x; t < ny + nx; t += kBlock) {
      if (t < ny) {
        tabY[t] = axis_entry(g.start_h, g.bin_h, t / g.gh, t % g.gh, g.gh, H, W);
      } else {
        const int u = t - ny;
        tabX[u] = axis_entry(g.start_w, g.bin_w, u / g.gw, u % g.gw, g.gw, W, 1);
      }
    }
    __syncthreads();
  }

  const int c0 = chunk * CT;
  const int cend = min(C, c0 + CT);
  const int total = (cend - c0) * bins;
  const size_t plane = static_cast<size_t>(H) * W;
  const float* base = in + (static_cast<size_t>(g.b) * C + c0) * plane;
  float* obase = out + (static_cast<size_t>(k) * C + c0) * bins;

  for (int o = threadIdx.x; o < total; o += kBlock) {
#pragma clang fp contract(off)
    const int cl = o / bins;
    const int bin = o - cl * bins;
    const int ph = bin / PW;
    const int pw = bin - ph * PW;
    const float* d = base + static_cast<size_t>(cl) * plane;
    float acc = 0.f;
    for (int iy = 0; iy < g.gh; ++iy) {
      const Tap ty = use_tab ? tabY[ph * g.gh + iy]
                             : axis_entry(g.start_h, g.bin_h, ph, iy, g.gh, H, W);
      const float* r0 = d + ty.lo;
      const float* r1 = d + ty.hi;
      for (int ix = 0; ix < g.gw; ++ix) {
        const Tap tx = use_tab ? tabX[pw * g.gw + ix]
                               : axis_entry(g.start_w, g.bin_w, pw, ix, g.gw, W, 1);
        const float w1 = ty.h * tx.h, w2 = ty.h * tx.l, w3 = ty.l * tx.h, w4 = ty.l * tx.l;
        const float v1 = r0[tx.lo], v2 = r0[tx.hi], v3 = r1[tx.lo], v4 = r1[tx.hi];
        acc += w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
      }
    }
    obase[o] = acc / g.count;
  }
}

// ------------------------------------------------------------------------------------------
// forward, fast path: fixed sampling grid (SR x SR, SR in {1,2}), compile-time bin counts.
//
//   * the ROI's footprint [channels x (py+1) x (px+1)] is staged into LDS with row-contiguous
//     global loads (each feature byte under the ROI crosses the memory system once per workgroup,
//     coalesced, instead of 4*SR*SR times as scattered 4-byte gathers that each cost the texture
//     path a cache-line lookup);
//   * a thread owns ONE output bin for a strided set of channels: its SR*SR samples' patch offsets
//     and 4 bilinear weights live in registers for the whole channel loop, so the inner loop is
//     4 LDS reads + 7 fp32 ops per sample and nothing else;
//   * taps are addressed as (lo, lo+1): the patch carries one extra row/column that replicates the
//     clamped border pixel, which is exactly what the reference reads when x_high == x_low
//     (weight 0 on that tap) — so the two taps of a row are adjacent and fetched as one
//     ds_read2_b32;
//   * the arithmetic keeps the reference's order with FP contraction off: bit-identical to the
//     reference CPU kernel for finite inputs.
// Footprints that do not fit the LDS budget fall through to the generic gather loop.
// ------------------------------------------------------------------------------------------
constexpr int kFwdDmaWps = 5;  // LDS-DMA forward: no staging registers
constexpr int kFwdWps = 4;   // waves per SIMD of the row-vector forward (register budget 80)
constexpr int kLdsPatchFloats = 8192 - 64;  // default ~32 KiB dynamic LDS per workgroup -> 4-5 workgroups / CU

// Register-staged variant (round 1; kept behind DETOPS_ROIALIGN_FWD=lds for A/B runs against the LDS-DMA kernel below).
// U = staging loads in flight per lane; patch_floats = LDS patch budget (DETOPS_ROIALIGN_FWD_LDS_KB /
// DETOPS_ROIALIGN_FWD_U select other points of the occupancy / loads-in-flight trade-off at run time)
// WPS: waves per SIMD the register allocation must allow (residency hides the stage -> compute latency chain)
template <int PH, int PW, int SR, int G, int U, int WPS>
__global__ void __launch_bounds__(((PH * PW * G + 63) / 64) * 64, WPS)
roi_align_fwd_lds_kernel(Levels L, const float* __restrict__ rois, const int32_t* __restrict__ levels_in,
                         int32_t* __restrict__ levels_out, float* __restrict__ out, int C, int K, int CT,
                         int chunks, int patch_floats) {
  constexpr int BINS = PH * PW;
  constexpr int NS = SR * SR;
  constexpr int NT = ((BINS * G + 63) / 64) * 64;
  DETOPS_DYNAMIC_LDS(float, patch);
  __shared__ Tap tabY[PH * SR];
  __shared__ Tap tabX[PW * SR];
  __shared__ int s_bounds[4];

  const int tid = threadIdx.x;
  const int bid = blockIdx.x;
  const int k = bid / chunks;
  const int chunk = bid - k * chunks;
  const float* roi = rois + static_cast<size_t>(k) * 5;
  int lvl = 0;
  if (L.num > 1) lvl = levels_in ? levels_in[k] : fpn_level(roi, L);
  if (levels_out && chunk == 0 && tid == 0) levels_out[k] = lvl;
  const float* in = L.lv[0].in; int H = L.lv[0].H, W = L.lv[0].W; float scale = L.lv[0].scale;
#pragma unroll
  for (int i = 1; i < DETOPS_MAX_LEVELS; ++i)
    if (i == lvl) { in = L.lv[i].in; H = L.lv[i].H; W = L.lv[i].W; scale = L.lv[i].scale; }

  const RoiGeom g = roi_geometry(roi, scale, PH, PW, SR);
  if (tid == 0) { s_bounds[0] = 0x7fffffff; s_bounds[1] = -1; s_bounds[2] = 0x7fffffff; s_bounds[3] = -1; }
  __syncthreads();
  if (tid < PH * SR + PW * SR) {
    if (tid < PH * SR) {
      const Tap e = axis_entry(g.start_h, g.bin_h, tid / SR, tid % SR, SR, H, 1);
      tabY[tid] = e;
      if (e.l != 0.f || e.h != 0.f) { atomicMin(&s_bounds[0], e.lo); atomicMax(&s_bounds[1], e.lo); }
    } else {
      const int u = tid - PH * SR;
      const Tap e = axis_entry(g.start_w, g.bin_w, u / SR, u % SR, SR, W, 1);
      tabX[u] = e;
      if (e.l != 0.f || e.h != 0.f) { atomicMin(&s_bounds[2], e.lo); atomicMax(&s_bounds[3], e.lo); }
    }
  }
  __syncthreads();
  const int c0 = chunk * CT;
  const int cend = min(C, c0 + CT);
  float* obase = out + (static_cast<size_t>(k) * C + c0) * BINS;
  if (s_bounds[1] < 0 || s_bounds[3] < 0) {  // every sample falls outside the map: all-zero output
    for (int o = tid; o < (cend - c0) * BINS; o += NT) obase[o] = 0.f;
    return;
  }
  const int ymin = s_bounds[0];
  const int rows = s_bounds[1] - ymin + 2;   // (lo range) + the lo+1 row
  const size_t plane = static_cast<size_t>(H) * W;
  const float* base = in + (static_cast<size_t>(g.b) * C + c0) * plane;
  // Row-vector staging: each patch row is copied as 16-byte pieces by a group of 8..64 lanes — one address
  // computation per 16 bytes instead of ~25 VALU instructions of index arithmetic per staged dword (rocprofv3
  // PMC of the dword version, profiles/r02a_pmc_diag.txt: 51 M VALU wave-instructions per launch, 2/3 of them
  // staging index math).  The global side is only 4-byte aligned (a patch starts at any column of any map
  // width): gfx950 global_load_dwordx4 takes dword-aligned addresses; the LDS side is 16-byte aligned.
  const int xmin = s_bounds[2];
  int ps = (s_bounds[3] - xmin + 2 + 3) & ~3;   // columns incl. the lo+1 column, whole float4s per row
  if (!((ps >> 2) & 1)) ps += 4;                // odd number of float4 per row: rows spread over the LDS banks
  const int area = rows * ps;

  if (area > patch_floats) {  // footprint too large for LDS: gather straight from the map
    for (int o = tid; o < (cend - c0) * BINS; o += NT) {
#pragma clang fp contract(off)
      const int cl = o / BINS;
      const int bin = o - cl * BINS;
      const int ph = bin / PW, pw = bin - ph * PW;
      const float* d = base + static_cast<size_t>(cl) * plane;
      float acc = 0.f;
      for (int iy = 0; iy < SR; ++iy) {
        const Tap ty = tabY[ph * SR + iy];
        const float* r0 = d + ty.lo * W;
        const float* r1 = d + ty.hi * W;
        for (int ix = 0; ix < SR; ++ix) {
          const Tap tx = tabX[pw * SR + ix];
          const float w1 = ty.h * tx.h, w2 = ty.h * tx.l, w3 = ty.l * tx.h, w4 = ty.l * tx.l;
          acc += w1 * r0[tx.lo] + w2 * r0[tx.hi] + w3 * r1[tx.lo] + w4 * r1[tx.hi];
        }
      }
      obase[o] = acc / g.count;
    }
    return;
  }

  // per-thread sample geometry (registers)
  const int bin = tid % BINS;
  const int csub = tid / BINS;  // >= G: idle lane of the last wave
  const int ph = bin / PW, pw = bin - ph * PW;
  int off[NS];
  float w1[NS], w2[NS], w3[NS], w4[NS];
#pragma unroll
  for (int iy = 0; iy < SR; ++iy) {
    const Tap ty = tabY[ph * SR + iy];
    const bool vy = (ty.l != 0.f || ty.h != 0.f);
#pragma unroll
    for (int ix = 0; ix < SR; ++ix) {
#pragma clang fp contract(off)
      const Tap tx = tabX[pw * SR + ix];
      const bool v = vy && (tx.l != 0.f || tx.h != 0.f);
      const int s = iy * SR + ix;
      off[s] = v ? (ty.lo - ymin) * ps + (tx.lo - xmin) : 0;
      w1[s] = v ? ty.h * tx.h : 0.f;
      w2[s] = v ? ty.h * tx.l : 0.f;
      w3[s] = v ? ty.l * tx.h : 0.f;
      w4[s] = v ? ty.l * tx.l : 0.f;
    }
  }
  const float inv_count = 1.f / static_cast<float>(NS);  // NS in {1,4}: exact reciprocal

  // staging: element-linear over the patch (lanes run along a patch row, wrap to the next row), four
  // independent loads in flight per lane before the first LDS store (the loop is latency-bound
  // otherwise: one L2/MALL round trip per iteration)
  const float inv_rows = 1.f / static_cast<float>(rows);
  const int ctb = min(cend - c0, patch_floats / area);
  for (int cs = c0; cs < cend; cs += ctb) {
    const int cn = min(ctb, cend - cs);
    const float* src = base + static_cast<size_t>(cs - c0) * plane;
    if (tid == 0) { DETOPS_STAT("fwd.stage_batches", 1); DETOPS_STAT("fwd.staged_floats", cn * area); }
    {
      const int w4 = ps >> 2;
      int lw = 8;
      while (lw < w4) lw <<= 1;                 // lanes per patch row (wave-uniform, <= 64: ps <= 256)
      const int v = tid & (lw - 1), rsub = tid / lw, rpi = NT / lw;
      const int nrows = cn * rows;
      const int gx = xmin + 4 * v;
      for (int rr0 = rsub; rr0 < nrows; rr0 += rpi * U) {
        float4 val[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int rr = rr0 + u * rpi;
          val[u] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (rr < nrows && v < w4) {
            const int c = static_cast<int>((static_cast<float>(rr) + 0.5f) * inv_rows);   // rr / rows (exact: rr < 16384)
            const int y = rr - c * rows;
            const float* rowp = src + static_cast<size_t>(c) * plane + static_cast<size_t>(min(ymin + y, H - 1)) * W;
            if (gx + 3 < W) {
              val[u] = load_float4_dword_aligned(rowp + gx);
            } else {                              // piece reaching beyond the row: the replicated border column
              val[u] = make_float4(rowp[min(gx, W - 1)], rowp[min(gx + 1, W - 1)], rowp[min(gx + 2, W - 1)], rowp[W - 1]);
            }
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int rr = rr0 + u * rpi;
          if (rr < nrows && v < w4) *reinterpret_cast<float4*>(patch + rr * ps + 4 * v) = val[u];
        }
      }
    }
    __syncthreads();
    if (csub < G) {
      float* o = obase + static_cast<size_t>(cs - c0) * BINS + bin;
#pragma unroll 1
      for (int c = csub; c < cn; c += G) {
#pragma clang fp contract(off)
        const float* p = patch + c * area;
        float acc = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const float* q = p + off[s];
          acc += w1[s] * q[0] + w2[s] * q[1] + w3[s] * q[ps] + w4[s] * q[ps + 1];
        }
        o[static_cast<size_t>(c) * BINS] = acc * inv_count;
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------
// forward, LDS-DMA variant of the fast path: the footprint is copied global -> LDS by
// `global_load_lds_dwordx4` (gfx950: 16 bytes per lane, no staging registers), the whole batch issued
// back to back (one memory round trip per batch instead of one per U loads), and the NEXT batch is
// issued before the current one is consumed (two LDS buffers, one barrier per batch).  LDS-DMA writes
// wave-uniform base + lane * 16, so the patch is lane-linear: float4 index = (channel * rows + y) * w4 + v.
// Pieces that reach beyond a map row (replicated border column) go through registers.
// Arithmetic and operation order are those of roi_align_fwd_lds_kernel: bit-identical output.
// ------------------------------------------------------------------------------------------
// Visiting order for the forward: ROIs sorted by (level, image, 8-row band, column) of their centre.  ROIs arrive
// in score / sampling order, i.e. spatially random; the forward is bound by the fabric traffic of re-fetching
// overlapping footprints (rocprofv3: L2 hit 40 %, TCC misses x 128 B = 600 MB for 342 MB of staged footprints,
// 183 MB of maps).  With neighbours adjacent in the launch — workgroup b runs on XCD b % 8, so the 4 channel chunks
// of ROIs k and k + 2 share an XCD and its L2 — the box-head launch drops from 128 to 96-100 us
// (tools/probe_fwd_order.py).  Rank sort: 8 lanes per ROI count the keys below their own in LDS.  The order is a
// locality heuristic only: ANY permutation gives the same output, bit for bit.
constexpr int kOrderMaxK = 4096;
constexpr int kOrderMinK = 384;
constexpr int kOrderLanes = 16;     // lanes that share one ROI's count
constexpr int kOrderBlock = 1024;   // 16 waves: enough to hide the LDS read latency of the count loop

__device__ __forceinline__ unsigned long long roi_order_key(const Levels& L, const float* __restrict__ rois,
                                                            const int32_t* __restrict__ levels_in, int i) {
  const float* roi = rois + static_cast<size_t>(i) * 5;
  int lvl = 0;
  if (L.num > 1) lvl = levels_in ? levels_in[i] : fpn_level(roi, L);
  float scale = L.lv[0].scale;
#pragma unroll
  for (int l = 1; l < DETOPS_MAX_LEVELS; ++l)
    if (l == lvl) scale = L.lv[l].scale;
  const int b = static_cast<int>(roi[0]);
  const int xc = static_cast<int>((roi[1] + roi[3]) * 0.5f * scale);
  const int yc = static_cast<int>((roi[2] + roi[4]) * 0.0625f * scale);
  const unsigned key = (static_cast<unsigned>(lvl & 7) << 29) | (static_cast<unsigned>(min(max(b, 0), 127)) << 22) |
                       (static_cast<unsigned>(min(max(yc, 0), 1023)) << 12) | static_cast<unsigned>(min(max(xc, 0), 4095));
  return (static_cast<unsigned long long>(key) << 32) | static_cast<unsigned>(i);
}

__global__ void __launch_bounds__(kOrderBlock)
roi_order_kernel(Levels L, const float* __restrict__ rois, const int32_t* __restrict__ levels_in, int K,
                 int32_t* __restrict__ order) {
  __shared__ __attribute__((aligned(16))) unsigned long long keys[kOrderMaxK + 2 * kOrderLanes];
  const int tid = threadIdx.x;
  const int Kp = (K + 2 * kOrderLanes - 1) / (2 * kOrderLanes) * (2 * kOrderLanes);   // padded with +inf keys
  for (int i = tid; i < Kp; i += kOrderBlock)
    keys[i] = i < K ? roi_order_key(L, rois, levels_in, i) : ~0ull;
  __syncthreads();
  const int r = (blockIdx.x * kOrderBlock + tid) / kOrderLanes;
  const int sub = tid & (kOrderLanes - 1);
  int cnt = 0;
  if (r < K) {
    const unsigned long long mine = keys[r];
#pragma unroll 8
    for (int j = 2 * sub; j < Kp; j += 2 * kOrderLanes) {   // one 16-byte LDS read = two keys; same address across ROIs: broadcast
      const unsigned long long k0 = keys[j], k1 = keys[j + 1];
      cnt += (k0 < mine ? 1 : 0) + (k1 < mine ? 1 : 0);
    }
  }
  cnt += __shfl_down(cnt, 8);
  cnt += __shfl_down(cnt, 4);
  cnt += __shfl_down(cnt, 2);
  cnt += __shfl_down(cnt, 1);
  if (r < K && sub == 0) { order[cnt] = r; DETOPS_STAT("fwd.ranked_rois", 1); }
}

#ifdef DETOPS_CPU_EMU
__device__ __forceinline__ void glds16(const float* g, float* lds_wave_base) {
  float* d = lds_wave_base + 4 * (threadIdx.x & 63);
  d[0] = g[0]; d[1] = g[1]; d[2] = g[2]; d[3] = g[3];
}
#else
__device__ __forceinline__ void glds16(const float* g, float* lds_wave_base) {
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  typedef const __attribute__((address_space(1))) void* glb_ptr_t;
  __builtin_amdgcn_global_load_lds((glb_ptr_t)g, (lds_ptr_t)lds_wave_base, 16, 0, 0);
}
#endif

template <int PH, int PW, int SR, int G, int WPS>
__global__ void __launch_bounds__(((PH * PW * G + 63) / 64) * 64, WPS)
roi_align_fwd_dma_kernel(Levels L, const float* __restrict__ rois, const int32_t* __restrict__ levels_in,
                         int32_t* __restrict__ levels_out, float* __restrict__ out, int C, int K, int CT,
                         int chunks, int buf_floats, const int32_t* __restrict__ order) {
  constexpr int BINS = PH * PW;
  constexpr int NS = SR * SR;
  constexpr int NT = ((BINS * G + 63) / 64) * 64;
  DETOPS_DYNAMIC_LDS(float, patch);          // two buffers of buf_floats (multiple of 256 floats = one wave-instruction)
  __shared__ Tap tabY[PH * SR];
  __shared__ Tap tabX[PW * SR];
  __shared__ int s_bounds[4];

  const int tid = threadIdx.x;
  const int bid = blockIdx.x;
  const int kk = bid / chunks;
  const int chunk = bid - kk * chunks;
  const int k = order ? order[kk] : kk;
  const float* roi = rois + static_cast<size_t>(k) * 5;
  int lvl = 0;
  if (L.num > 1) lvl = levels_in ? levels_in[k] : fpn_level(roi, L);
  if (levels_out && chunk == 0 && tid == 0) levels_out[k] = lvl;
  const float* in = L.lv[0].in; int H = L.lv[0].H, W = L.lv[0].W; float scale = L.lv[0].scale;
#pragma unroll
  for (int i = 1; i < DETOPS_MAX_LEVELS; ++i)
    if (i == lvl) { in = L.lv[i].in; H = L.lv[i].H; W = L.lv[i].W; scale = L.lv[i].scale; }

  const RoiGeom g = roi_geometry(roi, scale, PH, PW, SR);
  if (tid == 0) { s_bounds[0] = 0x7fffffff; s_bounds[1] = -1; s_bounds[2] = 0x7fffffff; s_bounds[3] = -1; }
  __syncthreads();
  if (tid < PH * SR + PW * SR) {
    if (tid < PH * SR) {
      const Tap e = axis_entry(g.start_h, g.bin_h, tid / SR, tid % SR, SR, H, 1);
      tabY[tid] = e;
      if (e.l != 0.f || e.h != 0.f) { atomicMin(&s_bounds[0], e.lo); atomicMax(&s_bounds[1], e.lo); }
    } else {
      const int u = tid - PH * SR;
      const Tap e = axis_entry(g.start_w, g.bin_w, u / SR, u % SR, SR, W, 1);
      tabX[u] = e;
      if (e.l != 0.f || e.h != 0.f) { atomicMin(&s_bounds[2], e.lo); atomicMax(&s_bounds[3], e.lo); }
    }
  }
  __syncthreads();
  const int c0 = chunk * CT;
  const int cend = min(C, c0 + CT);
  float* obase = out + (static_cast<size_t>(k) * C + c0) * BINS;
  if (s_bounds[1] < 0 || s_bounds[3] < 0) {  // every sample falls outside the map: all-zero output
    for (int o = tid; o < (cend - c0) * BINS; o += NT) obase[o] = 0.f;
    return;
  }
  const int ymin = s_bounds[0];
  const int rows = s_bounds[1] - ymin + 2;   // (lo range) + the lo+1 row
  const size_t plane = static_cast<size_t>(H) * W;
  const float* base = in + (static_cast<size_t>(g.b) * C + c0) * plane;
  const int xmin = s_bounds[2];
  int w4 = (s_bounds[3] - xmin + 2 + 3) >> 2;   // float4 pieces per patch row incl. the lo+1 column
  w4 |= 1;                                       // odd: rows spread over the LDS banks
  const int ps = 4 * w4;
  const int a4 = rows * w4;                      // float4 pieces per channel
  const int area = 4 * a4;

  if (area > buf_floats) {  // footprint too large for LDS: gather straight from the map
    for (int o = tid; o < (cend - c0) * BINS; o += NT) {
#pragma clang fp contract(off)
      const int cl = o / BINS;
      const int bin = o - cl * BINS;
      const int ph = bin / PW, pw = bin - ph * PW;
      const float* d = base + static_cast<size_t>(cl) * plane;
      float acc = 0.f;
      for (int iy = 0; iy < SR; ++iy) {
        const Tap ty = tabY[ph * SR + iy];
        const float* r0 = d + ty.lo * W;
        const float* r1 = d + ty.hi * W;
        for (int ix = 0; ix < SR; ++ix) {
          const Tap tx = tabX[pw * SR + ix];
          const float w1 = ty.h * tx.h, w2 = ty.h * tx.l, w3 = ty.l * tx.h, w4_ = ty.l * tx.l;
          acc += w1 * r0[tx.lo] + w2 * r0[tx.hi] + w3 * r1[tx.lo] + w4_ * r1[tx.hi];
        }
      }
      obase[o] = acc / g.count;
    }
    return;
  }

  // this lane's piece of a staging pass: piece index tid -> (channel, row, column piece); a pass advances by NT
  const float inv_a4 = 1.f / static_cast<float>(a4), inv_w4 = 1.f / static_cast<float>(w4);
  const int pc0 = static_cast<int>((static_cast<float>(tid) + 0.5f) * inv_a4);
  const int prem = tid - pc0 * a4;
  const int py0 = static_cast<int>((static_cast<float>(prem) + 0.5f) * inv_w4);
  const int pv0 = prem - py0 * w4;
  const int dc = static_cast<int>((static_cast<float>(NT) + 0.5f) * inv_a4);
  const int drem = NT - dc * a4;
  const int dy = static_cast<int>((static_cast<float>(drem) + 0.5f) * inv_w4);
  const int dv = drem - dy * w4;
  const int wave_off = __builtin_amdgcn_readfirstlane(tid >> 6) * 256;   // floats

  // channels per batch: as many as fit a buffer, then evened out over the batches (a short last batch costs a full
  // barrier + memory round trip: 2 x 13 KB buffers measured 93 us where the evenly filled 2 x 11 KB took 84)
  const int ctb_max = min(cend - c0, buf_floats / area);
  const int nbatch = (cend - c0 + ctb_max - 1) / ctb_max;
  const int ctb = (cend - c0 + nbatch - 1) / nbatch;
  // element offset of this lane's piece inside the batch's source planes, advanced incrementally per pass (the
  // from-scratch form costs ~25 VALU instructions per 16 bytes: 64-bit multiplies for channel and row)
  const int go0 = pc0 * static_cast<int>(plane) + (ymin + py0) * W + xmin + 4 * pv0;
  const int dgo = dc * static_cast<int>(plane) + dy * W + 4 * dv;
  const int wrap_v = W - 4 * w4;                          // (y + 1, v - w4)
  const int wrap_y = static_cast<int>(plane) - rows * W;  // (c + 1, y - rows)
  const bool planes_fit = static_cast<size_t>(ctb + 1) * plane < 0x7fffffffu;
  auto issue = [&](int cs, int cn, float* buf) {
    const float* src = base + static_cast<size_t>(cs - c0) * plane;
    const int total = cn * a4;
    int y = py0, v = pv0, go = go0;
    int c = pc0;
#pragma unroll 1
    for (int i0 = 0; i0 < total; i0 += NT) {
      if (i0 + tid < total) {
        const int gx = xmin + 4 * v;
        if (gx + 3 < W && ymin + y < H && planes_fit) {
          glds16(src + go, buf + 4 * i0 + wave_off);
        } else {                              // piece reaching beyond the map: replicated border column / row
          const float* rowp = src + static_cast<size_t>(c) * plane + static_cast<size_t>(min(ymin + y, H - 1)) * W;
          *reinterpret_cast<float4*>(buf + 4 * (i0 + tid)) =
              make_float4(rowp[min(gx, W - 1)], rowp[min(gx + 1, W - 1)], rowp[min(gx + 2, W - 1)], rowp[min(gx + 3, W - 1)]);
        }
      }
      go += dgo; v += dv; y += dy; c += dc;
      if (v >= w4) { v -= w4; ++y; go += wrap_v; }
      if (y >= rows) { y -= rows; ++c; go += wrap_y; }
    }
  };

  // per-thread sample geometry (registers)
  const int bin = tid % BINS;
  const int csub = tid / BINS;  // >= G: idle lane of the last wave
  const int ph = bin / PW, pw = bin - ph * PW;
  int off[NS];
  float w1[NS], w2[NS], w3[NS], w4s[NS];
#pragma unroll
  for (int iy = 0; iy < SR; ++iy) {
    const Tap ty = tabY[ph * SR + iy];
    const bool vy = (ty.l != 0.f || ty.h != 0.f);
#pragma unroll
    for (int ix = 0; ix < SR; ++ix) {
#pragma clang fp contract(off)
      const Tap tx = tabX[pw * SR + ix];
      const bool vv = vy && (tx.l != 0.f || tx.h != 0.f);
      const int s = iy * SR + ix;
      off[s] = vv ? (ty.lo - ymin) * ps + (tx.lo - xmin) : 0;
      w1[s] = vv ? ty.h * tx.h : 0.f;
      w2[s] = vv ? ty.h * tx.l : 0.f;
      w3[s] = vv ? ty.l * tx.h : 0.f;
      w4s[s] = vv ? ty.l * tx.l : 0.f;
    }
  }
  const float inv_count = 1.f / static_cast<float>(NS);  // NS in {1,4}: exact reciprocal

  issue(c0, min(ctb, cend - c0), patch);
  int n = 0;
  for (int cs = c0; cs < cend; cs += ctb, ++n) {
    const int cn = min(ctb, cend - cs);
    if (tid == 0) { DETOPS_STAT("fwd.stage_batches", 1); DETOPS_STAT("fwd.staged_floats", cn * area); }
    __syncthreads();   // batch n has landed (vmcnt(0) + barrier); every wave is done reading the other buffer
    const float* cur = patch + (n & 1) * buf_floats;
    if (cs + ctb < cend) issue(cs + ctb, min(ctb, cend - cs - ctb), patch + ((n + 1) & 1) * buf_floats);
    if (csub < G) {
      float* o = obase + static_cast<size_t>(cs - c0) * BINS + bin;
#pragma unroll 1
      for (int c = csub; c < cn; c += G) {
#pragma clang fp contract(off)
        const float* p = cur + c * area;
        float t0[NS], t1[NS], t2[NS], t3[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {     // all taps of the bin in flight before the first use
          const float* q = p + off[s];
          t0[s] = q[0]; t1[s] = q[1]; t2[s] = q[ps]; t3[s] = q[ps + 1];
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) DETOPS_PIN4(t0[s], t1[s], t2[s], t3[s]);
        float acc = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) acc += w1[s] * t0[s] + w2[s] * t1[s] + w3[s] * t2[s] + w4s[s] * t3[s];
        o[static_cast<size_t>(c) * BINS] = acc * inv_count;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// backward, tile-centric LDS scatter (FALLBACK: bin counts beyond the gather kernel's LDS plan, and
// A/B runs with DETOPS_ROIALIGN_BWD=tile): no global atomics, no separate zero-fill.
//
// The gradient maps are cut into TH x TW pixel tiles; one workgroup owns (level, image, tile,
// CT channels), accumulates the contribution of EVERY ROI that touches its tile into an LDS
// accumulator (ds_add_f32), and finally stores the tile with plain row-contiguous stores.  Each
// gradient-map element is therefore written exactly once (tiles no ROI touches store zeros — the
// zero-fill the reference gets from at::zeros, ROIAlign_cuda.cu:316, comes for free), and the
// ~130 M global atomics / launch of the ROI-centric formulation (16 per gradient element in the
// reference, ROIAlign_cuda.cu:246-249) disappear.  The price is redundant tap evaluation for ROIs
// that straddle tiles (~2.3x at 32 x 64 tiles for box-head ROIs), paid in LDS/VALU cycles that are
// not the bottleneck.  ROIs are visited in ascending index order (ballot compaction).
// ------------------------------------------------------------------------------------------
constexpr int kAccFloats = 8400;  // ~33 KiB LDS accumulator: CT * (TH * (TW + 1) + 3) <= kAccFloats
constexpr int kBwdList = 256;     // ROIs tested per scan round (one per thread)
constexpr int kBwdMaxCT = 4;      // channels per workgroup (register-resident gradient values)

struct BwdPlan {
  int first_item[DETOPS_MAX_LEVELS + 1];  // workgroup-id prefix per level
  int tiles_x[DETOPS_MAX_LEVELS], tiles_y[DETOPS_MAX_LEVELS];
  int TH, TW, CT, chunks, accumulate;
};

// Geometry of a ROI that touches the tile, parked in LDS for the unit loop.
struct HitGeom {
  float start_w, start_h, bin_w, bin_h;
  int k, gh, gw;
};

template <int PH_, int PW_, int SR_>
__global__ void __launch_bounds__(kBlock)
roi_align_bwd_tile_kernel(Levels L, BwdPlan P, const float* __restrict__ rois,
                          const int32_t* __restrict__ levels_in, const float* __restrict__ gout, int C,
                          int K, int PHr, int PWr, int sr) {
  const int PH = PH_ ? PH_ : PHr;
  const int PW = PW_ ? PW_ : PWr;
  const int bins = PH * PW;
  DETOPS_DYNAMIC_LDS(float, acc);
  __shared__ HitGeom s_hit[kBwdList];
  __shared__ int s_wcount[kBlock / kWave];

  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  // ---- decode the work item
  int lvl = 0;
#pragma unroll
  for (int i = 1; i < DETOPS_MAX_LEVELS; ++i)
    if (i < L.num && static_cast<int>(blockIdx.x) >= P.first_item[i]) lvl = i;
  float* gin = L.lv[0].gin; int H = L.lv[0].H, W = L.lv[0].W; float scale = L.lv[0].scale;
  int ntx = P.tiles_x[0], nty = P.tiles_y[0];
#pragma unroll
  for (int i = 1; i < DETOPS_MAX_LEVELS; ++i)
    if (i == lvl) { gin = L.lv[i].gin; H = L.lv[i].H; W = L.lv[i].W; scale = L.lv[i].scale;
                    ntx = P.tiles_x[i]; nty = P.tiles_y[i]; }
  int rem = static_cast<int>(blockIdx.x) - P.first_item[lvl];
  const int chunk = rem % P.chunks; rem /= P.chunks;
  const int tix = rem % ntx; rem /= ntx;
  const int tiy = rem % nty;
  const int b = rem / nty;
  const int TH = P.TH, TW = P.TW;
  const int y0 = tiy * TH, x0 = tix * TW;
  const int y1 = min(H, y0 + TH) - 1, x1 = min(W, x0 + TW) - 1;  // inclusive
  const int c0 = chunk * P.CT;
  const int cn = min(P.CT, C - c0);
  // LDS accumulator [cn][TH][TW+1] (+3 per plane): the odd row stride spreads the taps of bins that
  // share a column (same x, rows a few pixels apart) over different banks — with a 64-dword stride
  // every such ds_add_f32 of a wavefront would hit ONE bank (7-9-way serialisation, measured 5x)
  const int TWP = TW + 1;
  const int pstride = TH * TWP + 3;

  for (int e = tid; e < cn * pstride; e += kBlock) acc[e] = 0.f;
  // (the first __syncthreads below orders the clear before any accumulation)

  for (int kb = 0; kb < K; kb += kBwdList) {
    // ---- which of ROIs [kb, kb+256) touch this tile?  ordered compaction into s_hit
    const int r = kb + tid;
    bool hit = false;
    RoiGeom g{};
    if (r < K) {
      const float* roi = rois + static_cast<size_t>(r) * 5;
      const int rl = (L.num > 1) ? levels_in[r] : 0;
      if (rl == lvl && static_cast<int>(roi[0]) == b) {
#pragma clang fp contract(off)
        g = roi_geometry(roi, scale, PH, PW, SR_ ? SR_ : sr);
        const float rh = g.bin_h * static_cast<float>(PH), rw = g.bin_w * static_cast<float>(PW);
        // rows/cols any tap of this ROI can touch (conservative): floor(first) .. floor(last)+1
        const float fy0 = floorf(fmaxf(g.start_h, 0.f)), fy1 = floorf(g.start_h + rh) + 2.f;
        const float fx0 = floorf(fmaxf(g.start_w, 0.f)), fx1 = floorf(g.start_w + rw) + 2.f;
        hit = fy0 <= static_cast<float>(y1) && fy1 >= static_cast<float>(y0) &&
              fx0 <= static_cast<float>(x1) && fx1 >= static_cast<float>(x0);
      }
    }
    const unsigned long long m = __ballot(hit);
    if (lane == 0) s_wcount[wave] = __popcll(m);
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int j = 0; j < kBlock / kWave; ++j) {
      const int v = s_wcount[j];
      if (j < wave) before += v;
      total += v;
    }
    if (hit) {
      HitGeom h;
      h.start_w = g.start_w; h.start_h = g.start_h; h.bin_w = g.bin_w; h.bin_h = g.bin_h;
      h.k = r; h.gh = g.gh; h.gw = g.gw;
      s_hit[before + __popcll(m & ((1ull << lane) - 1ull))] = h;
    }
    __syncthreads();

    // ---- flat loop over (hit ROI, bin) pairs: no per-ROI barriers.  A lane derives its bin's taps
    //      once and applies them to all (<= kBwdMaxCT) channels of the chunk.
    const int pairs = total * bins;
    for (int q = tid; q < pairs; q += kBlock) {
#pragma clang fp contract(off)
      const int li = q / bins;
      const int bin = q - li * bins;
      const int ph = bin / PW;
      const int pw = bin - ph * PW;
      const HitGeom h = s_hit[li];
      const float* go = gout + (static_cast<size_t>(h.k) * C + c0) * bins + bin;
      float gv[kBwdMaxCT];
#pragma unroll
      for (int cl = 0; cl < kBwdMaxCT; ++cl) gv[cl] = cl < cn ? go[cl * bins] : 0.f;
      const int gh = SR_ ? SR_ : h.gh, gw = SR_ ? SR_ : h.gw;
      const int icount = gh * gw;
      const float count = static_cast<float>(icount);
      const bool pow2 = (icount & (icount - 1)) == 0;
      const float inv_count = 1.f / count;
      for (int iy = 0; iy < gh; ++iy) {
        const Tap ty = axis_entry(h.start_h, h.bin_h, ph, iy, gh, H, 1);
        if ((ty.l == 0.f && ty.h == 0.f) || ty.hi < y0 || ty.lo > y1) continue;
        const bool in0 = ty.lo >= y0, in1 = ty.hi <= y1;
        const int r0 = (ty.lo - y0) * TWP - x0, r1 = (ty.hi - y0) * TWP - x0;
        for (int ix = 0; ix < gw; ++ix) {
          const Tap tx = axis_entry(h.start_w, h.bin_w, pw, ix, gw, W, 1);
          if ((tx.l == 0.f && tx.h == 0.f) || tx.hi < x0 || tx.lo > x1) continue;
          const bool jn0 = tx.lo >= x0, jn1 = tx.hi <= x1;
          const float w1 = ty.h * tx.h, w2 = ty.h * tx.l, w3 = ty.l * tx.h, w4 = ty.l * tx.l;
#pragma unroll
          for (int cl = 0; cl < kBwdMaxCT; ++cl) {
            if (cl < cn) {
              const float gval = gv[cl];
              float g1 = gval * w1, g2 = gval * w2, g3 = gval * w3, g4 = gval * w4;
              if (pow2) { g1 *= inv_count; g2 *= inv_count; g3 *= inv_count; g4 *= inv_count; }
              else { g1 /= count; g2 /= count; g3 /= count; g4 /= count; }
              float* a = acc + cl * pstride;
              if (in0 && jn0) atomicAdd(a + r0 + tx.lo, g1);
              if (in0 && jn1) atomicAdd(a + r0 + tx.hi, g2);
              if (in1 && jn0) atomicAdd(a + r1 + tx.lo, g3);
              if (in1 && jn1) atomicAdd(a + r1 + tx.hi, g4);
            }
          }
        }
      }
    }
    __syncthreads();  // s_hit / s_wcount are rewritten by the next round
  }

  // ---- store the tile (row-contiguous); every in-map element of the tile is written exactly once
  const size_t plane = static_cast<size_t>(H) * W;
  float* gb = gin + (static_cast<size_t>(b) * C + c0) * plane;
  const int tw_valid = x1 - x0 + 1, th_valid = y1 - y0 + 1;
  const int tarea = TH * TW;
  for (int e = tid; e < cn * tarea; e += kBlock) {
    const int cl = e / tarea;
    const int rr = e - cl * tarea;
    const int y = rr / TW;
    const int x = rr - y * TW;
    if (y < th_valid && x < tw_valid) {
      float* dst = gb + static_cast<size_t>(cl) * plane + static_cast<size_t>(y0 + y) * W + (x0 + x);
      float v = acc[cl * pstride + y * TWP + x];
      if (P.accumulate) v += *dst;
      *dst = v;
    }
  }
}

// ------------------------------------------------------------------------------------------
// backward, pixel-owner ("adjoint gather"): NO atomics of any kind, deterministic.
//
// ROIAlign's sampling is separable, so its adjoint factors per ROI r and channel c into two tiny
// matrices:
//     grad_in[b_r, c, y, x] += sum_{ph,pw} AY_r[ph, y] * AX_r[pw, x] * grad_out[r, c, ph, pw]
//     AY_r[ph, y] = (1/gh) * sum_iy ([y == lo(ph,iy)] * hy + [y == hi(ph,iy)] * ly)     (AX alike)
// with lo/hi/ly/hy exactly the reference's sample taps (ROIAlign_cuda.cu:125-175).  A pixel only
// receives from bins whose samples lie within one pixel of it, so AY[., y] / AX[., x] have 2-3
// non-zeros for model-sized ROIs.
//
// A workgroup owns an 8 x 32 pixel tile of one gradient map and CT channels; every thread owns ONE
// pixel and keeps its CT partial sums in registers from the first ROI to the last, so nothing is
// ever accumulated in memory: no global atomics (205 M per launch in the scatter form; the
// reference issues 16 per gradient element, ROIAlign_cuda.cu:246-249), no LDS atomics (the
// tile-scatter kernel above is bound by same-address ds_add_f32 serialisation: 1.46 ms measured),
// and no zero-fill pass — each gradient-map element is written exactly once, as full 128-byte
// rows.  ROIs are visited in ascending index order and each pixel's sum is formed sequentially,
// so the result is bit-reproducible run to run (the reference's atomics are not).
//
// Per batch of hit ROIs the workgroup (1) builds the dense per-axis coefficient rows for its tile
// in LDS (one thread per (ROI, axis, bin), reference tap arithmetic), (2) stages
// grad_out[r, c0:c0+CT, :, :] transposed to [bin][channel] so that (3) a wave walks the (ph, pw)
// pairs any of its lanes needs (wave-uniform ballots skip the rest): per pair one broadcast
// ds_read_b128 per 4 channels + CT FMAs.  Waves are 8 x 8 pixel blocks: a ROI that misses a
// wave's columns is skipped by that wave.
// ------------------------------------------------------------------------------------------
constexpr int kGTH = 8, kGTW = 32;     // tile: 8 rows x 32 columns = 256 threads
constexpr int kGBins = 256;            // (ROIs per batch) * bins <= kGBins when bins <= kGBins
constexpr int kGRowPad = kGTW + 1;     // transposition buffer row stride (bank spread)

struct GPlan {
  int first_item[DETOPS_MAX_LEVELS];   // workgroup-id offset per level (coarsest level first)
  int n_items[DETOPS_MAX_LEVELS];
  int tiles_x[DETOPS_MAX_LEVELS], tiles_y[DETOPS_MAX_LEVELS];
  int chunks, accumulate, batch;       // batch = ROIs staged per round
  int groups;                          // > 1: the ROI list is split over blockIdx.y (small maps), see below
};

struct __align__(16) GHit {   // 32 bytes
  float start_w, start_h, bin_w, bin_h;
  int k, gh, gw;
  int xspan;   // (ix0 << 16) | ix1: columns the ROI's taps can reach (conservative), map width < 32768
};

template <int PH_, int PW_, int CT>
__global__ void __launch_bounds__(kBlock)
roi_align_bwd_gather_kernel(Levels L, GPlan P, const float* __restrict__ rois,
                            const int32_t* __restrict__ levels_in, const float* __restrict__ gout,
                            int C, int K, int PHr, int PWr, int sr) {
  static_assert(CT % 4 == 0, "channels are staged as float4 groups");
  constexpr int CG = CT / 4;
  const int PH = PH_ ? PH_ : PHr;
  const int PW = PW_ ? PW_ : PWr;
  const int bins = PH * PW;
  const int PPH = (PH + 3) & ~3, PPW = (PW + 3) & ~3;   // coefficient rows padded to float4
  const int slots = max(bins, kGBins);                   // (ROI, bin) slots per float4 channel group

  DETOPS_DYNAMIC_LDS(float, g_lds);
  float4* gs4 = reinterpret_cast<float4*>(g_lds);        // [CG][slots] float4   (also the store buffer)
  const int region = max(slots * CT, CT * kGTH * kGRowPad);
  float* ayt = g_lds + region;                           // [batch][kGTH][PPH]
  float* axt = ayt + P.batch * kGTH * PPH;               // [batch][kGTW][PPW]
  __shared__ GHit s_hit[kBlock];
  __shared__ int s_wcount[kBlock / kWave];

  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  // ---- decode the work item (levels are laid out coarsest first: their tiles see the most ROIs)
  int lvl = 0;
#pragma unroll
  for (int i = 1; i < DETOPS_MAX_LEVELS; ++i)
    if (i < L.num && static_cast<int>(blockIdx.x) >= P.first_item[i] &&
        static_cast<int>(blockIdx.x) < P.first_item[i] + P.n_items[i]) lvl = i;
  float* gin = L.lv[0].gin; int H = L.lv[0].H, W = L.lv[0].W; float scale = L.lv[0].scale;
  int ntx = P.tiles_x[0], nty = P.tiles_y[0], first = P.first_item[0];
#pragma unroll
  for (int i = 1; i < DETOPS_MAX_LEVELS; ++i)
    if (i == lvl) { gin = L.lv[i].gin; H = L.lv[i].H; W = L.lv[i].W; scale = L.lv[i].scale;
                    ntx = P.tiles_x[i]; nty = P.tiles_y[i]; first = P.first_item[i]; }
  int rem = static_cast<int>(blockIdx.x) - first;
  const int chunk = rem % P.chunks; rem /= P.chunks;
  const int tix = rem % ntx; rem /= ntx;
  const int tiy = rem % nty;
  const int b = rem / nty;
  const int y0 = tiy * kGTH, x0 = tix * kGTW;
  const int y1 = min(H, y0 + kGTH) - 1, x1 = min(W, x0 + kGTW) - 1;  // inclusive
  const int c0 = chunk * CT;
  // this thread's pixel: waves are 8x8 blocks side by side
  const int yl = lane >> 3, xl = wave * 8 + (lane & 7);
  const int wx0 = x0 + wave * 8, wx1 = wx0 + 7;

  float acc[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) acc[c] = 0.f;

  // Small maps (e.g. the 14 x 14 cfg-1 map: 2 tiles) leave the chip idle and make one workgroup walk
  // hundreds of ROIs.  There the ROI list is split over blockIdx.y; each group adds its partial
  // sums with global atomics into the pre-zeroed (tiny) map — the only configuration with atomics.
  const int r_begin = static_cast<int>(static_cast<int64_t>(K) * blockIdx.y / P.groups);
  const int r_end = static_cast<int>(static_cast<int64_t>(K) * (blockIdx.y + 1) / P.groups);
  for (int kb = r_begin; kb < r_end; kb += kBlock) {
    // ---- which of ROIs [kb, kb+256) touch this tile?  ordered compaction into s_hit
    const int r = kb + tid;
    bool hit = false;
    GHit h{};
    if (r < r_end) {
      const float* roi = rois + static_cast<size_t>(r) * 5;
      const int rl = (L.num > 1) ? levels_in[r] : 0;
      if (rl == lvl && static_cast<int>(roi[0]) == b) {
#pragma clang fp contract(off)
        const RoiGeom g = roi_geometry(roi, scale, PH, PW, sr);
        const float rh = g.bin_h * static_cast<float>(PH), rw = g.bin_w * static_cast<float>(PW);
        // rows/cols any tap of this ROI can touch (conservative): floor(first) .. floor(last)+2
        const float fy0 = floorf(fmaxf(g.start_h, 0.f)), fy1 = floorf(g.start_h + rh) + 2.f;
        const float fx0 = floorf(fmaxf(g.start_w, 0.f)), fx1 = floorf(g.start_w + rw) + 2.f;
        hit = fy0 <= static_cast<float>(y1) && fy1 >= static_cast<float>(y0) &&
              fx0 <= static_cast<float>(x1) && fx1 >= static_cast<float>(x0);
        h.start_w = g.start_w; h.start_h = g.start_h; h.bin_w = g.bin_w; h.bin_h = g.bin_h;
        h.k = r; h.gh = g.gh; h.gw = g.gw;
        const int ix0 = static_cast<int>(fminf(fx0, static_cast<float>(W)));
        const int ix1 = static_cast<int>(fminf(fmaxf(fx1, 0.f), static_cast<float>(W)));
        h.xspan = (ix0 << 16) | ix1;
      }
    }
    const unsigned long long m = __ballot(hit);
    if (lane == 0) s_wcount[wave] = __popcll(m);
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int j = 0; j < kBlock / kWave; ++j) {
      const int v = s_wcount[j];
      if (j < wave) before += v;
      total += v;
    }
    if (hit) s_hit[before + __popcll(m & ((1ull << lane) - 1ull))] = h;
    __syncthreads();
    if (tid == 0) { DETOPS_STAT("bwd.scan_rounds", 1); DETOPS_STAT("bwd.hits", total); }

    for (int h0 = 0; h0 < total; h0 += P.batch) {
      const int nb = min(P.batch, total - h0);
      if (tid == 0) DETOPS_STAT("bwd.batches", 1);
      // ---- (1) per-axis coefficient rows of the batch, restricted to this tile
      for (int t = tid; t < nb * (PH + PW); t += kBlock) {
        const int j = t / (PH + PW);
        const int q = t - j * (PH + PW);
        const GHit hj = s_hit[h0 + j];
        if (q < PH) {
          float* row = ayt + j * kGTH * PPH + q;
          for (int i = 0; i < kGTH; ++i) row[i * PPH] = 0.f;
          const float inv = 1.f / static_cast<float>(hj.gh);
          for (int i = 0; i < hj.gh; ++i) {
            const Tap e = axis_entry(hj.start_h, hj.bin_h, q, i, hj.gh, H, 1);
            const int a0 = e.lo - y0, a1 = e.hi - y0;
            if (a0 >= 0 && a0 < kGTH) row[a0 * PPH] += e.h * inv;
            if (a1 >= 0 && a1 < kGTH) row[a1 * PPH] += e.l * inv;
          }
        } else {
          const int qq = q - PH;
          float* row = axt + j * kGTW * PPW + qq;
          for (int i = 0; i < kGTW; ++i) row[i * PPW] = 0.f;
          const float inv = 1.f / static_cast<float>(hj.gw);
          for (int i = 0; i < hj.gw; ++i) {
            const Tap e = axis_entry(hj.start_w, hj.bin_w, qq, i, hj.gw, W, 1);
            const int a0 = e.lo - x0, a1 = e.hi - x0;
            if (a0 >= 0 && a0 < kGTW) row[a0 * PPW] += e.h * inv;
            if (a1 >= 0 && a1 < kGTW) row[a1 * PPW] += e.l * inv;
          }
        }
      }
      // ---- (2) stage grad_out[r, c0:c0+CT, :, :] as float4 channel groups: [cg][j*bins + bin]
      //      (global reads run along a channel's contiguous bins; LDS writes are 16-byte, lane-contiguous)
      for (int u = tid; u < nb * bins * CG; u += kBlock) {
        const int cg = u / (nb * bins);
        const int jb = u - cg * (nb * bins);
        const int j = jb / bins;
        const int bin = jb - j * bins;
        const int cbase = c0 + cg * 4;
        const float* src = gout + (static_cast<size_t>(s_hit[h0 + j].k) * C + cbase) * bins + bin;
        float4 v;
        v.x = (cbase + 0 < C) ? src[0] : 0.f;
        v.y = (cbase + 1 < C) ? src[bins] : 0.f;
        v.z = (cbase + 2 < C) ? src[2 * bins] : 0.f;
        v.w = (cbase + 3 < C) ? src[3 * bins] : 0.f;
        gs4[cg * slots + jb] = v;
      }
      __syncthreads();
      // ---- (3) every pixel gathers from the bins that reach it
      for (int j = 0; j < nb; ++j) {
        const int xspan = s_hit[h0 + j].xspan;
        const int jx0 = xspan >> 16, jx1 = xspan & 0xffff;
        if (jx1 < wx0 || jx0 > wx1) continue;   // this ROI misses the wave's 8 columns
        if (lane == 0) DETOPS_STAT("bwd.wave_roi_tasks", 1);
        const float* ayr = ayt + (j * kGTH + yl) * PPH;
        const float* axr = axt + (j * kGTW + xl) * PPW;
        const float4* gj = gs4 + j * bins;
        // every lane walks its OWN contiguous range of contributing bins (the bins with a sample
        // within one pixel of it: 2-4 per axis for model-sized ROIs); trip counts are the wave maxima,
        // gradient reads are per-lane ds_read_b128.  Measured against the union-of-the-wave walk it
        // replaced (profiles/r02a_opbench_experimental_ab.log): box head 217 -> 189 us, mask head
        // 211 -> 147 us, bit-identical sums.
        int ylo = PH, yhi = -1, xlo = PW, xhi = -1;
        for (int ph = 0; ph < PH; ++ph) if (ayr[ph] != 0.f) { ylo = min(ylo, ph); yhi = ph; }
        for (int pw = 0; pw < PW; ++pw) if (axr[pw] != 0.f) { xlo = min(xlo, pw); xhi = pw; }
        const int ny = yhi - ylo + 1, nx = xhi - xlo + 1;   // <= 0: nothing reaches this pixel
        int na = 0, nb_ = 0;
        while (__ballot(na < ny) != 0ull) ++na;
        while (__ballot(nb_ < nx) != 0ull) ++nb_;
        for (int a = 0; a < na; ++a) {
          const int ph = min(ylo + a, PH - 1);
          const float wy = (a < ny) ? ayr[ph] : 0.f;
          for (int b2 = 0; b2 < nb_; ++b2) {
            if (lane == 0) DETOPS_STAT("bwd.bodies_lane_walk", 1);
            const int pw = min(xlo + b2, PW - 1);
            const float w = (b2 < nx) ? wy * axr[pw] : 0.f;
            if (w != 0.f) {
              DETOPS_STAT("bwd.active_lane_bodies", 1);
              const float4* gp = gj + ph * PW + pw;
#pragma unroll
              for (int cg = 0; cg < CG; ++cg) {
                const float4 g4 = gp[cg * slots];
                acc[4 * cg + 0] = fmaf(w, g4.x, acc[4 * cg + 0]);
                acc[4 * cg + 1] = fmaf(w, g4.y, acc[4 * cg + 1]);
                acc[4 * cg + 2] = fmaf(w, g4.z, acc[4 * cg + 2]);
                acc[4 * cg + 3] = fmaf(w, g4.w, acc[4 * cg + 3]);
              }
            }
          }
        }
      }
      __syncthreads();  // the next batch (or scan round, or the store) rewrites the staging region
    }
  }

  if (tid == 0) DETOPS_STAT("bwd.workgroups", 1);
  // ---- store: registers -> LDS [c][8][33] -> full 128-byte rows; every in-map element of the
  //      tile is written exactly once (zeros where no ROI reaches)
  float* tb = g_lds;
#pragma unroll
  for (int c = 0; c < CT; ++c) tb[(c * kGTH + yl) * kGRowPad + xl] = acc[c];
  __syncthreads();
  const size_t plane = static_cast<size_t>(H) * W;
  const int cn = min(CT, C - c0);
  float* gb = gin + (static_cast<size_t>(b) * C + c0) * plane;
  for (int e = tid; e < cn * kGTH * kGTW; e += kBlock) {
    const int c = e / (kGTH * kGTW);
    const int pix = e - c * (kGTH * kGTW);
    const int yy = pix / kGTW, xx = pix - yy * kGTW;
    if (y0 + yy <= y1 && x0 + xx <= x1) {
      float* dst = gb + static_cast<size_t>(c) * plane + static_cast<size_t>(y0 + yy) * W + (x0 + xx);
      float v = tb[(c * kGTH + yy) * kGRowPad + xx];
      if (P.groups > 1) {
        if (v != 0.f) atomicAdd(dst, v);
      } else {
        if (P.accumulate) v += *dst;
        *dst = v;
      }
    }
  }
}

// Channel chunk per workgroup: enough workgroups to fill 256 CUs several times over while
// keeping the per-workgroup table build amortised over >= 16 channels.
inline int pick_chunk(int C, int K) {
  int CT = 64;
  while (CT > 16 && static_cast<int64_t>(K) * ceil_div64(C, CT) < 4 * kNumCU) CT >>= 1;
  if (CT > C) CT = C;
  return CT;
}

template <int V> using IC = std::integral_constant<int, V>;

template <typename F>
inline void dispatch_shape(int PH, int PW, int sr, F&& f) {
  const bool small = sr > 0 && PH * sr <= kTabSmall && PW * sr <= kTabSmall;
  if (PH == 7 && PW == 7 && small) f(IC<7>{}, IC<7>{}, IC<kTabSmall>{});
  else if (PH == 14 && PW == 14 && small) f(IC<14>{}, IC<14>{}, IC<kTabSmall>{});
  else if (PH == 7 && PW == 7) f(IC<7>{}, IC<7>{}, IC<kTabBig>{});
  else if (PH == 14 && PW == 14) f(IC<14>{}, IC<14>{}, IC<kTabBig>{});
  else if (small) f(IC<0>{}, IC<0>{}, IC<kTabSmall>{});
  else f(IC<0>{}, IC<0>{}, IC<kTabBig>{});
}

template <int PH, int PW, int SR, int G>
void launch_fwd_lds(const Levels& L, const float* rois, const int32_t* levels_in, int32_t* levels_out,
                    float* out, int C, int K, hipStream_t st, int32_t* order_ws) {
  constexpr int NT = ((PH * PW * G + 63) / 64) * 64;
  // channels per workgroup.  32: with C = 256 the 8 chunks of a ROI land on the 8 XCDs (workgroup b runs on XCD
  // b % 8), so every L2 caches ONE 32-channel slice of the maps and sees all ROIs in ranked order; and 8 K short
  // workgroups balance better over 256 x 7 slots than 4 K long ones (tools/probe_fwd_knobs.py: 84 vs 100 us)
  const char* fwd_env = getenv("DETOPS_ROIALIGN_FWD");
  const bool reg_staged = fwd_env && fwd_env[0] == 'l';   // DETOPS_ROIALIGN_FWD=lds: register-staged kernel (A/B runs)
  int64_t map_pixels = 0;
  for (int i = 0; i < L.num; ++i) map_pixels += static_cast<int64_t>(L.lv[i].H) * L.lv[i].W;
  // small maps = small footprints: the per-workgroup setup dominates, fewer and fatter workgroups win (cfg-1: 63 vs 86 us)
  int CT = (reg_staged || map_pixels < 16384) ? 64 : 32;
  while (CT > 16 && static_cast<int64_t>(K) * ceil_div64(C, CT) < 4 * kNumCU) CT >>= 1;
  if (const char* e = getenv("DETOPS_ROIALIGN_FWD_CT")) CT = max(1, atoi(e));   // A/B runs
  if (CT > C) CT = C;
  const int chunks = static_cast<int>(ceil_div64(C, CT));
  // measured (profiles/r01d_opbench.log): 7x7 bins are fastest at 32 KiB / 4 loads in flight (142 us box
  // head); the larger 14x14 footprints at 48 KiB / 8 (79 vs 87 us mask head)
  int patch_floats = kLdsPatchFloats, unroll = 4;
  if (PH * PW >= 196) { patch_floats = 48 * 256 - 64; unroll = 8; }
  if (const char* e = getenv("DETOPS_ROIALIGN_FWD_LDS_KB")) {
    const int kb = atoi(e);
    if (kb >= 4 && kb <= 60) patch_floats = kb * 256 - 64;
  }
  if (const char* e = getenv("DETOPS_ROIALIGN_FWD_U")) unroll = atoi(e);
  const dim3 grid(static_cast<unsigned>(K) * chunks);
  if (!reg_staged) {
    // two buffers, each a whole number of 1 KiB wave-instructions, sized so that LDS allows as many workgroups per
    // CU as the registers do (68 VGPRs -> 7 waves / SIMD = 28 waves / CU): 7 x (2 x 11 KB) for the 4-wave 7x7
    // kernel, 4 x (2 x 19 KB) for the 7-wave 14x14 kernel — the measured optimum of both sweeps
    const int wgs = max(1, 28 / (NT / 64));
    int buf_floats = max(4, (160 * 1024 / wgs - 768) / 2048) * 256;
    if (const char* e = getenv("DETOPS_ROIALIGN_FWD_BUF_KB")) {
      const int kb = atoi(e);
      if (kb >= 2 && kb <= 64) buf_floats = kb * 256;
    }
    const size_t lds2 = 2 * static_cast<size_t>(buf_floats) * sizeof(float);
    int wps = kFwdDmaWps;
    if (const char* e = getenv("DETOPS_ROIALIGN_FWD_WPS")) wps = atoi(e);
    const int32_t* order = nullptr;
    // order_ws != nullptr implies K >= order_min_k(); maps that fit an L2 slice several times over need no ranking
    const char* oe = getenv("DETOPS_ROIALIGN_FWD_ORDER");          // "force": rank even for small maps (tests)
    if (order_ws && K <= kOrderMaxK && (map_pixels * C * 4 > (2 << 20) || (oe && oe[0] == 'f'))) {
      const unsigned og = static_cast<unsigned>(ceil_div64(static_cast<int64_t>(K) * kOrderLanes, kOrderBlock));
      hipLaunchKernelGGL(roi_order_kernel, dim3(og), dim3(kOrderBlock), 0, st, L, rois, levels_in, K, order_ws);
      order = order_ws;
    }
#define FWD_DMA_LAUNCH(W_)                                                                                        \
  hipLaunchKernelGGL((roi_align_fwd_dma_kernel<PH, PW, SR, G, W_>), grid, dim3(NT), lds2, st, L, rois, levels_in, \
                     levels_out, out, C, K, CT, chunks, buf_floats, order)
    if (wps == 8) FWD_DMA_LAUNCH(8); else if (wps == 6) FWD_DMA_LAUNCH(6); else if (wps == 4) FWD_DMA_LAUNCH(4); else FWD_DMA_LAUNCH(5);
#undef FWD_DMA_LAUNCH
    return;
  }
  const size_t lds = (patch_floats + 64) * sizeof(float);
#define FWD_LDS_LAUNCH(U_)                                                                                       \
  hipLaunchKernelGGL((roi_align_fwd_lds_kernel<PH, PW, SR, G, U_, kFwdWps>), grid, dim3(NT), lds, st, L, rois,   \
                     levels_in, levels_out, out, C, K, CT, chunks, patch_floats)
  if (unroll == 8) FWD_LDS_LAUNCH(8); else if (unroll == 2) FWD_LDS_LAUNCH(2); else FWD_LDS_LAUNCH(4);
#undef FWD_LDS_LAUNCH
}

int run_forward(const Levels& L, const float* rois, const int32_t* levels_in, int32_t* levels_out,
                float* out, int C, int K, int PH, int PW, int sr, hipStream_t st, int32_t* order_ws = nullptr) {
  if (const char* e = getenv("DETOPS_ROIALIGN_FWD_ORDER"))
    if (e[0] == '0') order_ws = nullptr;            // A/B runs: launch order = ROI order
  if (K == 0 || C == 0) return 0;
  // DETOPS_ROIALIGN_FWD=generic forces the gather kernel (A/B measurements; default: LDS fast path)
  static const bool force_generic = [] {
    const char* e = getenv("DETOPS_ROIALIGN_FWD");
    return e && e[0] == 'g';
  }();
  if (force_generic) {
  } else if (PH == 7 && PW == 7 && sr == 2) {
    launch_fwd_lds<7, 7, 2, 5>(L, rois, levels_in, levels_out, out, C, K, st, order_ws);
    return launch_status();
  }
  else if (PH == 14 && PW == 14 && sr == 2) {
    launch_fwd_lds<14, 14, 2, 2>(L, rois, levels_in, levels_out, out, C, K, st, order_ws);
    return launch_status();
  }
  else if (PH == 7 && PW == 7 && sr == 1) {
    launch_fwd_lds<7, 7, 1, 5>(L, rois, levels_in, levels_out, out, C, K, st, order_ws);
    return launch_status();
  }
  const int CT = pick_chunk(C, K);
  const int chunks = static_cast<int>(ceil_div64(C, CT));
  const dim3 grid(static_cast<unsigned>(K) * chunks);
  dispatch_shape(PH, PW, sr, [&](auto ph, auto pw, auto tab) {
    hipLaunchKernelGGL((roi_align_fwd_kernel<decltype(ph)::value, decltype(pw)::value, decltype(tab)::value>),
                       grid, dim3(kBlock), 0, st, L, rois, levels_in, levels_out, out, C, K, PH, PW, sr, CT, chunks);
  });
  return launch_status();
}

inline int pow2_at_least(int v, int cap) {
  int p = 1;
  while (p < v && p < cap) p <<= 1;
  return p;
}

int run_backward_tiles(const Levels& L, const float* rois, const int32_t* levels_in, const float* gout,
                       int N, int C, int K, int PH, int PW, int sr, int accumulate, hipStream_t st) {
  if (C == 0 || N == 0) return 0;
  int Hmax = 0, Wmax = 0;
  for (int i = 0; i < L.num; ++i) { Hmax = max(Hmax, L.lv[i].H); Wmax = max(Wmax, L.lv[i].W); }
  BwdPlan P{};
  P.TW = pow2_at_least(Wmax, 64);
  P.TH = pow2_at_least(Hmax, 32);
  P.accumulate = accumulate;
  int CT = min(kBwdMaxCT, max(1, kAccFloats / (P.TH * (P.TW + 1) + 3)));
  CT = min(CT, C);
  auto count_items = [&](int ct) {
    int64_t items = 0;
    for (int i = 0; i < L.num; ++i)
      items += static_cast<int64_t>(N) * ceil_div64(L.lv[i].H, P.TH) * ceil_div64(L.lv[i].W, P.TW) * ceil_div64(C, ct);
    return items;
  };
  while (CT > 1 && count_items(CT) < 4 * kNumCU) CT >>= 1;
  P.CT = CT;
  P.chunks = static_cast<int>(ceil_div64(C, CT));
  int64_t items = 0;
  for (int i = 0; i < L.num; ++i) {
    P.tiles_x[i] = static_cast<int>(ceil_div64(L.lv[i].W, P.TW));
    P.tiles_y[i] = static_cast<int>(ceil_div64(L.lv[i].H, P.TH));
    P.first_item[i] = static_cast<int>(items);
    items += static_cast<int64_t>(N) * P.tiles_x[i] * P.tiles_y[i] * P.chunks;
  }
  for (int i = L.num; i <= DETOPS_MAX_LEVELS; ++i) P.first_item[i] = static_cast<int>(items);
  if (items > 0x7fffffff) return DETOPS_EUNSUPPORTED;
  const size_t lds = sizeof(float) * static_cast<size_t>(CT) * (P.TH * (P.TW + 1) + 3);
  const dim3 grid(static_cast<unsigned>(items));
#define BWD_LAUNCH(PH_, PW_, SR_)                                                                          \
  hipLaunchKernelGGL((roi_align_bwd_tile_kernel<PH_, PW_, SR_>), grid, dim3(kBlock), lds, st, L, P, rois, \
                     levels_in, gout, C, K, PH, PW, sr)
  if (PH == 7 && PW == 7 && sr == 2) BWD_LAUNCH(7, 7, 2);
  else if (PH == 14 && PW == 14 && sr == 2) BWD_LAUNCH(14, 14, 2);
  else if (PH == 7 && PW == 7) BWD_LAUNCH(7, 7, 0);
  else if (PH == 14 && PW == 14) BWD_LAUNCH(14, 14, 0);
  else BWD_LAUNCH(0, 0, 0);
#undef BWD_LAUNCH
  return launch_status();
}

// ------------------------------------------------------------------------------------------
// backward, BINNED pixel-owner (the default for filled launches; needs a caller workspace).
//
// Same adjoint-gather formulation and the same per-lane walk as roi_align_bwd_gather_kernel, with
// everything that kernel re-derived per (tile, channel chunk) workgroup hoisted into one small
// pre-pass launch (rocprofv3 PMC of the scan kernel, profiles/r02a_pmc_diag.txt: 63 % of the wave
// cycles parked at barriers / memory waits, 68 M VALU wave-instructions of which ~40 M were the
// 4 ROI-scan rounds of every workgroup and the per-batch coefficient-row builds, replicated over
// the 16 channel chunks):
//   roi_bwd_prep_kernel, role A — one wave per ROI, a lane per (axis, footprint pixel): the pixel's row of
//       the adjoint matrix AY_r[y, 0:PH] (or AX_r[x, 0:PW]) from the reference's sample taps, plus
//       the pixel's contiguous range of contributing bins packed into the row's spare column.
//       Rows are stored footprint-relative (row 0 = the ROI's first reachable pixel) in a per-ROI
//       slot of the workspace.
//   roi_bwd_prep_kernel, role B — a wave per gradient-map tile (8 tiles per workgroup sharing one LDS copy
//       of the ROI footprints): ordered (ascending ROI index) ballot compaction of the ROIs that reach the tile -> hit list [tile][cap] of
//       {roi, fy0, fx0, ny|nx} + count.  Tiles nothing reaches get count 0.
// The main kernel then only (1) reads its tile's list, (2) per batch of hits copies the 8 + 32
// coefficient rows and the pooled gradients of its channel chunk into LDS — the NEXT batch's
// global loads are issued before the walk of the current one —, (3) walks.  Empty tiles are pure
// zero-stores.  Determinism and the one-store-per-element property are unchanged.
// ------------------------------------------------------------------------------------------
struct BinPlan {
  int first_tile[DETOPS_MAX_LEVELS];   // tile-id offset per level (coarsest level first)
  int n_tiles[DETOPS_MAX_LEVELS];
  int tiles_x[DETOPS_MAX_LEVELS], tiles_y[DETOPS_MAX_LEVELS];
  int num_tiles, cap;                  // cap = hit-list capacity per tile (= K)
  int chunks, accumulate, batch;
  int PPH, PPW;                        // coefficient-row strides: bins + >= 1 spare column, multiple of 4
  int Hmax, Wmax;                      // per-ROI table slot = Hmax * PPH + Wmax * PPW floats
  int tab_blocks;                      // prep launch: blocks [0, tab_blocks) = role A, the rest role B
  int debug;                           // ablation knob (DETOPS_ROIALIGN_BWD_DEBUG): 1 = skip the walk
};

struct BinWs {
  float* tabs;       // [K][Hmax * PPH + Wmax * PPW]
  int4* heads;       // [num_tiles] {hit count, level | image << 8, y0, x0}
  int4* lists;       // [num_tiles][cap]
  unsigned long long* prof;   // DETOPS_ROIALIGN_BWD_DEBUG & 128: per-phase shader-clock sums (diagnosis only), else NULL
};

struct RoiExtent {
  int b;
  int fy0, ny, fx0, nx;   // rows / columns any tap of the ROI can reach, clipped to the map (n <= 0: none)
};

// Conservative footprint from the scaled ROI rectangle alone (cheap: evaluated per (tile, ROI) pair by
// the binning role): taps lie in floor(first sample) .. floor(last sample) + 1, samples in
// [start, start + extent * (1 + a few ulp)); "+ 2" covers the rounding of the sample coordinate.
__device__ __forceinline__ RoiExtent roi_extent(const float* __restrict__ roi, float scale, int H, int W) {
#pragma clang fp contract(off)
  RoiExtent e;
  e.b = static_cast<int>(roi[0]);
  const float start_w = roi[1] * scale, start_h = roi[2] * scale;
  const float rw = fmaxf(roi[3] * scale - start_w, 1.f), rh = fmaxf(roi[4] * scale - start_h, 1.f);
  const float fy0 = fminf(floorf(fmaxf(start_h, 0.f)), static_cast<float>(H));
  const float fy1 = fminf(fmaxf(floorf(start_h + rh) + 2.f, -1.f), static_cast<float>(H - 1));
  const float fx0 = fminf(floorf(fmaxf(start_w, 0.f)), static_cast<float>(W));
  const float fx1 = fminf(fmaxf(floorf(start_w + rw) + 2.f, -1.f), static_cast<float>(W - 1));
  e.fy0 = static_cast<int>(fy0); e.ny = static_cast<int>(fy1) - e.fy0 + 1;
  e.fx0 = static_cast<int>(fx0); e.nx = static_cast<int>(fx1) - e.fx0 + 1;
  if (!(fy1 >= fy0)) e.ny = 0;   // also catches NaN coordinates
  if (!(fx1 >= fx0)) e.nx = 0;
  return e;
}

constexpr int kPrepTiles = 4;      // tiles per role-B workgroup (1 per wave)
constexpr int kPrepRois = 1024;    // ROI footprints parked in LDS per pass

__global__ void __launch_bounds__(kBlock)
roi_bwd_prep_kernel(Levels L, BinPlan P, BinWs ws, const float* __restrict__ rois,
                    const int32_t* __restrict__ levels_in, int K, int PH, int PW, int sr) {
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  if (static_cast<int>(blockIdx.x) < P.tab_blocks) {
    // ---- role A: adjoint rows.  One wave per ROI; lanes stride over its ny + nx footprint pixels
    const int r = static_cast<int>(blockIdx.x) * (kBlock / kWave) + wave;
    if (r >= K || (P.debug & 2)) return;
    const float* roi = rois + static_cast<size_t>(r) * 5;
    const int lvl = (L.num > 1) ? levels_in[r] : 0;
    int H = L.lv[0].H, W = L.lv[0].W; float scale = L.lv[0].scale;
#pragma unroll
    for (int i = 1; i < DETOPS_MAX_LEVELS; ++i)
      if (i == lvl) { H = L.lv[i].H; W = L.lv[i].W; scale = L.lv[i].scale; }
    if (lvl < 0 || lvl >= L.num) return;
    const RoiGeom g = roi_geometry(roi, scale, PH, PW, sr);
    const RoiExtent e = roi_extent(roi, scale, H, W);
    float* slot = ws.tabs + static_cast<size_t>(r) * (static_cast<size_t>(P.Hmax) * P.PPH + static_cast<size_t>(P.Wmax) * P.PPW);
    const int ny = max(e.ny, 0), nx = max(e.nx, 0);
    for (int p = lane; p < ny + nx; p += kWave) {
      const bool isy = p < ny;
      const int pi = isy ? p : p - ny;
      const int pix = (isy ? e.fy0 : e.fx0) + pi;
      const int PB = isy ? PH : PW, PP = isy ? P.PPH : P.PPW;
      const int grid = isy ? g.gh : g.gw, size = isy ? H : W;
      const float start = isy ? g.start_h : g.start_w, bin = isy ? g.bin_h : g.bin_w;
      float* row = slot + (isy ? 0 : static_cast<size_t>(P.Hmax) * P.PPH) + static_cast<size_t>(pi) * PP;
      const float inv = 1.f / static_cast<float>(grid);
      for (int q = 0; q < PP; ++q) row[q] = 0.f;
      // only samples whose coordinate lies within one pixel of `pix` can have a tap on it (border pixels
      // also collect the clamped samples: c in [-1, 0] -> pixel 0, c in [size-1, size] -> pixel size-1).
      // Candidate sample range from the inverse of c(s) = start + (s + .5) * bin / grid, widened by one
      // sample on each side; the exact reference arithmetic then decides.
      const float step = bin * inv;
      const float clo = (pix == 0) ? -1.f : static_cast<float>(pix - 1);
      const float chi = (pix == size - 1) ? static_cast<float>(size) : static_cast<float>(pix + 1);
      const int ns = PB * grid;
      int s0 = static_cast<int>(fminf(fmaxf(floorf((clo - start) / step - 0.5f) - 1.f, 0.f), static_cast<float>(ns)));
      int s1 = static_cast<int>(fminf(fmaxf(ceilf((chi - start) / step - 0.5f) + 1.f, -1.f), static_cast<float>(ns - 1)));
      if (!(step > 0.f)) { s0 = 0; s1 = ns - 1; }   // degenerate geometry (NaN / inf): look at everything
      // compact row: [0] = first contributing bin | count << 16, [1 ...] = the weights of bins lo, lo + 1, ...
      // (zero-filled beyond), so the walk gets range + the first three weights with ONE 16-byte LDS read
      int lo = -1, hi = -1;
      int q = s0 / grid, i = s0 - q * grid;
      float w = 0.f;
      for (int sidx = s0; sidx <= s1; ++sidx) {   // per bin: samples in ascending order, like the scan kernel
        const Tap tp = axis_entry(start, bin, q, i, grid, size, 1);
        if (tp.lo == pix) w += tp.h * inv;
        if (tp.hi == pix) w += tp.l * inv;
        if (++i == grid || sidx == s1) {
          if (w != 0.f) {
            if (lo < 0) lo = q;
            hi = q;
            row[1 + q - lo] = w;
          }
          w = 0.f; i = 0; ++q;
        }
      }
      row[0] = __int_as_float((lo >= 0) ? (lo | ((hi - lo + 1) << 16)) : 0);
    }
    return;
  }
  // ---- role B: hit lists.  A workgroup owns kPrepTiles tiles.  Per pass of <= 1024 ROIs: (1) all threads
  //      park the ROIs' footprints {roi | level | image, fy0, fx0, ny | nx} in LDS — every ROI is read from
  //      memory once per WORKGROUP, not once per tile: with a block per tile the ~3000 waves of the launch
  //      queued on the same few cache lines of `rois` (5.5 us floor measured at K = 2) —, (2) each wave
  //      takes tiles of the set and compacts the ROIs reaching it in ascending index (ballot + popcount).
  __shared__ int4 s_ext[kPrepRois];
  const int tile0 = (static_cast<int>(blockIdx.x) - P.tab_blocks) * kPrepTiles;
  if (P.debug & 4) return;
  int cnt[kPrepTiles / (kBlock / kWave)];
#pragma unroll
  for (int t = 0; t < kPrepTiles / (kBlock / kWave); ++t) cnt[t] = 0;
  for (int r0 = 0; r0 < K; r0 += kPrepRois) {
    const int nr = min(kPrepRois, K - r0);
    // all four ROIs of a thread are fetched before any is used (one memory round trip per pass, not four)
    constexpr int kPer = kPrepRois / kBlock;
    float rv[kPer][5];
    int rlv[kPer];
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const int i = tid + k * kBlock;
      const int r = r0 + min(i, nr - 1);
      const float* roi = rois + static_cast<size_t>(r) * 5;
#pragma unroll
      for (int c = 0; c < 5; ++c) rv[k][c] = roi[c];
      rlv[k] = (L.num > 1) ? levels_in[r] : 0;
    }
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const int i = tid + k * kBlock;
      if (i >= nr) continue;
      const int r = r0 + i;
      const int rl = rlv[k];
      int H = L.lv[0].H, W = L.lv[0].W; float scale = L.lv[0].scale;
#pragma unroll
      for (int q = 1; q < DETOPS_MAX_LEVELS; ++q)
        if (q == rl) { H = L.lv[q].H; W = L.lv[q].W; scale = L.lv[q].scale; }
      int4 ent = make_int4(-1, 0, 0, 0);   // never matches a tile
      if (rl >= 0 && rl < L.num) {
        const RoiExtent e = roi_extent(rv[k], scale, H, W);
        if (e.ny > 0 && e.nx > 0 && e.b >= 0 && e.b < 4096)
          ent = make_int4(r | (rl << 16) | (e.b << 19), e.fy0, e.fx0, (e.ny << 16) | e.nx);
      }
      s_ext[i] = ent;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < kPrepTiles / (kBlock / kWave); ++t) {
      const int tile = tile0 + t * (kBlock / kWave) + wave;
      if (tile >= P.num_tiles || (P.debug & 8)) continue;
      int lvl = 0;
#pragma unroll
      for (int i = 1; i < DETOPS_MAX_LEVELS; ++i)
        if (i < L.num && tile >= P.first_tile[i] && tile < P.first_tile[i] + P.n_tiles[i]) lvl = i;
      int H = L.lv[0].H, W = L.lv[0].W;
      int ntx = P.tiles_x[0], nty = P.tiles_y[0], first = P.first_tile[0];
#pragma unroll
      for (int i = 1; i < DETOPS_MAX_LEVELS; ++i)
        if (i == lvl) { H = L.lv[i].H; W = L.lv[i].W; ntx = P.tiles_x[i]; nty = P.tiles_y[i]; first = P.first_tile[i]; }
      int rem = tile - first;
      const int tix = rem % ntx; rem /= ntx;
      const int tiy = rem % nty;
      const int b = rem / nty;
      const int y0 = tiy * kGTH, x0 = tix * kGTW;
      const int y1 = min(H, y0 + kGTH) - 1, x1 = min(W, x0 + kGTW) - 1;
      const int key = (lvl << 16) | (b << 19);
      int4* list = ws.lists + static_cast<size_t>(tile) * P.cap;
      int c = cnt[t];
      for (int i0 = 0; i0 < nr; i0 += 4 * kWave) {   // four footprints per lane in flight per trip
        int4 en[4];
        bool hit[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) en[u] = s_ext[min(i0 + u * kWave + lane, nr - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          hit[u] = i0 + u * kWave + lane < nr && (en[u].x & ~0xffff) == key && en[u].x >= 0 && en[u].y <= y1 &&
                   en[u].y + (en[u].w >> 16) - 1 >= y0 && en[u].z <= x1 && en[u].z + (en[u].w & 0xffff) - 1 >= x0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const unsigned long long m = __ballot(hit[u]);
          if (hit[u] && !(P.debug & 16)) { en[u].x &= 0xffff; list[c + __popcll(m & ((1ull << lane) - 1ull))] = en[u]; }
          c += __popcll(m);
        }
      }
      cnt[t] = c;
    }
    __syncthreads();   // s_ext is rewritten by the next pass
  }
#pragma unroll
  for (int t = 0; t < kPrepTiles / (kBlock / kWave); ++t) {
    const int tile = tile0 + t * (kBlock / kWave) + wave;
    if (tile >= P.num_tiles || lane != 0) continue;
    int lvl = 0;
#pragma unroll
    for (int i = 1; i < DETOPS_MAX_LEVELS; ++i)
      if (i < L.num && tile >= P.first_tile[i] && tile < P.first_tile[i] + P.n_tiles[i]) lvl = i;
    int ntx = P.tiles_x[0], nty = P.tiles_y[0], first = P.first_tile[0];
#pragma unroll
    for (int i = 1; i < DETOPS_MAX_LEVELS; ++i)
      if (i == lvl) { ntx = P.tiles_x[i]; nty = P.tiles_y[i]; first = P.first_tile[i]; }
    int rem = tile - first;
    const int tix = rem % ntx; rem /= ntx;
    ws.heads[tile] = make_int4(cnt[t], lvl | ((rem / nty) << 8), (rem % nty) * kGTH, tix * kGTW);
  }
}

// One bin row of a (wave, hit) task of the binned walk with a compile-time column count: NB bodies,
// straight-line, so the 4 * NB gradient reads are in flight together (the generic loop is a chain of
// dependent LDS round trips per body).  wx is zero beyond a lane's own range, bin indices are clamped into
// the hit's staged block (a zero weight times a staged value: exact for finite gradients).
template <int NB, int CG>
__device__ __forceinline__ void binned_walk_row(float* __restrict__ acc, const float4* __restrict__ grow, int slots,
                                                int PW, int xlo, float wya, const float* wx) {
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const float w = wya * wx[b];
    const float4* gp = grow + min(xlo + b, PW - 1);
#pragma unroll
    for (int cg = 0; cg < CG; ++cg) {
      const float4 g4 = gp[cg * slots];
      acc[4 * cg + 0] = fmaf(w, g4.x, acc[4 * cg + 0]);
      acc[4 * cg + 1] = fmaf(w, g4.y, acc[4 * cg + 1]);
      acc[4 * cg + 2] = fmaf(w, g4.z, acc[4 * cg + 2]);
      acc[4 * cg + 3] = fmaf(w, g4.w, acc[4 * cg + 3]);
    }
  }
}

constexpr int kBinMaxBatch = 8;      // hits staged per batch (<= 256 bins of gradients per float4 channel group)
constexpr int kBinRound = 64;        // hit-list entries parked in LDS per round

// Main kernel.  One workgroup per work item (tile, CT-channel chunk); the design goal is RESIDENCY: the
// kernel is a chain of short dependent phases (hit list -> staged gradients / adjoint rows -> walk -> store),
// so it is the number of workgroups a CU can interleave that hides their latencies (measured on the way here:
// 4 -> 5 waves per SIMD = 140 -> 118 us; a persistent, register-prefetching variant at 4 waves per SIMD was
// slower than either).  Hence: <= 64 VGPRs (no data held in registers across a barrier except the CT
// accumulators), ~19 KiB of LDS -> 8 workgroups per CU.
//   lanes       = pixels: wave w owns rows 2w, 2w+1 of the 8 x 32 tile, lane = (row & 1) * 32 + column.  A
//                 wave's store of one channel is two full 128-byte rows straight from the accumulator
//                 registers — no LDS transposition, no barrier on the store path.
//   batch       = <= P.batch hits: global -> LDS staging of their pooled gradients (float4 channel groups,
//                 slot = hit * bins + bin) and of the 8 + 32 compact adjoint rows of the tile | barrier | walk
//                 | barrier.  The walk reads {first bin | count, 3 weights} of a pixel with one 16-byte LDS
//                 read per axis and runs bin rows with the column count as a compile-time constant.
template <int PH_, int PW_, int CT>
__global__ void __launch_bounds__(kBlock, (CT <= 16 ? 6 : 3))   // 2nd argument: waves per SIMD the register budget must allow
roi_align_bwd_binned_kernel(Levels L, BinPlan P, BinWs ws, const float* __restrict__ gout, int C, int PHr, int PWr) {
  static_assert(CT % 4 == 0, "channels are staged as float4 groups");
  constexpr int CG = CT / 4;
  const int PH = PH_ ? PH_ : PHr;
  const int PW = PW_ ? PW_ : PWr;
  const int bins = PH * PW;
  const int PPH = PH_ ? ((PH_ + 4) & ~3) : P.PPH;          // compile-time strides for the model's shapes
  const int PPW = PW_ ? ((PW_ + 4) & ~3) : P.PPW;
  const int PXS = PPW + 1;                                 // AX rows in LDS: odd stride, 32 columns -> 32 banks
  const int slots = P.batch * bins;                        // (hit, bin) slots per float4 channel group

  // Separable walk (7 bin columns): grad_in[y, x] += sum_b AX[x, b] * (sum_a AY[y, a] * g[a, b]).  Pass 1: lane =
  // (row of the wave, bin column, channel group) builds T[row][bin column] = sum_a AY * g for its channels and parks
  // it in a per-wave LDS strip; pass 2: lane = pixel sums <= 3 strip entries.  3 + 3 multiply-adds per pixel and
  // channel where the direct form has up to 9, and a third of its LDS reads.
  constexpr bool kSep = (PW_ == 7);   // measured: box head 117 -> 110 us; 14 columns (16-wide strip, one hit per batch) 85 -> 88 us, not used
  constexpr int PWP = (PW_ <= 8) ? 8 : 16;                 // strip columns
  constexpr int LPC = 32 / PWP;                            // pass-1 lanes side by side on one (row, bin column)
  constexpr int CPL = (CG + LPC - 1) / LPC;                // channel groups per pass-1 lane

  DETOPS_DYNAMIC_LDS(float, g_lds);
  float4* gs4 = reinterpret_cast<float4*>(g_lds);          // [CG][slots] float4
  float* ayt = g_lds + slots * CT;                         // [batch][kGTH][PPH]
  float* axt = ayt + P.batch * kGTH * PPH;                 // [batch][kGTW][PXS]
  float4* tb4 = reinterpret_cast<float4*>(axt + P.batch * kGTW * PXS);   // [wave][2][PWP][CG] float4 (kSep only)
  __shared__ int4 s_ent[kBinRound];

  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  const int yl = 2 * wave + (lane >> 5), xl = lane & 31;   // this thread's pixel within the tile
  const int tile = static_cast<int>(blockIdx.x) / P.chunks;
  const int c0 = (static_cast<int>(blockIdx.x) - tile * P.chunks) * CT;
  const int4 hd = ws.heads[tile];                          // {hit count, level | image << 8, y0, x0}, wave-uniform
  const int total = __builtin_amdgcn_readfirstlane(hd.x);
  const int lvl = __builtin_amdgcn_readfirstlane(hd.y) & 0xff, b = __builtin_amdgcn_readfirstlane(hd.y) >> 8;
  const int y0 = __builtin_amdgcn_readfirstlane(hd.z), x0 = __builtin_amdgcn_readfirstlane(hd.w);
  const size_t tab_stride = static_cast<size_t>(P.Hmax) * P.PPH + static_cast<size_t>(P.Wmax) * P.PPW;
  const size_t ax_off = static_cast<size_t>(P.Hmax) * P.PPH;
  const int4* list = ws.lists + static_cast<size_t>(tile) * P.cap;
  const bool full = c0 + CT <= C;                          // no channel tail in this chunk
  const int wy0 = y0 + 2 * wave;                           // the wave's two rows: wy0, wy0 + 1

  float acc[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) acc[c] = 0.f;
#ifndef DETOPS_CPU_EMU
  // phase clocks of wave 0 (diagnosis build path: ws.prof != NULL only under DETOPS_ROIALIGN_BWD_DEBUG & 128)
  const bool prof = ws.prof != nullptr && tid == 0;
  long long pt0 = 0, pt = 0, p_stage = 0, p_walk = 0, p_list = 0;
  if (prof) pt0 = pt = clock64();
#define BIN_PROF(acc_) do { if (prof) { const long long n_ = clock64(); acc_ += n_ - pt; pt = n_; } } while (0)
#else
#define BIN_PROF(acc_) do { } while (0)
#endif

  // staging roles (no divisions inside the loops): gradients — thread t owns slot t = (hit, bin) of every
  // channel group (t < batch * bins <= 256); adjoint rows — float4 unit u = tid + i * 256 of the batch
  const int g_j = tid / bins, g_bin = tid - g_j * bins;
  const int tabq = (kGTH * PPH + kGTW * PPW) / 4;          // float4 units per hit
  const int qy = kGTH * PPH / 4, rqy = PPH / 4, rqx = PPW / 4;

  for (int r0 = 0; r0 < total; r0 += kBinRound) {
    const int nr = min(kBinRound, total - r0);
    if (tid < nr) s_ent[tid] = list[r0 + tid];
    __syncthreads();
    BIN_PROF(p_list);
    if (tid == 0) { DETOPS_STAT("bwdb.rounds", r0 > 0); if (r0 == 0) DETOPS_STAT("bwdb.hits", total); }
    for (int h0 = 0; h0 < nr; h0 += P.batch) {
      const int nb = min(P.batch, nr - h0);
      if (tid == 0) DETOPS_STAT("bwdb.batches", 1);
      // ---- stage the batch: global -> (registers) -> LDS
      if (g_j < nb) {
        const float* src = gout + (static_cast<size_t>(s_ent[h0 + g_j].x) * C + c0) * bins + g_bin;
        if (full) {
#pragma unroll
          for (int cg = 0; cg < CG; ++cg)
            gs4[cg * slots + tid] = make_float4(src[(cg * 4 + 0) * bins], src[(cg * 4 + 1) * bins],
                                                src[(cg * 4 + 2) * bins], src[(cg * 4 + 3) * bins]);
        } else {
#pragma unroll
          for (int cg = 0; cg < CG; ++cg) {
            const int cb = c0 + cg * 4;
            gs4[cg * slots + tid] = make_float4((cb + 0 < C) ? src[(cg * 4 + 0) * bins] : 0.f,
                                                (cb + 1 < C) ? src[(cg * 4 + 1) * bins] : 0.f,
                                                (cb + 2 < C) ? src[(cg * 4 + 2) * bins] : 0.f,
                                                (cb + 3 < C) ? src[(cg * 4 + 3) * bins] : 0.f);
          }
        }
      }
      for (int u = tid; u < nb * tabq; u += kBlock) {
        const int j = u / tabq, q = u - j * tabq;
        const int4 en = s_ent[h0 + j];
        const float* slot = ws.tabs + static_cast<size_t>(en.x) * tab_stride;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q < qy) {            // AY row of tile row yy
          const int yy = q / rqy, k4 = q - yy * rqy;
          const int ri = y0 + yy - en.y;
          if (ri >= 0 && ri < (en.w >> 16)) v = reinterpret_cast<const float4*>(slot + static_cast<size_t>(ri) * P.PPH)[k4];
          reinterpret_cast<float4*>(ayt + j * kGTH * PPH)[q] = v;
        } else {                 // AX row of tile column xx (odd LDS stride: scalar writes)
          const int qq = q - qy;
          const int xx = qq / rqx, k4 = qq - xx * rqx;
          const int ri = x0 + xx - en.z;
          if (ri >= 0 && ri < (en.w & 0xffff)) v = reinterpret_cast<const float4*>(slot + ax_off + static_cast<size_t>(ri) * P.PPW)[k4];
          float* d = axt + (j * kGTW + xx) * PXS + k4 * 4;
          d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
      }
      __syncthreads();
      BIN_PROF(p_stage);
      // ---- walk: every pixel gathers from the bins of this batch's hits that reach it
      for (int j = 0; j < ((P.debug & 1) ? 0 : nb); ++j) {
        const int4 en = s_ent[h0 + j];
        if (en.y + (en.w >> 16) - 1 < wy0 || en.y > wy0 + 1) continue;   // this ROI misses the wave's two rows
        if (lane == 0) DETOPS_STAT("bwdb.wave_roi_tasks", 1);
        const float* ayr = ayt + (j * kGTH + yl) * PPH;
        const float* axr = axt + (j * kGTW + xl) * PXS;
        const float4* gj = gs4 + j * bins;
        // head of the pixel's compact rows: {first bin | count, w[0], w[1], w[2]}
        const float4 hy = *reinterpret_cast<const float4*>(ayr);
        const float hx0 = axr[0], hx1 = axr[1], hx2 = axr[2], hx3 = axr[3];   // AX rows have an odd LDS stride
        const int ry = __float_as_int(hy.x), rx = __float_as_int(hx0);
        const int ylo = ry & 0xffff, ny = ry >> 16, xlo = rx & 0xffff, nx = rx >> 16;
        int na = 0, nb_ = 0;
        while (__ballot(na < ny) != 0ull) ++na;
        while (__ballot(nb_ < nx) != 0ull) ++nb_;
        if (lane == 0) DETOPS_STAT("bwdb.bodies", na * nb_);
        if (kSep && na <= 3 && nb_ <= 3 && !(P.debug & 32)) {
          // ---- pass 1
          const int r1 = lane >> 5, pw1 = (lane & 31) / LPC, cgl = (lane & 31) % LPC;
          const float4 hy1 = *reinterpret_cast<const float4*>(ayt + (j * kGTH + 2 * wave + r1) * PPH);
          const int ylo1 = __float_as_int(hy1.x) & 0xffff;
          float4 t[CPL];
#pragma unroll
          for (int k = 0; k < CPL; ++k) t[k] = make_float4(0.f, 0.f, 0.f, 0.f);
          const float4* gc = gj + min(pw1, PW - 1) + cgl * slots;
          for (int a = 0; a < na; ++a) {   // wave-uniform trip count
            const float wya = (a == 0) ? hy1.y : (a == 1) ? hy1.z : hy1.w;   // zero beyond the row's own range
            const float4* gr = gc + min(ylo1 + a, PH - 1) * PW;
#pragma unroll
            for (int k = 0; k < CPL; ++k) {
              if (cgl + k * LPC < CG) {
                const float4 g4 = gr[k * LPC * slots];
                t[k].x = fmaf(wya, g4.x, t[k].x); t[k].y = fmaf(wya, g4.y, t[k].y);
                t[k].z = fmaf(wya, g4.z, t[k].z); t[k].w = fmaf(wya, g4.w, t[k].w);
              }
            }
          }
          float4* tw = tb4 + wave * (2 * PWP * CG);
          DETOPS_WAVE_SYNC();              // the previous hit's pass-2 reads of the strip are done
#pragma unroll
          for (int k = 0; k < CPL; ++k)
            if (cgl + k * LPC < CG) tw[(r1 * PWP + pw1) * CG + cgl + k * LPC] = t[k];
          DETOPS_WAVE_SYNC();
          // ---- pass 2
          const float4* tr = tw + (lane >> 5) * PWP * CG;
          for (int b2 = 0; b2 < nb_; ++b2) {   // wave-uniform trip count
            const float wxb = (b2 == 0) ? hx1 : (b2 == 1) ? hx2 : hx3;       // zero beyond the column's own range
            const float4* tp = tr + min(xlo + b2, PW - 1) * CG;
#pragma unroll
            for (int cg = 0; cg < CG; ++cg) {
              const float4 t4 = tp[cg];
              acc[4 * cg + 0] = fmaf(wxb, t4.x, acc[4 * cg + 0]);
              acc[4 * cg + 1] = fmaf(wxb, t4.y, acc[4 * cg + 1]);
              acc[4 * cg + 2] = fmaf(wxb, t4.z, acc[4 * cg + 2]);
              acc[4 * cg + 3] = fmaf(wxb, t4.w, acc[4 * cg + 3]);
            }
          }
        } else if (na <= 3 && nb_ <= 3) {
          const float wx[3] = {hx1, hx2, hx3};
          for (int a = 0; a < na; ++a) {   // wave-uniform trip count; one LDS round trip per bin row
            const float wya = (a == 0) ? hy.y : (a == 1) ? hy.z : hy.w;   // zero beyond the lane's own range
            const float4* grow = gj + min(ylo + a, PH - 1) * PW;
            if (nb_ >= 2) binned_walk_row<2, CG>(acc, grow, slots, PW, xlo, wya, wx);            // 8 reads in flight
            if (nb_ & 1) binned_walk_row<1, CG>(acc, grow, slots, PW, xlo + nb_ - 1, wya, wx + nb_ - 1);
          }
        } else {
          // long ranges (ROIs smaller than their bin grid): generic loop over the compact rows
          for (int a = 0; a < na; ++a) {
            const int ph = min(ylo + a, PH - 1);
            const float wya = (a < ny) ? ayr[1 + a] : 0.f;
            for (int b2 = 0; b2 < nb_; ++b2) {
              const int pw = min(xlo + b2, PW - 1);
              const float w = (b2 < nx) ? wya * axr[1 + b2] : 0.f;
              if (w != 0.f) {
                const float4* gp = gj + ph * PW + pw;
#pragma unroll
                for (int cg = 0; cg < CG; ++cg) {
                  const float4 g4 = gp[cg * slots];
                  acc[4 * cg + 0] = fmaf(w, g4.x, acc[4 * cg + 0]);
                  acc[4 * cg + 1] = fmaf(w, g4.y, acc[4 * cg + 1]);
                  acc[4 * cg + 2] = fmaf(w, g4.z, acc[4 * cg + 2]);
                  acc[4 * cg + 3] = fmaf(w, g4.w, acc[4 * cg + 3]);
                }
              }
            }
          }
        }
      }
      __syncthreads();  // the next batch / round rewrites the staging region
      BIN_PROF(p_walk);
    }
  }

  if (tid == 0) DETOPS_STAT("bwdb.workgroups", 1);
  if (total == 0 && P.accumulate) return;   // nothing to add
  // ---- store: two full 128-byte rows per wave and channel, straight from the accumulators; every in-map
  //      element of the tile is written exactly once (zeros where no ROI reaches)
  float* gin = L.lv[0].gin; int H = L.lv[0].H, W = L.lv[0].W;
#pragma unroll
  for (int i = 1; i < DETOPS_MAX_LEVELS; ++i)
    if (i == lvl) { gin = L.lv[i].gin; H = L.lv[i].H; W = L.lv[i].W; }
  if (y0 + yl < H && x0 + xl < W) {
    const size_t plane = static_cast<size_t>(H) * W;
    float* dst = gin + (static_cast<size_t>(b) * C + c0) * plane + static_cast<size_t>(y0 + yl) * W + (x0 + xl);
    if (full && !P.accumulate) {
#pragma unroll
      for (int c = 0; c < CT; ++c) dst[c * plane] = acc[c];
    } else {
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        if (c0 + c < C) {
          float v = acc[c];
          if (P.accumulate) v += dst[c * plane];
          dst[c * plane] = v;
        }
      }
    }
  }
#ifndef DETOPS_CPU_EMU
  if (prof) {
    const long long end = clock64();
    const int cls = total == 0 ? 0 : (total <= 4 ? 1 : (total <= 16 ? 2 : 3));   // workgroups by hit count
    unsigned long long* o = ws.prof + cls * 8;
    atomicAdd(o + 0, 1ull);
    atomicAdd(o + 1, static_cast<unsigned long long>(end - pt0));
    atomicAdd(o + 2, static_cast<unsigned long long>(p_list));
    atomicAdd(o + 3, static_cast<unsigned long long>(p_stage));
    atomicAdd(o + 4, static_cast<unsigned long long>(p_walk));
    atomicAdd(o + 5, static_cast<unsigned long long>(end - pt));          // store tail
    atomicAdd(o + 6, static_cast<unsigned long long>(total));
  }
#endif
#undef BIN_PROF
}

// per-wave LDS strips of the separable walk (7 bin columns): [waves][2 rows][8 | 16 columns][CT] floats
inline int bin_strip_floats(int PW, int CT) {
  if (PW != 7) return 0;
  return (kBlock / kWave) * 2 * (PW <= 8 ? 8 : 16) * CT;
}

// Plan + workspace carve of the binned backward.  Returns false when the shape is outside its plan
// (bins > 256, coefficient rows beyond the prefetch registers, maps wider than the packed fields).
struct BinLayout { size_t off_counts, off_lists, off_tabs, total; };

bool bin_plan(const Levels& L, int N, int C, int K, int PH, int PW, int CT, BinPlan& P, BinLayout& lay) {
  const int bins = PH * PW;
  if (bins > kGBins || K <= 0 || K > 65535 || N > 4096) return false;
  P = BinPlan{};
  P.PPH = (PH + 4) & ~3;
  P.PPW = (PW + 4) & ~3;
  P.batch = max(1, min(kBinMaxBatch, kGBins / bins));
  if (const char* e = getenv("DETOPS_ROIALIGN_BWD_BATCH")) P.batch = max(1, min(P.batch, atoi(e)));   // tuning knob
  // LDS plan: gradients of a batch (16 x bins floats per hit at CT = 16) + its adjoint rows, <= 48 KiB
  while (P.batch > 1 && static_cast<int64_t>(P.batch) * (bins * 32 + kGTH * P.PPH + kGTW * (P.PPW + 1)) * 4 > 48 * 1024) --P.batch;
  // shapes with the separable walk carry a per-wave strip (bin_strip_floats); keep 6 workgroups per CU (<= 26 KiB)
  while (P.batch > 2 && bin_strip_floats(PW, CT) > 0 &&
         (static_cast<int64_t>(P.batch) * (bins * CT + kGTH * P.PPH + kGTW * (P.PPW + 1)) + bin_strip_floats(PW, CT)) * 4 > 25 * 1024)
    --P.batch;
  if (static_cast<int64_t>(bins * 32 + kGTH * P.PPH + kGTW * (P.PPW + 1)) * 4 > 60 * 1024) return false;
  P.cap = K;
  P.chunks = static_cast<int>(ceil_div64(C, CT));
  int64_t tiles = 0;
  for (int i = L.num - 1; i >= 0; --i) {  // coarsest level first (its tiles see the most ROIs)
    if (L.lv[i].W > 32767 || L.lv[i].H > 32767) return false;
    P.Hmax = max(P.Hmax, L.lv[i].H);
    P.Wmax = max(P.Wmax, L.lv[i].W);
    P.tiles_x[i] = static_cast<int>(ceil_div64(L.lv[i].W, kGTW));
    P.tiles_y[i] = static_cast<int>(ceil_div64(L.lv[i].H, kGTH));
    const int64_t n = static_cast<int64_t>(N) * P.tiles_x[i] * P.tiles_y[i];
    P.first_tile[i] = static_cast<int>(tiles);
    P.n_tiles[i] = static_cast<int>(n);
    tiles += n;
  }
  if (tiles <= 0 || tiles * P.chunks > 0x7fffffff) return false;
  P.num_tiles = static_cast<int>(tiles);
  P.tab_blocks = static_cast<int>(ceil_div64(K, kBlock / kWave));   // role A: one wave per ROI
  auto up = [](size_t v) { return (v + 255) & ~static_cast<size_t>(255); };
  size_t o = 0;
  lay.off_counts = o; o = up(o + sizeof(int4) * P.num_tiles);
  lay.off_lists = o;  o = up(o + sizeof(int4) * static_cast<size_t>(P.num_tiles) * P.cap);
  lay.off_tabs = o;   o = up(o + sizeof(float) * static_cast<size_t>(K) *
                                 (static_cast<size_t>(P.Hmax) * P.PPH + static_cast<size_t>(P.Wmax) * P.PPW));
  lay.total = o;
  return true;
}

inline int bin_default_ct() {
  if (const char* e = getenv("DETOPS_ROIALIGN_BWD_CT")) return atoi(e) == 32 ? 32 : 16;   // tuning / test knob
  return 16;
}

// -1: not applicable (no / too small workspace, shape outside the plan, underfilled launch) -> scan kernel
int run_backward_binned(const Levels& L, const float* rois, const int32_t* levels_in, const float* gout,
                        int N, int C, int K, int PH, int PW, int sr, int accumulate, void* workspace,
                        size_t workspace_bytes, bool forced, hipStream_t st) {
  if (C == 0 || N == 0) return 0;
  if (!workspace || K == 0) return -1;
  const int CT = bin_default_ct();
  BinPlan P; BinLayout lay;
  if (!bin_plan(L, N, C, K, PH, PW, CT, P, lay) || workspace_bytes < lay.total) return -1;
  // underfilled launches (a handful of tiles): the scan kernel's ROI-list split serves them better
  if (!forced && static_cast<int64_t>(P.num_tiles) * ceil_div64(C, 16) < 2 * kNumCU) return -1;
  P.accumulate = accumulate;
  if (const char* e = getenv("DETOPS_ROIALIGN_BWD_DEBUG")) P.debug = atoi(e);
  unsigned char* base = static_cast<unsigned char*>(workspace);
  BinWs ws{reinterpret_cast<float*>(base + lay.off_tabs), reinterpret_cast<int4*>(base + lay.off_counts),
           reinterpret_cast<int4*>(base + lay.off_lists), nullptr};
#ifndef DETOPS_CPU_EMU
  if (P.debug & 128) {   // diagnosis only: phase clocks of every workgroup's wave 0, printed after a device sync
    static unsigned long long* prof_buf = nullptr;
    if (!prof_buf) DETOPS_HIP_TRY(hipMalloc(&prof_buf, 32 * sizeof(unsigned long long)));
    DETOPS_HIP_TRY(hipMemsetAsync(prof_buf, 0, 32 * sizeof(unsigned long long), st));
    ws.prof = prof_buf;
  }
#endif
  hipLaunchKernelGGL(roi_bwd_prep_kernel, dim3(static_cast<unsigned>(P.tab_blocks + ceil_div64(P.num_tiles, kPrepTiles))), dim3(kBlock), 0, st,
                     L, P, ws, rois, levels_in, K, PH, PW, sr);
  const size_t lds = sizeof(float) * (static_cast<size_t>(P.batch) * PH * PW * CT +
                                      static_cast<size_t>(P.batch) * (kGTH * P.PPH + kGTW * (P.PPW + 1)) +
                                      ((PH == PW) ? bin_strip_floats(PW, CT) : 0));
  const dim3 grid(static_cast<unsigned>(static_cast<int64_t>(P.num_tiles) * P.chunks));
#define BINNED_LAUNCH(PH_, PW_, CT_)                                                                              \
  hipLaunchKernelGGL((roi_align_bwd_binned_kernel<PH_, PW_, CT_>), grid, dim3(kBlock), lds, st, L, P, ws, gout, \
                     C, PH, PW)
  if (PH == 7 && PW == 7) { if (CT == 32) BINNED_LAUNCH(7, 7, 32); else BINNED_LAUNCH(7, 7, 16); }
  else if (PH == 14 && PW == 14) { if (CT == 32) BINNED_LAUNCH(14, 14, 32); else BINNED_LAUNCH(14, 14, 16); }
  else { if (CT == 32) BINNED_LAUNCH(0, 0, 32); else BINNED_LAUNCH(0, 0, 16); }
#undef BINNED_LAUNCH
#ifndef DETOPS_CPU_EMU
  if (ws.prof) {
    unsigned long long h[32];
    DETOPS_HIP_TRY(hipStreamSynchronize(st));
    DETOPS_HIP_TRY(hipMemcpy(h, ws.prof, sizeof(h), hipMemcpyDeviceToHost));
    static const char* cls[4] = {"0 hits", "1-4 hits", "5-16 hits", ">16 hits"};
    for (int c = 0; c < 4; ++c) {
      const double n = h[c * 8] ? static_cast<double>(h[c * 8]) : 1.0;
      fprintf(stderr, "[bwd-binned %dx%d] %-9s workgroups %6llu  hits/wg %6.1f  cycles/wg: total %8.0f  list %7.0f  stage %8.0f  walk %8.0f  store %7.0f\n",
              PH, PW, cls[c], h[c * 8], h[c * 8 + 6] / n, h[c * 8 + 1] / n, h[c * 8 + 2] / n, h[c * 8 + 3] / n, h[c * 8 + 4] / n, h[c * 8 + 5] / n);
    }
  }
#endif
  return launch_status();
}

// Pixel-owner backward launch.  Returns -1 when the shape does not fit its LDS plan (huge bin
// counts): the caller then uses the tile-scatter kernel.
int run_backward_gather(const Levels& L, const float* rois, const int32_t* levels_in, const float* gout,
                        int N, int C, int K, int PH, int PW, int sr, int accumulate, hipStream_t st) {
  if (C == 0 || N == 0) return 0;
  const int bins = PH * PW;
  const int PPH = (PH + 3) & ~3, PPW = (PW + 3) & ~3;
  auto count_items = [&](int ct) {
    int64_t items = 0;
    for (int i = 0; i < L.num; ++i)
      items += static_cast<int64_t>(N) * ceil_div64(L.lv[i].H, kGTH) * ceil_div64(L.lv[i].W, kGTW) * ceil_div64(C, ct);
    return items;
  };
  // 16 channels per workgroup unless that leaves the chip underfilled (small maps) or the staged
  // gradient block would not fit in LDS (bins > 256)
  int CT = 16;
  if (count_items(16) < 2 * kNumCU) CT = 4;
  if (const char* e = getenv("DETOPS_ROIALIGN_BWD_CT")) CT = (atoi(e) == 16) ? 16 : 4;  // tuning / test knob
  if (bins > kGBins) CT = 4;
  const int slots = max(bins, kGBins);
  const int batch = max(1, kGBins / bins);
  const size_t region = static_cast<size_t>(max(slots * CT, CT * kGTH * kGRowPad));
  const size_t lds = sizeof(float) * (region + static_cast<size_t>(batch) * (kGTH * PPH + kGTW * PPW));
  if (lds > 56 * 1024) return -1;
  for (int i = 0; i < L.num; ++i)
    if (L.lv[i].W > 32767) return -1;  // GHit::xspan packs two 15-bit column indices
  GPlan P{};
  P.chunks = static_cast<int>(ceil_div64(C, CT));
  P.accumulate = accumulate;
  P.batch = batch;
  // ROI-list split for underfilled launches: ~16 ROIs per workgroup, at most 32 groups
  // (cfg-1, 512 ROIs on one 14 x 14 map: 1037 -> 111 us at 7x7 bins, 4640 -> 374 us at 14x14)
  P.groups = 1;
  if (count_items(CT) < 2 * kNumCU && K > 64) P.groups = static_cast<int>(std::min<int64_t>(32, ceil_div64(K, 16)));
  if (const char* e = getenv("DETOPS_ROIALIGN_BWD_GROUPS")) P.groups = max(1, min(64, atoi(e)));  // tuning / test knob
  int64_t items = 0;
  for (int i = L.num - 1; i >= 0; --i) {  // coarsest level first
    P.tiles_x[i] = static_cast<int>(ceil_div64(L.lv[i].W, kGTW));
    P.tiles_y[i] = static_cast<int>(ceil_div64(L.lv[i].H, kGTH));
    const int64_t n = static_cast<int64_t>(N) * P.tiles_x[i] * P.tiles_y[i] * P.chunks;
    if (items + n > 0x7fffffff) return DETOPS_EUNSUPPORTED;
    P.first_item[i] = static_cast<int>(items);
    P.n_items[i] = static_cast<int>(n);
    items += n;
  }
  if (items == 0) return 0;
  if (P.groups > 1 && !accumulate)
    for (int i = 0; i < L.num; ++i)
      DETOPS_HIP_TRY(hipMemsetAsync(L.lv[i].gin, 0, sizeof(float) * static_cast<size_t>(N) * C * L.lv[i].H * L.lv[i].W, st));
  const dim3 grid(static_cast<unsigned>(items), static_cast<unsigned>(P.groups));
#define GATHER_LAUNCH(PH_, PW_, CT_)                                                                          \
  hipLaunchKernelGGL((roi_align_bwd_gather_kernel<PH_, PW_, CT_>), grid, dim3(kBlock), lds, st, L, P, rois, \
                     levels_in, gout, C, K, PH, PW, sr)
  if (PH == 7 && PW == 7) { if (CT == 16) GATHER_LAUNCH(7, 7, 16); else GATHER_LAUNCH(7, 7, 4); }
  else if (PH == 14 && PW == 14) { if (CT == 16) GATHER_LAUNCH(14, 14, 16); else GATHER_LAUNCH(14, 14, 4); }
  else { if (CT == 16) GATHER_LAUNCH(0, 0, 16); else GATHER_LAUNCH(0, 0, 4); }
#undef GATHER_LAUNCH
  return launch_status();
}

// Dispatch: the binned pixel-owner kernel when the caller supplies a workspace and the launch fills the
// chip; otherwise the scan pixel-owner kernel (small maps: ROI-list split); the LDS-scatter tile kernel for
// bin counts beyond both plans.  DETOPS_ROIALIGN_BWD = "tile" | "gather" (= scan) | "binned" forces one
// (A/B measurements, tests); read per call.
int run_backward(const Levels& L, const float* rois, const int32_t* levels_in, const float* gout,
                 int N, int C, int K, int PH, int PW, int sr, int accumulate, hipStream_t st,
                 void* workspace = nullptr, size_t workspace_bytes = 0) {
  const char* e = getenv("DETOPS_ROIALIGN_BWD");
  const bool force_tile = e && e[0] == 't';
  const bool force_scan = e && e[0] == 'g';
  if (!force_tile && !force_scan) {
    const bool forced = e && e[0] == 'b' && K > 0;
    const int rc = run_backward_binned(L, rois, levels_in, gout, N, C, K, PH, PW, sr, accumulate, workspace,
                                       workspace_bytes, forced, st);
    if (rc != -1) return rc;
    if (forced) return DETOPS_EWORKSPACE;   // "binned" was demanded but is not applicable
  }
  if (!force_tile) {
    const int rc = run_backward_gather(L, rois, levels_in, gout, N, C, K, PH, PW, sr, accumulate, st);
    if (rc != -1) return rc;
  }
  return run_backward_tiles(L, rois, levels_in, gout, N, C, K, PH, PW, sr, accumulate, st);
}

inline bool bad_dims(int N, int C, int K, int PH, int PW) {
  return N < 0 || C < 0 || K < 0 || PH <= 0 || PW <= 0;
}

}  // namespace

static int order_min_k() {
  if (const char* e = getenv("DETOPS_ROIALIGN_FWD_ORDER_MINK")) return max(2, atoi(e));   // A/B runs
  return kOrderMinK;
}

DETOPS_API size_t detops_roi_align_forward_workspace_bytes(int K) {
  if (K < order_min_k() || K > kOrderMaxK) return 0;
  return (sizeof(int32_t) * static_cast<size_t>(K) + 255) & ~static_cast<size_t>(255);
}

static int32_t* order_workspace(int K, void* workspace, size_t workspace_bytes) {
  const size_t need = detops_roi_align_forward_workspace_bytes(K);
  return (workspace && need && workspace_bytes >= need) ? static_cast<int32_t*>(workspace) : nullptr;
}

DETOPS_API int detops_roi_align_forward_ws_f32(const float* input, const float* rois, float* output,
                                               int N, int C, int H, int W, int K, int PH, int PW,
                                               float spatial_scale, int sampling_ratio, void* workspace,
                                               size_t workspace_bytes, detops_stream_t stream) {
  if (bad_dims(N, C, K, PH, PW) || H < 0 || W < 0) return DETOPS_EINVAL;
  if (K == 0 || C == 0) return 0;
  if (!input || !rois || !output || H == 0 || W == 0 || N == 0) return DETOPS_EINVAL;
  Levels L{};
  L.num = 1;
  L.lv[0] = Level{input, nullptr, H, W, spatial_scale};
  return run_forward(L, rois, nullptr, nullptr, output, C, K, PH, PW, sampling_ratio,
                     as_stream(stream), order_workspace(K, workspace, workspace_bytes));
}

DETOPS_API int detops_roi_align_forward_f32(const float* input, const float* rois, float* output,
                                            int N, int C, int H, int W, int K, int PH, int PW,
                                            float spatial_scale, int sampling_ratio,
                                            detops_stream_t stream) {
  return detops_roi_align_forward_ws_f32(input, rois, output, N, C, H, W, K, PH, PW, spatial_scale,
                                         sampling_ratio, nullptr, 0, stream);
}

DETOPS_API int detops_roi_align_backward_ws_f32(const float* grad_out, const float* rois,
                                                float* grad_in, int N, int C, int H, int W, int K,
                                                int PH, int PW, float spatial_scale,
                                                int sampling_ratio, int zero_grad_in, void* workspace,
                                                size_t workspace_bytes, detops_stream_t stream) {
  if (bad_dims(N, C, K, PH, PW) || H < 0 || W < 0) return DETOPS_EINVAL;
  const size_t bytes = sizeof(float) * static_cast<size_t>(N) * C * H * W;
  if (bytes == 0) return 0;
  if (!grad_in) return DETOPS_EINVAL;
  if (K > 0 && (!grad_out || !rois)) return DETOPS_EINVAL;
  if (K == 0 && !zero_grad_in) return 0;
  Levels L{};
  L.num = 1;
  L.lv[0] = Level{nullptr, grad_in, H, W, spatial_scale};
  return run_backward(L, rois, nullptr, grad_out, N, C, K, PH, PW, sampling_ratio,
                      zero_grad_in ? 0 : 1, as_stream(stream), workspace, workspace_bytes);
}

DETOPS_API int detops_roi_align_backward_f32(const float* grad_out, const float* rois,
                                             float* grad_in, int N, int C, int H, int W, int K,
                                             int PH, int PW, float spatial_scale,
                                             int sampling_ratio, int zero_grad_in,
                                             detops_stream_t stream) {
  return detops_roi_align_backward_ws_f32(grad_out, rois, grad_in, N, C, H, W, K, PH, PW, spatial_scale,
                                          sampling_ratio, zero_grad_in, nullptr, 0, stream);
}

DETOPS_API size_t detops_roi_align_backward_workspace_bytes(const int* H_host, const int* W_host,
                                                            int num_levels, int N, int C, int K, int PH,
                                                            int PW) {
  if (!H_host || !W_host || num_levels < 1 || num_levels > DETOPS_MAX_LEVELS || bad_dims(N, C, K, PH, PW) ||
      K == 0 || C == 0 || N == 0)
    return 0;
  Levels L{};
  L.num = num_levels;
  for (int i = 0; i < num_levels; ++i) {
    if (H_host[i] <= 0 || W_host[i] <= 0) return 0;
    L.lv[i] = Level{nullptr, nullptr, H_host[i], W_host[i], 1.f};
  }
  BinPlan P; BinLayout lay;
  if (!bin_plan(L, N, C, K, PH, PW, 16, P, lay)) return 0;
  return lay.total;
}

DETOPS_API int detops_roi_align_fpn_forward_ws_f32(
    const float* const* inputs_host, const int* H_host, const int* W_host, const float* scale_host,
    int num_levels, const float* rois, float* output, int32_t* levels_out, int N, int C, int K,
    int PH, int PW, int sampling_ratio, int k_min, int k_max, float canonical_scale,
    float canonical_level, float eps, void* workspace, size_t workspace_bytes, detops_stream_t stream) {
  if (bad_dims(N, C, K, PH, PW) || num_levels < 1 || num_levels > DETOPS_MAX_LEVELS ||
      !inputs_host || !H_host || !W_host || !scale_host)
    return DETOPS_EINVAL;
  if (k_max - k_min + 1 != num_levels) return DETOPS_EINVAL;
  if (K == 0 || C == 0) return 0;
  if (!rois || !output) return DETOPS_EINVAL;
  Levels L{};
  L.num = num_levels;
  L.k_min = k_min; L.k_max = k_max; L.s0 = canonical_scale; L.lvl0 = canonical_level; L.eps = eps;
  for (int i = 0; i < num_levels; ++i) {
    if (!inputs_host[i] || H_host[i] <= 0 || W_host[i] <= 0) return DETOPS_EINVAL;
    L.lv[i] = Level{inputs_host[i], nullptr, H_host[i], W_host[i], scale_host[i]};
  }
  hipStream_t st = as_stream(stream);
  if (num_levels == 1 && levels_out) DETOPS_HIP_TRY(hipMemsetAsync(levels_out, 0, sizeof(int32_t) * K, st));
  return run_forward(L, rois, nullptr, levels_out, output, C, K, PH, PW, sampling_ratio, st,
                     order_workspace(K, workspace, workspace_bytes));
}

DETOPS_API int detops_roi_align_fpn_forward_f32(
    const float* const* inputs_host, const int* H_host, const int* W_host, const float* scale_host,
    int num_levels, const float* rois, float* output, int32_t* levels_out, int N, int C, int K,
    int PH, int PW, int sampling_ratio, int k_min, int k_max, float canonical_scale,
    float canonical_level, float eps, detops_stream_t stream) {
  return detops_roi_align_fpn_forward_ws_f32(inputs_host, H_host, W_host, scale_host, num_levels, rois, output,
                                             levels_out, N, C, K, PH, PW, sampling_ratio, k_min, k_max,
                                             canonical_scale, canonical_level, eps, nullptr, 0, stream);
}

DETOPS_API int detops_roi_align_fpn_backward_ws_f32(
    const float* grad_out, const float* rois, const int32_t* levels, float* const* grad_inputs_host,
    const int* H_host, const int* W_host, const float* scale_host, int num_levels, int N, int C,
    int K, int PH, int PW, int sampling_ratio, int zero_grad_in, void* workspace, size_t workspace_bytes,
    detops_stream_t stream) {
  if (bad_dims(N, C, K, PH, PW) || num_levels < 1 || num_levels > DETOPS_MAX_LEVELS ||
      !grad_inputs_host || !H_host || !W_host || !scale_host)
    return DETOPS_EINVAL;
  Levels L{};
  L.num = num_levels;
  for (int i = 0; i < num_levels; ++i) {
    if (!grad_inputs_host[i] || H_host[i] <= 0 || W_host[i] <= 0) return DETOPS_EINVAL;
    L.lv[i] = Level{nullptr, grad_inputs_host[i], H_host[i], W_host[i], scale_host[i]};
  }
  if (C == 0 || N == 0) return 0;
  if (K > 0 && (!grad_out || !rois || (num_levels > 1 && !levels))) return DETOPS_EINVAL;
  if (K == 0 && !zero_grad_in) return 0;
  return run_backward(L, rois, levels, grad_out, N, C, K, PH, PW, sampling_ratio,
                      zero_grad_in ? 0 : 1, as_stream(stream), workspace, workspace_bytes);
}

DETOPS_API int detops_roi_align_fpn_backward_f32(
    const float* grad_out, const float* rois, const int32_t* levels, float* const* grad_inputs_host,
    const int* H_host, const int* W_host, const float* scale_host, int num_levels, int N, int C,
    int K, int PH, int PW, int sampling_ratio, int zero_grad_in, detops_stream_t stream) {
  return detops_roi_align_fpn_backward_ws_f32(grad_out, rois, levels, grad_inputs_host, H_host, W_host,
                                              scale_host, num_levels, N, C, K, PH, PW, sampling_ratio,
                                              zero_grad_in, nullptr, 0, stream);
}
