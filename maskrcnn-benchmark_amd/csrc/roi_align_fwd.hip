// roi_align_fwd.hip — ROIAlign forward for gfx950 (MI355X), fp32 NCHW.
//
// Replaces the reference's RoIAlignForward (maskrcnn_benchmark/csrc/cuda/ROIAlign_cuda.cu:64-122) and its CPU
// forward (csrc/cpu/ROIAlign_cpu.cpp:113-219) behind detops_roi_align_forward_* and the multi-level
// detops_roi_align_fpn_forward_* entry points (include/detops.h).
//
// Design (not a translation of the reference's one-thread-per-output kernel):
//   * one workgroup per (ROI, channel chunk).  The sampling geometry of a ROI is SEPARABLE: the y taps depend on
//     (ph, iy) only and the x taps on (pw, ix) only, so two small axis tables in LDS — PH*gh + PW*gw entries of
//     {low, high, frac, 1-frac} — replace re-deriving 4 indices + 4 weights per output element per channel.
//   * the per-sample arithmetic keeps the reference's evaluation order with FP contraction off
//     (w = hy*hx ...; val = w1*v1 + w2*v2 + w3*v3 + w4*v4; acc += val; acc /= count): bit-identical to the
//     reference CPU kernel for finite inputs.
//   * fast path (fixed 1x1 / 2x2 sampling, 7x7 / 14x14 bins): the ROI footprint is staged in LDS by LDS-DMA
//     (global_load_lds_dwordx4, double-buffered), a thread keeps its bin's sample offsets + weights in registers
//     across the channel loop; ROIs are visited in (level, image, position) rank order (roi_fwd_prep_kernel) so that
//     overlapping footprints are fetched while still in the XCD's L2.
//   * everything else (adaptive sampling, other bin counts): roi_align_fwd_kernel, direct gathers.
#include "roi_align_common.h"

namespace {

constexpr int kTabBig = 512;    // axis-table entries per axis kept in LDS (adaptive grids);
constexpr int kTabSmall = 32;   // fixed sampling_ratio: PH*sr, PW*sr <= 32 covers 7x7..14x14 @ sr 2

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
//   * the ROI's footprint [channels x (py+1) x (px+1)] is staged into LDS (each feature byte under the ROI
//     crosses the memory system once per workgroup, as 16-byte row pieces);
//   * a thread owns ONE output bin for a strided set of channels: its SR*SR samples' patch offsets and 4
//     bilinear weights live in registers for the whole channel loop;
//   * taps are addressed as (lo, lo+1): the patch carries one extra row/column that replicates the clamped
//     border pixel, which is exactly what the reference reads when x_high == x_low (weight 0 on that tap);
//   * reference operation order, FP contraction off: bit-identical to the reference CPU kernel.
// Footprints that do not fit the LDS budget fall through to direct gathers inside the same kernel.
// ------------------------------------------------------------------------------------------
constexpr int kFwdDmaWps = 5;  // waves per SIMD the register budget must allow (no staging registers)

// Per-ROI sample records of the fast path (workspace): what EVERY channel-chunk workgroup of a ROI used to re-derive
// (geometry, 2 x PH*SR axis taps by the reference arithmetic, the footprint bounds, per bin 4 patch offsets + 16
// bilinear weights: ~350 VALU instructions per wave, 8 chunk workgroups per ROI = 40 % of the launch's VALU work)
// is computed ONCE per ROI by one wave of the pre-pass and loaded by the main kernel.
//   hdr  [K][2] int4   {level, image, ymin, rows | valid << 16}, {xmin, w4, 0, 0}
//   rec  [K][5][BINS] float4: j = 0 patch offsets of the bin's SR*SR samples (int), j = 1..4 the weights w1..w4
struct FwdRecs {
  int4* hdr;
  float4* rec;
};

constexpr int kPrepWaves = kOrderBlock / kWave;   // ROIs per record-role workgroup (one wave each)

template <int PH, int PW, int SR>
__device__ __forceinline__ void roi_record_role(const Levels& L, const float* __restrict__ rois,
                                                const int32_t* __restrict__ levels_in,
                                                int32_t* __restrict__ levels_out, int K, FwdRecs R, Tap* tabs, int block) {
  constexpr int BINS = PH * PW, NS = SR * SR;
  static_assert(PH * SR <= 32 && PW * SR <= 32 && NS <= 4, "one wave builds both axis tables at once");
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int k = block * kPrepWaves + wave;
  if (k >= K) return;
  Tap* tabY = tabs + wave * 64;
  Tap* tabX = tabY + 32;
  const float* roi = rois + static_cast<size_t>(k) * 5;
  int lvl = 0;
  if (L.num > 1) lvl = levels_in ? levels_in[k] : fpn_level(roi, L);
  if (levels_out && lane == 0) levels_out[k] = lvl;
  int H = L.lv[0].H, W = L.lv[0].W; float scale = L.lv[0].scale;
#pragma unroll
  for (int i = 1; i < DETOPS_MAX_LEVELS; ++i)
    if (i == lvl) { H = L.lv[i].H; W = L.lv[i].W; scale = L.lv[i].scale; }
  const RoiGeom g = roi_geometry(roi, scale, PH, PW, SR);
  // lanes 0..31: y taps, lanes 32..63: x taps
  const bool isy = lane < 32;
  const int u = lane & 31;
  const int nent = isy ? PH * SR : PW * SR;
  int lo_min = 0x7fffffff, lo_max = -1;
  if (u < nent) {
    const Tap e = isy ? axis_entry(g.start_h, g.bin_h, u / SR, u % SR, SR, H, 1)
                      : axis_entry(g.start_w, g.bin_w, u / SR, u % SR, SR, W, 1);
    (isy ? tabY : tabX)[u] = e;
    if (e.l != 0.f || e.h != 0.f) { lo_min = e.lo; lo_max = e.lo; }
  }
#pragma unroll
  for (int off = 16; off >= 1; off >>= 1) {   // min / max within each half of the wave
    lo_min = min(lo_min, __shfl(lo_min, lane ^ off));
    lo_max = max(lo_max, __shfl(lo_max, lane ^ off));
  }
  const int ymin = __shfl(lo_min, 0), ymax = __shfl(lo_max, 0);
  const int xmin = __shfl(lo_min, 32), xmax = __shfl(lo_max, 32);
  DETOPS_WAVE_SYNC();   // the tables are read by other lanes of this wave
  const bool valid = ymax >= 0 && xmax >= 0;
  const int rows = ymax - ymin + 2;
  int w4 = (xmax - xmin + 2 + 3) >> 2;
  w4 |= 1;
  const int ps = 4 * w4;
  if (lane == 0) {
    R.hdr[2 * static_cast<size_t>(k)] = make_int4(lvl, g.b, ymin, (valid ? rows : 0) | ((valid ? 1 : 0) << 16));
    R.hdr[2 * static_cast<size_t>(k) + 1] = make_int4(xmin, w4, 0, 0);
    DETOPS_STAT("fwd.records", 1);
  }
  if (!valid) return;
  float4* rec = R.rec + static_cast<size_t>(k) * 5 * BINS;
  for (int bin = lane; bin < BINS; bin += kWave) {
    const int ph = bin / PW, pw = bin - ph * PW;
    int off[4] = {0, 0, 0, 0};
    float w1[4] = {0.f, 0.f, 0.f, 0.f}, w2[4] = {0.f, 0.f, 0.f, 0.f}, w3[4] = {0.f, 0.f, 0.f, 0.f}, w4s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int iy = 0; iy < SR; ++iy) {
      const Tap ty = tabY[ph * SR + iy];
      const bool vy = (ty.l != 0.f || ty.h != 0.f);
#pragma unroll
      for (int ix = 0; ix < SR; ++ix) {
#pragma clang fp contract(off)
        const Tap tx = tabX[pw * SR + ix];
        const bool vv = vy && (tx.l != 0.f || tx.h != 0.f);
        const int s_ = iy * SR + ix;
        off[s_] = vv ? (ty.lo - ymin) * ps + (tx.lo - xmin) : 0;
        w1[s_] = vv ? ty.h * tx.h : 0.f;
        w2[s_] = vv ? ty.h * tx.l : 0.f;
        w3[s_] = vv ? ty.l * tx.h : 0.f;
        w4s[s_] = vv ? ty.l * tx.l : 0.f;
      }
    }
    rec[0 * BINS + bin] = make_float4(__int_as_float(off[0]), __int_as_float(off[1]), __int_as_float(off[2]), __int_as_float(off[3]));
    rec[1 * BINS + bin] = make_float4(w1[0], w1[1], w1[2], w1[3]);
    rec[2 * BINS + bin] = make_float4(w2[0], w2[1], w2[2], w2[3]);
    rec[3 * BINS + bin] = make_float4(w3[0], w3[1], w3[2], w3[3]);
    rec[4 * BINS + bin] = make_float4(w4s[0], w4s[1], w4s[2], w4s[3]);
  }
}

// The forward's pre-pass, ONE launch, two roles by block index: blocks [0, order_blocks) rank the ROIs (visiting
// order, L2 locality), the rest build the per-ROI sample records (one wave per ROI).
template <int PH, int PW, int SR>
__global__ void __launch_bounds__(kOrderBlock)
roi_fwd_prep_kernel(Levels L, const float* __restrict__ rois, const int32_t* __restrict__ levels_in,
                    int32_t* __restrict__ levels_out, int K, int32_t* __restrict__ order, int order_blocks, FwdRecs R) {
  constexpr size_t kKeyBytes = sizeof(unsigned long long) * (kOrderMaxK + 2 * kOrderLanes);
  constexpr size_t kTabBytes = sizeof(Tap) * 64 * kPrepWaves;
  __shared__ __attribute__((aligned(16))) unsigned char smem[kKeyBytes > kTabBytes ? kKeyBytes : kTabBytes];
  if (static_cast<int>(blockIdx.x) < order_blocks)
    roi_order_role(L, rois, levels_in, K, order, reinterpret_cast<unsigned long long*>(smem), static_cast<int>(blockIdx.x));
  else
    roi_record_role<PH, PW, SR>(L, rois, levels_in, levels_out, K, R, reinterpret_cast<Tap*>(smem),
                                static_cast<int>(blockIdx.x) - order_blocks);
}

template <int PH, int PW, int SR, int G, int WPS, int MODE>
__global__ void __launch_bounds__(((PH * PW * G + 63) / 64) * 64, WPS)
roi_align_fwd_dma_kernel(Levels L, const float* __restrict__ rois, const int32_t* __restrict__ levels_in,
                         int32_t* __restrict__ levels_out, float* __restrict__ out, int C, int K, int CT,
                         int chunks, int buf_floats, const int32_t* __restrict__ order, FwdRecs R) {
  constexpr int BINS = PH * PW;
  constexpr int NS = SR * SR;
  constexpr int NT = ((BINS * G + 63) / 64) * 64;
  constexpr bool PRE = MODE >= 1;            // per-ROI sample records from the pre-pass
  constexpr bool STATIC_ISSUE = MODE >= 2;   // staging offsets fixed per lane for the whole workgroup (see below)
  constexpr int kMaxPass = 3;                // staging passes per batch the static form unrolls
  DETOPS_DYNAMIC_LDS(float, patch);          // two buffers of buf_floats (multiple of 256 floats = one wave-instruction)
  __shared__ Tap tabY[PH * SR];
  __shared__ Tap tabX[PW * SR];
  __shared__ int s_bounds[4];

  const int tid = threadIdx.x;
  const int bid = blockIdx.x;
  const int kk = bid / chunks;
  const int chunk = bid - kk * chunks;
  const int k = order ? order[kk] : kk;
  const int c0 = chunk * CT;
  const int cend = min(C, c0 + CT);
  float* obase = out + (static_cast<size_t>(k) * C + c0) * BINS;
  int lvl = 0, roi_b = 0, ymin = 0, rows = 0, xmin = 0, w4 = 0;
  bool any_inside = true;
  if constexpr (PRE) {   // everything per ROI comes from the pre-pass (roi_record_role): wave-uniform scalar loads
    const int4 h0 = R.hdr[2 * static_cast<size_t>(k)], h1 = R.hdr[2 * static_cast<size_t>(k) + 1];
    lvl = __builtin_amdgcn_readfirstlane(h0.x); roi_b = __builtin_amdgcn_readfirstlane(h0.y);
    ymin = __builtin_amdgcn_readfirstlane(h0.z);
    const int rv = __builtin_amdgcn_readfirstlane(h0.w);
    rows = rv & 0xffff; any_inside = (rv >> 16) != 0;
    xmin = __builtin_amdgcn_readfirstlane(h1.x); w4 = __builtin_amdgcn_readfirstlane(h1.y);
  } else {
    const float* roi = rois + static_cast<size_t>(k) * 5;
    if (L.num > 1) lvl = levels_in ? levels_in[k] : fpn_level(roi, L);
    if (levels_out && chunk == 0 && tid == 0) levels_out[k] = lvl;
  }
  const float* in = L.lv[0].in; int H = L.lv[0].H, W = L.lv[0].W; float scale = L.lv[0].scale;
#pragma unroll
  for (int i = 1; i < DETOPS_MAX_LEVELS; ++i)
    if (i == lvl) { in = L.lv[i].in; H = L.lv[i].H; W = L.lv[i].W; scale = L.lv[i].scale; }

  if constexpr (!PRE) {
    const float* roi = rois + static_cast<size_t>(k) * 5;
    const RoiGeom g = roi_geometry(roi, scale, PH, PW, SR);
    roi_b = g.b;
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
    any_inside = !(s_bounds[1] < 0 || s_bounds[3] < 0);
    if (any_inside) {
      ymin = s_bounds[0];
      rows = s_bounds[1] - ymin + 2;   // (lo range) + the lo+1 row
      xmin = s_bounds[2];
      w4 = (s_bounds[3] - xmin + 2 + 3) >> 2;   // float4 pieces per patch row incl. the lo+1 column
      w4 |= 1;                                   // odd: rows spread over the LDS banks
    }
  }
  (void)scale;
  if (!any_inside) {  // every sample falls outside the map: all-zero output
    for (int o = tid; o < (cend - c0) * BINS; o += NT) obase[o] = 0.f;
    return;
  }
  const size_t plane = static_cast<size_t>(H) * W;
  const float* base = in + (static_cast<size_t>(roi_b) * C + c0) * plane;
  const int ps = 4 * w4;
  const int a4 = rows * w4;                      // float4 pieces per channel
  const int area = 4 * a4;

  if (area > buf_floats) {  // footprint too large for LDS: gather straight from the map
    if constexpr (PRE) {    // the records hold patch-relative offsets: rebuild the axis tables for the direct gathers
      const RoiGeom g = roi_geometry(rois + static_cast<size_t>(k) * 5, scale, PH, PW, SR);
      if (tid < PH * SR) tabY[tid] = axis_entry(g.start_h, g.bin_h, tid / SR, tid % SR, SR, H, 1);
      else if (tid < PH * SR + PW * SR) { const int u = tid - PH * SR; tabX[u] = axis_entry(g.start_w, g.bin_w, u / SR, u % SR, SR, W, 1); }
      __syncthreads();
    }
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
      obase[o] = acc / static_cast<float>(NS);
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
  // Static form: every batch has the same channel count and footprint, so piece p = tid + i * NT of a batch maps to the
  // same (channel, row, column piece) in EVERY batch: its source offset relative to the batch's first plane is computed
  // once; per batch only the scalar base moves (global_load_lds with saddr + per-lane offset, no vector arithmetic
  // per piece — the incremental form below costs ~12 VALU instructions per piece and pass).  Completion is counted
  // by hand (DETOPS_VMCNT_WAIT(0) in front of the batch barrier).
  unsigned so_off[kMaxPass];
  int so_slow = 0;                           // bit i: piece tid + i * NT reaches beyond the map (register path)
  const bool use_static = STATIC_ISSUE && ctb * a4 <= kMaxPass * NT && static_cast<size_t>(ctb + 1) * plane < 0x3fffffffu;
  if (use_static) {
#pragma unroll
    for (int i = 0; i < kMaxPass; ++i) {
      const int p = tid + i * NT;
      const int c = static_cast<int>((static_cast<float>(p) + 0.5f) * inv_a4);
      const int rem = p - c * a4;
      const int y = static_cast<int>((static_cast<float>(rem) + 0.5f) * inv_w4);
      const int v = rem - y * w4;
      const int gx = xmin + 4 * v;
      if (!(gx + 3 < W && ymin + y < H)) so_slow |= 1 << i;
      so_off[i] = static_cast<unsigned>(c * static_cast<int>(plane) + (ymin + y) * W + gx) * 4u;
    }
  }
  auto issue = [&](int cs, int cn, float* buf) {
    const float* src = base + static_cast<size_t>(cs - c0) * plane;
    const int total = cn * a4;
    if (use_static) {
#pragma unroll
      for (int i = 0; i < kMaxPass; ++i)
        if (i * NT < total)                         // wave-uniform
          glds16_async_so(tid + i * NT < total && !((so_slow >> i) & 1), src, so_off[i], buf + 4 * i * NT + wave_off);
      if (so_slow) {                                // pieces reaching beyond the map: replicated border column / row
#pragma unroll 1
        for (int i = 0; i < kMaxPass; ++i) {
          const int p = tid + i * NT;
          if (((so_slow >> i) & 1) && p < total) {
            const int c = static_cast<int>((static_cast<float>(p) + 0.5f) * inv_a4);
            const int rem = p - c * a4;
            const int y = static_cast<int>((static_cast<float>(rem) + 0.5f) * inv_w4);
            const int gx = xmin + 4 * (rem - y * w4);
            const float* rowp = src + static_cast<size_t>(c) * plane + static_cast<size_t>(min(ymin + y, H - 1)) * W;
            *reinterpret_cast<float4*>(buf + 4 * p) =
                make_float4(rowp[min(gx, W - 1)], rowp[min(gx + 1, W - 1)], rowp[min(gx + 2, W - 1)], rowp[min(gx + 3, W - 1)]);
          }
        }
      }
      return;
    }
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
  if constexpr (PRE) {
    const float4* rec = R.rec + static_cast<size_t>(k) * 5 * BINS + bin;
    const float4 r0 = rec[0], r1 = rec[BINS], r2 = rec[2 * BINS], r3 = rec[3 * BINS], r4 = rec[4 * BINS];
    const float ro[4] = {r0.x, r0.y, r0.z, r0.w}, ra[4] = {r1.x, r1.y, r1.z, r1.w}, rb[4] = {r2.x, r2.y, r2.z, r2.w};
    const float rc[4] = {r3.x, r3.y, r3.z, r3.w}, rd[4] = {r4.x, r4.y, r4.z, r4.w};
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_) { off[s_] = __float_as_int(ro[s_]); w1[s_] = ra[s_]; w2[s_] = rb[s_]; w3[s_] = rc[s_]; w4s[s_] = rd[s_]; }
  } else {
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
  }
  (void)ph; (void)pw;
  const float inv_count = 1.f / static_cast<float>(NS);  // NS in {1,4}: exact reciprocal

  issue(c0, min(ctb, cend - c0), patch);
  int n = 0;
  for (int cs = c0; cs < cend; cs += ctb, ++n) {
    const int cn = min(ctb, cend - cs);
    if (tid == 0) { DETOPS_STAT("fwd.stage_batches", 1); DETOPS_STAT("fwd.staged_floats", cn * area); }
    // this wave's LDS-DMA pieces of batch n have landed BEFORE the barrier — explicitly: the asm-issued form is invisible
    // to the compiler, and for the builtin form hipcc only waits in front of the issuing wave's own LDS reads (seen on
    // the device: `s_waitcnt lgkmcnt(0); s_barrier` here, another wave's pieces still in flight behind the barrier)
    DETOPS_VMCNT_WAIT(0);
    __syncthreads();   // batch n has landed everywhere; every wave is done reading the other buffer
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
// Channel chunk per workgroup of the generic kernel: enough workgroups to fill 256 CUs several times over while
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

// Workspace of the fast path: [order: K int32][hdr: 2 K int4][rec: 5 K BINS float4], each part 256-byte aligned.
struct FwdWs {
  int32_t* order;   // nullptr: no ranking for this K
  FwdRecs recs;     // hdr == nullptr: no records (workspace absent / too small)
};

int order_min_k() {
  const int v = detops_tuning().roi_fwd_order_mink;
  return v > 0 ? max(2, v) : kOrderMinK;
}

inline bool fwd_dma_shape(int PH, int PW, int sr) {
  return (PH == 7 && PW == 7 && (sr == 1 || sr == 2)) || (PH == 14 && PW == 14 && sr == 2);
}

inline size_t up256(size_t v) { return (v + 255) & ~static_cast<size_t>(255); }

struct FwdWsLayout { size_t off_hdr, off_rec, total; bool order, recs; };

inline FwdWsLayout fwd_ws_layout(int K, int PH, int PW, int sr) {
  FwdWsLayout l{};
  l.order = K >= order_min_k() && K <= kOrderMaxK;
  l.recs = K > 0 && fwd_dma_shape(PH, PW, sr) && detops_tuning().roi_fwd_records != 1;
  size_t o = l.order ? up256(sizeof(int32_t) * static_cast<size_t>(K)) : 0;
  l.off_hdr = o;
  if (l.recs) o = up256(o + sizeof(int4) * 2 * static_cast<size_t>(K));
  l.off_rec = o;
  if (l.recs) o = up256(o + sizeof(float4) * 5 * static_cast<size_t>(K) * PH * PW);
  l.total = o;
  return l;
}

inline FwdWs fwd_workspace(int K, int PH, int PW, int sr, void* workspace, size_t workspace_bytes) {
  FwdWs w{};
  const FwdWsLayout l = fwd_ws_layout(K, PH, PW, sr);
  if (!workspace || l.total == 0 || workspace_bytes < l.total) return w;
  unsigned char* base = static_cast<unsigned char*>(workspace);
  if (l.order) w.order = reinterpret_cast<int32_t*>(base);
  if (l.recs) { w.recs.hdr = reinterpret_cast<int4*>(base + l.off_hdr); w.recs.rec = reinterpret_cast<float4*>(base + l.off_rec); }
  return w;
}

template <int PH, int PW, int SR, int G>
void launch_fwd_dma(const Levels& L, const float* rois, const int32_t* levels_in, int32_t* levels_out,
                    float* out, int C, int K, hipStream_t st, const FwdWs& ws) {
  constexpr int NT = ((PH * PW * G + 63) / 64) * 64;
  // channels per workgroup.  32: with C = 256 the 8 chunks of a ROI land on the 8 XCDs (workgroup b runs on XCD
  // b % 8), so every L2 caches ONE 32-channel slice of the maps and sees all ROIs in ranked order; and 8 K short
  // workgroups balance better over 256 x 7 slots than 4 K long ones (84 vs 100 us, profiles/r02g_*)
  int64_t map_pixels = 0;
  for (int i = 0; i < L.num; ++i) map_pixels += static_cast<int64_t>(L.lv[i].H) * L.lv[i].W;
  // small maps = small footprints: the per-workgroup setup dominates, fewer and fatter workgroups win (cfg-1: 63 vs 86 us)
  int CT = (map_pixels < 16384) ? 64 : 32;
  if (detops_tuning().roi_fwd_ct) CT = max(8, min(256, detops_tuning().roi_fwd_ct));   // A/B
  while (CT > 16 && static_cast<int64_t>(K) * ceil_div64(C, CT) < 4 * kNumCU) CT >>= 1;
  if (CT > C) CT = C;
  const int chunks = static_cast<int>(ceil_div64(C, CT));
  const dim3 grid(static_cast<unsigned>(K) * chunks);
  // two buffers, each a whole number of 1 KiB wave-instructions, sized so that LDS allows as many workgroups per
  // CU as the registers do (68 VGPRs -> 7 waves / SIMD = 28 waves / CU): 7 x (2 x 11 KB) for the 4-wave 7x7
  // kernel, 4 x (2 x 19 KB) for the 7-wave 14x14 kernel — the measured optimum of both sweeps
  const int wgs = max(1, 28 / (NT / 64));
  const int buf_floats = max(4, (160 * 1024 / wgs - 768) / 2048) * 256;
  const size_t lds2 = 2 * static_cast<size_t>(buf_floats) * sizeof(float);
  const int32_t* order = nullptr;
  // ws.order != nullptr implies K >= order_min_k(); maps that fit an L2 slice several times over need no ranking
  const int order_mode = detops_tuning().roi_fwd_order;   // 0 auto, 1 off, 2 force (tests: rank even for tiny maps)
  unsigned og = 0;
  if (ws.order && order_mode != 1 && K <= kOrderMaxK && (map_pixels * C * 4 > (2 << 20) || order_mode == 2)) {
    og = static_cast<unsigned>(ceil_div64(static_cast<int64_t>(K) * kOrderLanes, kOrderBlock));
    order = ws.order;
  }
  const bool pre = ws.recs.hdr != nullptr;
  const unsigned rg = pre ? static_cast<unsigned>(ceil_div64(K, kPrepWaves)) : 0u;
  if (og + rg > 0)
    hipLaunchKernelGGL((roi_fwd_prep_kernel<PH, PW, SR>), dim3(og + rg), dim3(kOrderBlock), 0, st, L, rois, levels_in,
                       levels_out, K, ws.order, static_cast<int>(og), ws.recs);
#define FWD_LAUNCH(MODE_) hipLaunchKernelGGL((roi_align_fwd_dma_kernel<PH, PW, SR, G, kFwdDmaWps, MODE_>), grid, dim3(NT), lds2, st, L, rois, levels_in, \
                                             levels_out, out, C, K, CT, chunks, buf_floats, order, ws.recs)
  if (!pre) FWD_LAUNCH(0);
  else if (detops_tuning().roi_fwd_records == 2) FWD_LAUNCH(1);   // records, incremental staging offsets (A/B)
  else FWD_LAUNCH(2);
#undef FWD_LAUNCH
}

int run_forward(const Levels& L, const float* rois, const int32_t* levels_in, int32_t* levels_out,
                float* out, int C, int K, int PH, int PW, int sr, hipStream_t st, const FwdWs& ws = FwdWs{}) {
  if (K == 0 || C == 0) return 0;
  if (detops_tuning().roi_fwd_impl == 1) {   // generic kernel forced (tests, A/B)
  } else if (PH == 7 && PW == 7 && sr == 2) {
    launch_fwd_dma<7, 7, 2, 5>(L, rois, levels_in, levels_out, out, C, K, st, ws);
    return launch_status();
  } else if (PH == 14 && PW == 14 && sr == 2) {
    launch_fwd_dma<14, 14, 2, 2>(L, rois, levels_in, levels_out, out, C, K, st, ws);
    return launch_status();
  } else if (PH == 7 && PW == 7 && sr == 1) {
    launch_fwd_dma<7, 7, 1, 5>(L, rois, levels_in, levels_out, out, C, K, st, ws);
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

}  // namespace

DETOPS_API size_t detops_roi_align_forward_workspace_bytes(int K, int PH, int PW, int sampling_ratio) {
  if (K <= 0 || PH <= 0 || PW <= 0) return 0;
  return fwd_ws_layout(K, PH, PW, sampling_ratio).total;
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
                     as_stream(stream), fwd_workspace(K, PH, PW, sampling_ratio, workspace, workspace_bytes));
}

DETOPS_API int detops_roi_align_forward_f32(const float* input, const float* rois, float* output,
                                            int N, int C, int H, int W, int K, int PH, int PW,
                                            float spatial_scale, int sampling_ratio,
                                            detops_stream_t stream) {
  return detops_roi_align_forward_ws_f32(input, rois, output, N, C, H, W, K, PH, PW, spatial_scale,
                                         sampling_ratio, nullptr, 0, stream);
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
                     fwd_workspace(K, PH, PW, sampling_ratio, workspace, workspace_bytes));
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
