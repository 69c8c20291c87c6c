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
//     across the channel loop; ROIs are visited in (level, image, position) rank order (roi_order_kernel) so that
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

template <int PH, int PW, int SR, int G>
void launch_fwd_dma(const Levels& L, const float* rois, const int32_t* levels_in, int32_t* levels_out,
                    float* out, int C, int K, hipStream_t st, int32_t* order_ws) {
  constexpr int NT = ((PH * PW * G + 63) / 64) * 64;
  // channels per workgroup.  32: with C = 256 the 8 chunks of a ROI land on the 8 XCDs (workgroup b runs on XCD
  // b % 8), so every L2 caches ONE 32-channel slice of the maps and sees all ROIs in ranked order; and 8 K short
  // workgroups balance better over 256 x 7 slots than 4 K long ones (84 vs 100 us, profiles/r02g_*)
  int64_t map_pixels = 0;
  for (int i = 0; i < L.num; ++i) map_pixels += static_cast<int64_t>(L.lv[i].H) * L.lv[i].W;
  // small maps = small footprints: the per-workgroup setup dominates, fewer and fatter workgroups win (cfg-1: 63 vs 86 us)
  int CT = (map_pixels < 16384) ? 64 : 32;
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
  // order_ws != nullptr implies K >= order_min_k(); maps that fit an L2 slice several times over need no ranking
  const int order_mode = detops_tuning().roi_fwd_order;   // 0 auto, 1 off, 2 force (tests: rank even for tiny maps)
  if (order_ws && order_mode != 1 && K <= kOrderMaxK && (map_pixels * C * 4 > (2 << 20) || order_mode == 2)) {
    const unsigned og = static_cast<unsigned>(ceil_div64(static_cast<int64_t>(K) * kOrderLanes, kOrderBlock));
    hipLaunchKernelGGL(roi_order_kernel, dim3(og), dim3(kOrderBlock), 0, st, L, rois, levels_in, K, order_ws);
    order = order_ws;
  }
  hipLaunchKernelGGL((roi_align_fwd_dma_kernel<PH, PW, SR, G, kFwdDmaWps>), grid, dim3(NT), lds2, st, L, rois, levels_in,
                     levels_out, out, C, K, CT, chunks, buf_floats, order);
}

int run_forward(const Levels& L, const float* rois, const int32_t* levels_in, int32_t* levels_out,
                float* out, int C, int K, int PH, int PW, int sr, hipStream_t st, int32_t* order_ws = nullptr) {
  if (K == 0 || C == 0) return 0;
  if (detops_tuning().roi_fwd_impl == 1) {   // generic kernel forced (tests, A/B)
  } else if (PH == 7 && PW == 7 && sr == 2) {
    launch_fwd_dma<7, 7, 2, 5>(L, rois, levels_in, levels_out, out, C, K, st, order_ws);
    return launch_status();
  } else if (PH == 14 && PW == 14 && sr == 2) {
    launch_fwd_dma<14, 14, 2, 2>(L, rois, levels_in, levels_out, out, C, K, st, order_ws);
    return launch_status();
  } else if (PH == 7 && PW == 7 && sr == 1) {
    launch_fwd_dma<7, 7, 1, 5>(L, rois, levels_in, levels_out, out, C, K, st, order_ws);
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

int order_min_k() {
  const int v = detops_tuning().roi_fwd_order_mink;
  return v > 0 ? max(2, v) : kOrderMinK;
}

int32_t* order_workspace(int K, void* workspace, size_t workspace_bytes) {
  const size_t need = detops_roi_align_forward_workspace_bytes(K);
  return (workspace && need && workspace_bytes >= need) ? static_cast<int32_t*>(workspace) : nullptr;
}

}  // namespace

DETOPS_API size_t detops_roi_align_forward_workspace_bytes(int K) {
  if (K < order_min_k() || K > kOrderMaxK) return 0;
  return (sizeof(int32_t) * static_cast<size_t>(K) + 255) & ~static_cast<size_t>(255);
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
