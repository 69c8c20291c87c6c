// roi_align_nhwc.hip — ROIAlign over a CHANNELS-LAST feature pyramid for gfx950 (MI355X), fp32.
//
// The same operator as roi_align_fwd.hip / roi_align_bwd.hip (reference: maskrcnn_benchmark/csrc/cuda/ROIAlign_cuda.cu:64-122
// forward, :125-254 backward; csrc/cpu/ROIAlign_cpu.cpp:113-219; modeling/poolers.py:91-121 for the multi-level form), for
// feature maps stored [N, H, W, C] — what MIOpen's implicit-GEMM convolutions read and write natively on this chip, so a
// detector whose backbone runs channels-last needs no layout transposes around its pooler.
//
// Why a separate formulation and not a flag on the NCHW kernels: in NHWC a bilinear tap is ONE contiguous channel vector
// (1 KiB at C = 256).  The natural mapping is therefore lane = channel(s), and everything that depends on the ROI — sample
// coordinates, bilinear weights, which bins reach which pixel — is wave-uniform: it lives in scalar registers / LDS
// broadcast reads, control flow never diverges, and every memory access of a wave is one fully coalesced piece.
//
//   forward   one workgroup per (ROI, bin group); a wave owns whole bins, a lane 4 consecutive channels (float4 loads of
//             the 4 taps of every sample straight from global memory — no LDS staging: a tap IS a coalesced 1 KiB read).
//             Reference operation order with FP contraction off (w = hy*hx ...; val = w1*v1 + w2*v2 + w3*v3 + w4*v4;
//             acc += val; acc /= count): bit-identical to the reference CPU kernel for finite inputs.  Output either
//             [K, C, PH, PW] (the box head's FC layer and every checkpoint keep their layout: the workgroup assembles the
//             ROI's C x bins block in LDS and writes it as one contiguous 50 KB piece) or channels-last [K, PH, PW, C]
//             (the mask head's convolutions: direct coalesced stores).
//   backward  pixel-owner and atomic-free like the NCHW ring kernel, but transposed and HIT-PARALLEL: a workgroup owns a
//             4 x 8 pixel tile x 64 channels; a LANE IS ONE CHANNEL with the tile's 32 pixel sums in registers; each of the
//             4 waves takes every 4th ROI that reaches the tile ("hit", ascending ROI index) and walks the separable adjoint
//                 grad_in[y, x, c] += sum_ph AY[y, ph] * ( sum_pw AX[x, pw] * grad_out[r, c, ph, pw] )
//             for the WHOLE tile with scalar trip counts taken from the ROI's compact adjoint rows (exact zero skipping, the
//             x pass of a bin row shared by the tile's pixel rows).  No barrier inside the hit loop — a wave stages its own hit
//             (the [channel][bin] block of an NCHW gradient by LDS-DMA, 12.5 KB contiguous, read back conflict-free because
//             the bin count is odd; a channels-last gradient is read as coalesced channel vectors) — so a crowded tile's
//             chain is a quarter as long as its hit list, and the waves of a CU hide each other's staging latency.  The
//             four partial tiles are added in wave order through LDS (deterministic); every gradient-map element is
//             written exactly once (zero tiles included): no zero-fill pass, no atomics.
#include "roi_align_common.h"

namespace {

// ------------------------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------------------------
constexpr int kNfBlockNhwc = 256;  // channels-last output (no LDS tile): 4 waves, many workgroups per CU
constexpr int kNfBlockNchw = 1024; // [K, C, PH, PW] output: the 50 KB LDS tile caps the workgroups per CU — 16 waves each keep the CU's 32 wave slots full (measured 74 / 58 / 50 us at 256 / 512 / 1024 threads: the kernel is latency-bound)
constexpr int kNfTab = 64;        // axis-table entries per axis kept in LDS (fixed sampling: PH * sr <= 64)
constexpr int kNfGroupBins = 49;  // bins per workgroup (the 7 x 7 box head in one group)

template <int V> struct NVec;
template <> struct NVec<4> { typedef float4 type; };
template <> struct NVec<1> { typedef float type; };

template <int V> __device__ __forceinline__ void nv_load(float (&d)[V], const float* p);
template <> __device__ __forceinline__ void nv_load<4>(float (&d)[4], const float* p) {
  const float4 v = *reinterpret_cast<const float4*>(p);
  d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
}
template <> __device__ __forceinline__ void nv_load<1>(float (&d)[1], const float* p) { d[0] = *p; }

// Ranking launch of the forward (visiting order for L2 locality, as in the NCHW kernel: roi_order_role)
__global__ void __launch_bounds__(kOrderBlock)
roi_nhwc_order_kernel(Levels L, const float* __restrict__ rois, const int32_t* __restrict__ levels_in, int K,
                      int32_t* __restrict__ order) {
  __shared__ unsigned long long keys[kOrderMaxK + 2 * kOrderLanes];
  roi_order_role(L, rois, levels_in, K, order, keys, static_cast<int>(blockIdx.x));
}

template <int V, bool kOutNhwc, int kNfBlock>
__global__ void __launch_bounds__(kNfBlock)
roi_align_fwd_nhwc_kernel(Levels L, const float* __restrict__ rois, const int32_t* __restrict__ levels_in,
                          int32_t* __restrict__ levels_out, const int32_t* __restrict__ order, float* __restrict__ out,
                          int C, int K, int PH, int PW, int sr, int groups, int gbins, int cchunks) {
  DETOPS_DYNAMIC_LDS(float, tile);       // kOutNhwc == false: the workgroup's [channels][bins of the group] output block
  __shared__ Tap tabY[kNfTab];
  __shared__ Tap tabX[kNfTab];
  constexpr int kNfWaves = kNfBlock / kWave;
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  const int bins = PH * PW;
  const int per_roi = groups * cchunks;
  const int64_t lid = xcd_contiguous(static_cast<int64_t>(blockIdx.x), static_cast<int64_t>(gridDim.x));
  const int kk = static_cast<int>(lid / per_roi);
  const int rem = static_cast<int>(lid - static_cast<int64_t>(kk) * per_roi);
  const int grp = rem / cchunks, chunk = rem - grp * cchunks;
  const int k = order ? order[kk] : kk;
  const float* roi = rois + static_cast<size_t>(k) * 5;
  int lvl = 0;
  if (L.num > 1) lvl = levels_in ? levels_in[k] : fpn_level(roi, L);
  if (levels_out && rem == 0 && tid == 0) levels_out[k] = lvl;
  const float* in = L.lv[0].in; int H = L.lv[0].H, W = L.lv[0].W; float scale = L.lv[0].scale;
#pragma unroll
  for (int i = 1; i < DETOPS_MAX_LEVELS; ++i)
    if (i == lvl) { in = L.lv[i].in; H = L.lv[i].H; W = L.lv[i].W; scale = L.lv[i].scale; }

  const RoiGeom g = roi_geometry(roi, scale, PH, PW, sr);
  const int ny = PH * g.gh, nx = PW * g.gw;
  const bool use_tab = (ny <= kNfTab) && (nx <= kNfTab);
  if (use_tab) {
    for (int t = tid; t < ny + nx; t += kNfBlock) {
      if (t < ny) tabY[t] = axis_entry(g.start_h, g.bin_h, t / g.gh, t % g.gh, g.gh, H, 1);
      else { const int u = t - ny; tabX[u] = axis_entry(g.start_w, g.bin_w, u / g.gw, u % g.gw, g.gw, W, 1); }
    }
    __syncthreads();
  }

  const int c0 = chunk * (kWave * V);
  const int cl = c0 + lane * V;                   // this lane's first channel
  const bool active = cl < C;                     // C % V == 0 (host)
  const int cc = min(C - c0, kWave * V);          // channels of this chunk
  const int g0 = grp * gbins;
  const int nb = min(gbins, bins - g0);           // bins of this group
  const float* base = in + static_cast<size_t>(g.b) * H * W * C + (active ? cl : 0);

  for (int bi = wave; bi < nb; bi += kNfWaves) {
#pragma clang fp contract(off)
    const int bin = g0 + bi;
    const int ph = bin / PW, pw = bin - ph * PW;
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.f;
    for (int iy = 0; iy < g.gh; ++iy) {
      const Tap ty = use_tab ? tabY[ph * g.gh + iy] : axis_entry(g.start_h, g.bin_h, ph, iy, g.gh, H, 1);
      const size_t r0 = static_cast<size_t>(ty.lo) * W, r1 = static_cast<size_t>(ty.hi) * W;
      for (int ix = 0; ix < g.gw; ++ix) {
        const Tap tx = use_tab ? tabX[pw * g.gw + ix] : axis_entry(g.start_w, g.bin_w, pw, ix, g.gw, W, 1);
        const float w1 = ty.h * tx.h, w2 = ty.h * tx.l, w3 = ty.l * tx.h, w4 = ty.l * tx.l;
        float v1[V], v2[V], v3[V], v4[V];
        nv_load<V>(v1, base + (r0 + tx.lo) * C);
        nv_load<V>(v2, base + (r0 + tx.hi) * C);
        nv_load<V>(v3, base + (r1 + tx.lo) * C);
        nv_load<V>(v4, base + (r1 + tx.hi) * C);
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] += w1 * v1[j] + w2 * v2[j] + w3 * v3[j] + w4 * v4[j];
      }
    }
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = acc[j] / g.count;
    if (kOutNhwc) {
      if (active) {
        float* o = out + (static_cast<size_t>(k) * bins + bin) * C + cl;
        if constexpr (V == 4) *reinterpret_cast<float4*>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        else o[0] = acc[0];
      }
    } else if (active) {
#pragma unroll
      for (int j = 0; j < V; ++j) tile[(lane * V + j) * nb + bi] = acc[j];
    }
  }
  if (!kOutNhwc) {
    __syncthreads();
    const int total = cc * nb;
    if (per_roi == 1 && (total & 3) == 0) {
      // the ROI's whole [C][bins] block: one contiguous, 16-byte aligned piece of the output
      float4* dst = reinterpret_cast<float4*>(out + static_cast<size_t>(k) * C * bins);
      const float4* src = reinterpret_cast<const float4*>(tile);
      for (int o = tid; o < total / 4; o += kNfBlock) dst[o] = src[o];
    } else {
      for (int o = tid; o < total; o += kNfBlock) {
        const int c = o / nb, j = o - c * nb;
        out[(static_cast<size_t>(k) * C + c0 + c) * bins + g0 + j] = tile[o];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------------------------
constexpr int kNbTH = 4, kNbTW = 8; // tile: 4 rows x 8 columns of pixels (a lane keeps its 32 sums in registers)
constexpr int kNbCh = kWave;       // channels per workgroup (a lane each)
constexpr int kNbBlock = 256;      // 4 waves, each walks whole hits
constexpr int kNbWaves = kNbBlock / kWave;
constexpr int kNbMaxC = 4;         // fast walk: bins per pixel and axis held in registers (longer ranges: generic walk)
constexpr int kNbStageBins = 64;   // NCHW gradient: the hit's [64 channels][bins] block is staged in LDS when bins <= this
constexpr int kNbPrepTiles = 4;    // tiles per list-role workgroup of the pre-pass (a wave each)
constexpr int kNbPrepRois = 1024;  // ROI extents parked in LDS per pass of the list role

struct NbPlan {
  int first_tile[DETOPS_MAX_LEVELS], n_tiles[DETOPS_MAX_LEVELS], tiles_x[DETOPS_MAX_LEVELS], tiles_y[DETOPS_MAX_LEVELS];
  int num_tiles, cap;       // cap = hit-list capacity per tile (= K)
  int cchunks;              // channel chunks of kNbCh
  int PPH, PPW, Hmax, Wmax; // adjoint-row strides (floats, multiple of 4) and the per-ROI table slot = Hmax * PPH + Wmax * PPW
  int rec_blocks;           // pre-pass: blocks [0, rec_blocks) build records + rows (a wave per ROI), the rest the hit lists
  int accumulate;
};

struct NbWs {
  int4* recs;       // [K][2]: {level, image, fy0, ny}, {fx0, nx, longest y range, longest x range}
  int* counts;      // [num_tiles] hits per tile
  int4* lists;      // [num_tiles][cap] {ROI index, fy0 | ny << 16, fx0 | nx << 16, 0}, ascending ROI index
  float* tabs;      // [K][Hmax * PPH + Wmax * PPW] compact adjoint rows (build_adjoint_rows), then 16 bytes of zeros
};

struct NbLayout { size_t off_counts, off_lists, off_tabs, total; };

bool nb_plan(const Levels& L, int N, int C, int K, int PH, int PW, NbPlan& P, NbLayout& lay) {
  if (K <= 0 || K > (1 << 20) || N <= 0 || C <= 0 || PH > 255 || PW > 255) return false;
  P = NbPlan{};
  P.PPH = (PH + 4) & ~3;
  P.PPW = (PW + 4) & ~3;
  P.cap = K;
  P.cchunks = static_cast<int>(ceil_div64(C, kNbCh));
  int64_t tiles = 0;
  for (int i = L.num - 1; i >= 0; --i) {   // coarsest level first: its tiles see the most ROIs, their chains start first
    if (L.lv[i].W > 32767 || L.lv[i].H > 32767) return false;
    P.Hmax = max(P.Hmax, L.lv[i].H);
    P.Wmax = max(P.Wmax, L.lv[i].W);
    P.tiles_x[i] = static_cast<int>(ceil_div64(L.lv[i].W, kNbTW));
    P.tiles_y[i] = static_cast<int>(ceil_div64(L.lv[i].H, kNbTH));
    const int64_t n = static_cast<int64_t>(N) * P.tiles_x[i] * P.tiles_y[i];
    P.first_tile[i] = static_cast<int>(tiles);
    P.n_tiles[i] = static_cast<int>(n);
    tiles += n;
  }
  if (tiles <= 0 || tiles * P.cchunks > 0x3fffffff) return false;
  P.num_tiles = static_cast<int>(tiles);
  P.rec_blocks = static_cast<int>(ceil_div64(K, kBlock / kWave));
  auto up = [](size_t v) { return (v + 255) & ~static_cast<size_t>(255); };
  size_t o = up(sizeof(int4) * 2 * static_cast<size_t>(K));
  lay.off_counts = o; o = up(o + sizeof(int) * static_cast<size_t>(P.num_tiles));
  lay.off_lists = o;  o = up(o + sizeof(int4) * static_cast<size_t>(P.num_tiles) * P.cap);
  lay.off_tabs = o;   o = up(o + sizeof(float) * static_cast<size_t>(K) *
                                 (static_cast<size_t>(P.Hmax) * P.PPH + static_cast<size_t>(P.Wmax) * P.PPW) + 16);
  lay.total = o;
  return true;
}

struct NbTile { int lvl, b, y0, x0; };
__device__ __forceinline__ NbTile nb_tile(const Levels& L, const NbPlan& P, int tile) {
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
  NbTile t;
  t.lvl = lvl; t.b = rem / nty; t.y0 = (rem % nty) * kNbTH; t.x0 = tix * kNbTW;
  return t;
}

// Pre-pass, one launch, two roles (neither reads what the other writes).
//   record role — one wave per ROI: {level, image, footprint} record + the ROI's compact adjoint rows (build_adjoint_rows:
//       per footprint pixel and axis {first bin | other axis' longest range << 8 | count << 16, w0, w1, ...}).
//   list role — one wave per gradient-map tile: ordered (ascending ROI index) ballot compaction of the ROIs whose
//       footprint reaches the tile -> hit list + count.
__global__ void __launch_bounds__(kBlock)
roi_bwd_nhwc_prep_kernel(Levels L, NbPlan P, NbWs ws, const float* __restrict__ rois, const int32_t* __restrict__ levels_in,
                         int K, int PH, int PW, int sr) {
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  const size_t slot = static_cast<size_t>(P.Hmax) * P.PPH + static_cast<size_t>(P.Wmax) * P.PPW;
  if (static_cast<int>(blockIdx.x) < P.rec_blocks) {
    const int r = static_cast<int>(blockIdx.x) * (kBlock / kWave) + wave;
    if (blockIdx.x == 0 && tid < 4) ws.tabs[static_cast<size_t>(K) * slot + tid] = 0.f;   // the zero row (pixels outside a footprint)
    if (r >= K) return;
    const float* roi = rois + static_cast<size_t>(r) * 5;
    int lvl = 0;
    if (L.num > 1) lvl = levels_in ? levels_in[r] : fpn_level(roi, L);
    int H = L.lv[0].H, W = L.lv[0].W; float scale = L.lv[0].scale;
#pragma unroll
    for (int i = 1; i < DETOPS_MAX_LEVELS; ++i)
      if (i == lvl) { H = L.lv[i].H; W = L.lv[i].W; scale = L.lv[i].scale; }
    float* rows = ws.tabs + static_cast<size_t>(r) * slot;
    build_adjoint_rows(roi, scale, H, W, PH, PW, sr, rows, static_cast<size_t>(P.Hmax) * P.PPH, P.PPH, P.PPW, lane);
    if (lane == 0) {
      const RoiExtent e = roi_extent(roi, scale, H, W, PH, PW, sr);
      const bool ok = e.ny > 0 && e.nx > 0;
      ws.recs[2 * static_cast<size_t>(r)] = make_int4(lvl, e.b, e.fy0, ok ? e.ny : 0);
      ws.recs[2 * static_cast<size_t>(r) + 1] = make_int4(e.fx0, ok ? e.nx : 0, 0, 0);
    }
    return;
  }
  // ---- list role
  __shared__ int4 ext[kNbPrepRois];     // {level | image << 8, fy0 | ny << 16, fx0 | nx << 16, 0}
  const int tile = (static_cast<int>(blockIdx.x) - P.rec_blocks) * kNbPrepTiles + wave;
  const bool live = tile < P.num_tiles;
  NbTile t{0, 0, 0, 0};
  if (live) t = nb_tile(L, P, tile);
  int count = 0;
  int4* list = ws.lists + static_cast<size_t>(live ? tile : 0) * P.cap;
  for (int r0 = 0; r0 < K; r0 += kNbPrepRois) {
    const int n = min(kNbPrepRois, K - r0);
    __syncthreads();
    for (int i = tid; i < n; i += kBlock) {
      const float* roi = rois + static_cast<size_t>(r0 + i) * 5;
      int lvl = 0;
      if (L.num > 1) lvl = levels_in ? levels_in[r0 + i] : fpn_level(roi, L);
      int H = L.lv[0].H, W = L.lv[0].W; float scale = L.lv[0].scale;
#pragma unroll
      for (int q = 1; q < DETOPS_MAX_LEVELS; ++q)
        if (q == lvl) { H = L.lv[q].H; W = L.lv[q].W; scale = L.lv[q].scale; }
      const RoiExtent e = roi_extent(roi, scale, H, W, PH, PW, sr);
      const bool ok = e.ny > 0 && e.nx > 0;
      ext[i] = make_int4(lvl | (e.b << 8), ok ? (e.fy0 | (e.ny << 16)) : 0, ok ? (e.fx0 | (e.nx << 16)) : 0, 0);
    }
    __syncthreads();
    if (live) {
      for (int i0 = 0; i0 < n; i0 += kWave) {
        const int i = i0 + lane;
        bool hit = false;
        int4 e = make_int4(0, 0, 0, 0);
        if (i < n) {
          e = ext[i];
          const int fy0 = e.y & 0xffff, eny = e.y >> 16, fx0 = e.z & 0xffff, enx = e.z >> 16;
          hit = (e.x == (t.lvl | (t.b << 8))) && eny > 0 && enx > 0 && fy0 < t.y0 + kNbTH && fy0 + eny > t.y0 &&
                fx0 < t.x0 + kNbTW && fx0 + enx > t.x0;
        }
        const unsigned long long m = __ballot(hit);
        if (hit) list[count + __popcll(m & ((1ull << lane) - 1ull))] = make_int4(r0 + i, e.y, e.z, 0);
        count += __popcll(m);
      }
    }
  }
  if (live && lane == 0) ws.counts[tile] = count;
}

// One hit's walk by one wave: the whole tile (kNbTH x kNbTW pixels), a lane per channel.
//   rows  LDS (this wave's), [kNbTH + kNbTW][PP]: compact adjoint rows of the tile's pixel rows, then of its pixel columns
//   G     the hit's pooled gradient as seen by this lane: G(bin) -> value (any bin index >= 0 is readable)
// The x pass of bin row ph (u[x] = sum_pw AX[x, pw] * g[ph, pw]) is computed once and used by every pixel row ph reaches.
// kFast (every column's bin range has at most kNbMaxC entries): BRANCH-FREE — the x weights live in registers (compact rows
// are zero-padded, so the entries beyond a column's range multiply neighbouring bins by 0), a pixel row outside ph's reach
// gets the weight 0.  For finite gradients this is exact; a non-finite pooled gradient (a GradScaler overflow step, which
// is skipped anyway) turns into NaN in the pixels around it instead of staying in the bins' own footprint.
// Otherwise: scalar loops over the exact ranges.
template <bool kFast, typename GFn>
__device__ __forceinline__ void nb_walk(const float* rows, int PP, int PW, float (&acc)[kNbTH][kNbTW], GFn G) {
  int lo_y[kNbTH], cnt_y[kNbTH];
  int ph_lo = 0x7fffffff, ph_hi = -1;
#pragma unroll
  for (int j = 0; j < kNbTH; ++j) {
    const unsigned h = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(__float_as_int(rows[j * PP])));
    lo_y[j] = static_cast<int>(h & 0xffu);
    cnt_y[j] = static_cast<int>(h >> 16);
    if (cnt_y[j] > 0) { ph_lo = min(ph_lo, lo_y[j]); ph_hi = max(ph_hi, lo_y[j] + cnt_y[j] - 1); }
  }
  if (ph_hi < 0) return;
  int lo_x[kNbTW], cnt_x[kNbTW];
#pragma unroll
  for (int x = 0; x < kNbTW; ++x) {
    const unsigned h = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(__float_as_int(rows[(kNbTH + x) * PP])));
    cnt_x[x] = static_cast<int>(h >> 16);
    lo_x[x] = cnt_x[x] > 0 ? static_cast<int>(h & 0xffu) : 0;
  }
  if (kFast) {
    float wx[kNbTW][kNbMaxC];
#pragma unroll
    for (int x = 0; x < kNbTW; ++x)
#pragma unroll
      for (int t = 0; t < kNbMaxC; ++t) wx[x][t] = rows[(kNbTH + x) * PP + 1 + t];
    for (int ph = ph_lo; ph <= ph_hi; ++ph) {
      const int rb = ph * PW;
      float u[kNbTW];
#pragma unroll
      for (int x = 0; x < kNbTW; ++x) {
        const int b0 = rb + lo_x[x];
        u[x] = wx[x][0] * G(b0);
#pragma unroll
        for (int t = 1; t < kNbMaxC; ++t) u[x] += wx[x][t] * G(b0 + t);
      }
#pragma unroll
      for (int j = 0; j < kNbTH; ++j) {
        const int d = ph - lo_y[j];
        const bool on = d >= 0 && d < cnt_y[j];
        const float w = rows[j * PP + 1 + (on ? d : 0)];
        const float wy = on ? w : 0.f;
#pragma unroll
        for (int x = 0; x < kNbTW; ++x) acc[j][x] += wy * u[x];
      }
    }
  } else {
    for (int ph = ph_lo; ph <= ph_hi; ++ph) {
      const int rb = ph * PW;
      float u[kNbTW];
#pragma unroll
      for (int x = 0; x < kNbTW; ++x) {
        u[x] = 0.f;
        for (int t = 0; t < cnt_x[x]; ++t) u[x] += rows[(kNbTH + x) * PP + 1 + t] * G(rb + lo_x[x] + t);
      }
#pragma unroll
      for (int j = 0; j < kNbTH; ++j) {
        if (cnt_y[j] > 0 && ph >= lo_y[j] && ph < lo_y[j] + cnt_y[j]) {
          const float wy = rows[j * PP + 1 + (ph - lo_y[j])];
#pragma unroll
          for (int x = 0; x < kNbTW; ++x)
            if (cnt_x[x] > 0) acc[j][x] += wy * u[x];
        }
      }
    }
  }
}

// kGNhwc: grad_out is [K, PH, PW, C] (read as coalesced channel vectors); else [K, C, PH, PW].
// kStage (NCHW gradient only): the hit's [channels of the chunk][bins] block goes through this wave's LDS buffer (LDS-DMA).
// PERSISTENT: the grid is what the chip holds at once; a workgroup takes the (tile, channel chunk) units b, b + grid, ...
// (tiles are numbered coarsest level first, so every workgroup starts on a crowded tile and ends on empty ones, which
// cost a count and one row of stores per wave).
template <bool kGNhwc, bool kStage>
__global__ void __launch_bounds__(kNbBlock, 3)
roi_align_bwd_nhwc_kernel(Levels L, NbPlan P, NbWs ws, const float* __restrict__ gout, int C, int PH, int PW, int units) {
  DETOPS_DYNAMIC_LDS(float, smem);
  // smem: per wave { rows [kNbTH + kNbTW][PP] | (kStage) gblk [kNbCh * bins, padded to 1 KiB pieces] }; reused for the combine
  constexpr int kR = kNbTH + kNbTW;
  static_assert(kNbTH == kNbWaves, "the epilogue hands pixel row w of the tile to wave w");
  const int PP = max(P.PPH, P.PPW);
  const int bins = PH * PW;
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  // the wave index is wave-uniform, but the compiler only knows that when it comes out of a scalar register: everything
  // derived from it (hit index, list entry, row addresses) then lives in SGPRs (162 -> 134 VGPRs)
  const int wave = static_cast<int>(__builtin_amdgcn_readfirstlane(static_cast<unsigned>(tid / kWave)));
  const size_t slot = static_cast<size_t>(P.Hmax) * P.PPH + static_cast<size_t>(P.Wmax) * P.PPW;
  const int rows_f = kR * PP;
  const int gblk_f = kStage ? ((kNbCh * bins + 255) & ~255) : 0;     // whole 1 KiB DMA pieces
  const int wave_f = rows_f + gblk_f;
  float* rw = smem + wave * wave_f;                                   // this wave's rows
  float* gb = rw + rows_f;                                            // this wave's gradient block
  bool lds_dirty = false;                                             // the previous unit's combine still owns the LDS

  for (int unit = static_cast<int>(blockIdx.x); unit < units; unit += static_cast<int>(gridDim.x)) {
    const int tile = unit / P.cchunks, chunk = unit - tile * P.cchunks;
    const NbTile t = nb_tile(L, P, tile);
    float* gin = L.lv[0].gin; int H = L.lv[0].H, W = L.lv[0].W;
#pragma unroll
    for (int i = 1; i < DETOPS_MAX_LEVELS; ++i)
      if (i == t.lvl) { gin = L.lv[i].gin; H = L.lv[i].H; W = L.lv[i].W; }
    const int c0 = chunk * kNbCh;
    const int cc = min(kNbCh, C - c0);
    const bool chan = lane < cc;
    const int count = ws.counts[tile];
    auto put = [&](int y, int x, float v) {
      const int xx = t.x0 + x;
      if (y < H && xx < W && chan) {
        float* o = gin + ((static_cast<size_t>(t.b) * H + y) * W + xx) * C + c0 + lane;
        *o = P.accumulate ? (*o + v) : v;
      }
    };
    if (count == 0) {
      // no ROI reaches the tile: zeros, one pixel row per wave (every element is written exactly once)
#pragma unroll
      for (int x = 0; x < kNbTW; ++x) put(t.y0 + wave, x, 0.f);
      continue;
    }
    const int4* list = ws.lists + static_cast<size_t>(tile) * P.cap;
    if (lds_dirty) { __syncthreads(); lds_dirty = false; }

    float acc[kNbTH][kNbTW];
#pragma unroll
    for (int j = 0; j < kNbTH; ++j)
#pragma unroll
      for (int x = 0; x < kNbTW; ++x) acc[j][x] = 0.f;

    for (int i = wave; i < count; i += kNbWaves) {
      const int4 e = list[i];
      const int r = e.x;
      const int fy0 = e.y & 0xffff, eny = e.y >> 16, fx0 = e.z & 0xffff, enx = e.z >> 16;
      DETOPS_WAVE_SYNC();           // the previous hit's LDS reads are done before its buffers are overwritten
      if (kStage) {
        // [r][c0 ..][bins]: cc * bins contiguous floats (host: (C * bins) % 4 == 0), copied lane-linear by LDS-DMA
        const float* src = gout + (static_cast<size_t>(r) * C + c0) * bins;
        const int total4 = cc * bins / 4;
        for (int o = 0; o < total4; o += kWave)
          if (o + lane < total4) glds16(src + 4 * (o + lane), gb + 4 * o);
      }
      {
        const int q4 = PP / 4;                                           // float4 pieces per row
        const float* rbase = ws.tabs + static_cast<size_t>(r) * slot;
        for (int p = lane; p < kR * q4; p += kWave) {
          const int rr = p / q4, piece = p - rr * q4;
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (rr < kNbTH) {
            const int yi = t.y0 + rr - fy0;
            if (yi >= 0 && yi < eny && piece * 4 < P.PPH) v = *reinterpret_cast<const float4*>(rbase + static_cast<size_t>(yi) * P.PPH + piece * 4);
          } else {
            const int xi = t.x0 + (rr - kNbTH) - fx0;
            if (xi >= 0 && xi < enx && piece * 4 < P.PPW)
              v = *reinterpret_cast<const float4*>(rbase + static_cast<size_t>(P.Hmax) * P.PPH + static_cast<size_t>(xi) * P.PPW + piece * 4);
          }
          *reinterpret_cast<float4*>(rw + rr * PP + piece * 4) = v;
        }
      }
      DETOPS_WAVE_SYNC();           // rows written by some lanes are read by all lanes (the DMA'd block is waited for by the compiler)
      // longest bin range of any of the tile's columns for this ROI: selects the walk
      int cmax = 0;
#pragma unroll
      for (int x = 0; x < kNbTW; ++x)
        cmax = max(cmax, static_cast<int>(__builtin_amdgcn_readfirstlane(static_cast<unsigned>(__float_as_int(rw[(kNbTH + x) * PP]))) >> 16));
      if (kStage) {
        const float* gl = gb + lane * bins;          // reads up to kNbMaxC - 1 floats past a row's end stay inside the padded block
        auto G = [&](int bin) { return gl[bin]; };
        if (cmax <= kNbMaxC) nb_walk<true>(rw, PP, PW, acc, G);
        else nb_walk<false>(rw, PP, PW, acc, G);
      } else if (kGNhwc) {
        const float* gp = gout + static_cast<size_t>(r) * bins * C + c0 + (chan ? lane : 0);
        auto G = [&](int bin) { return gp[static_cast<size_t>(min(bin, bins - 1)) * C]; };
        if (cmax <= kNbMaxC) nb_walk<true>(rw, PP, PW, acc, G);
        else nb_walk<false>(rw, PP, PW, acc, G);
      } else {
        const float* gp = gout + (static_cast<size_t>(r) * C + c0 + (chan ? lane : 0)) * bins;
        auto G = [&](int bin) { return gp[min(bin, bins - 1)]; };
        if (cmax <= kNbMaxC) nb_walk<true>(rw, PP, PW, acc, G);
        else nb_walk<false>(rw, PP, PW, acc, G);
      }
    }

    // ---- the result = the waves' partial tiles added in wave order (deterministic); every element stored exactly once
    const int nparts = min(count, kNbWaves);            // waves that walked at least one hit (workgroup-uniform)
    if (nparts == 1) {
      if (wave == 0) {                                   // wave 0's registers are the result
#pragma unroll
        for (int j = 0; j < kNbTH; ++j)
#pragma unroll
          for (int x = 0; x < kNbTW; ++x) put(t.y0 + j, x, acc[j][x]);
      }
    } else {
      // part[wave][row][x][channel] through LDS (32 KiB, over the staging buffers); wave w then adds and stores pixel row w
      float* part = smem;
      __syncthreads();          // every wave is done with its staging buffers
      if (wave < nparts) {
#pragma unroll
        for (int j = 0; j < kNbTH; ++j)
#pragma unroll
          for (int x = 0; x < kNbTW; ++x) part[((wave * kNbTH + j) * kNbTW + x) * kNbCh + lane] = acc[j][x];
      }
      __syncthreads();
#pragma unroll
      for (int x = 0; x < kNbTW; ++x) {
        float v = part[((0 * kNbTH + wave) * kNbTW + x) * kNbCh + lane];
        for (int w2 = 1; w2 < nparts; ++w2) v += part[((w2 * kNbTH + wave) * kNbTW + x) * kNbCh + lane];
        put(t.y0 + wave, x, v);
      }
      lds_dirty = true;
    }
  }
}

int nhwc_levels(Levels& L, const float* const* in_host, float* const* gin_host, const int* H_host, const int* W_host,
                const float* scale_host, int num_levels) {
  if (num_levels < 1 || num_levels > DETOPS_MAX_LEVELS || !H_host || !W_host || !scale_host) return DETOPS_EINVAL;
  L = Levels{};
  L.num = num_levels;
  for (int i = 0; i < num_levels; ++i) {
    if (H_host[i] <= 0 || W_host[i] <= 0) return DETOPS_EINVAL;
    if ((in_host && !in_host[i]) || (gin_host && !gin_host[i])) return DETOPS_EINVAL;
    L.lv[i] = Level{in_host ? in_host[i] : nullptr, gin_host ? gin_host[i] : nullptr, H_host[i], W_host[i], scale_host[i]};
  }
  return 0;
}

}  // namespace

DETOPS_API size_t detops_roi_align_fpn_forward_nhwc_workspace_bytes(int K) {
  return (K >= kOrderMinK && K <= kOrderMaxK) ? sizeof(int32_t) * static_cast<size_t>(K) : 0;
}

DETOPS_API int detops_roi_align_fpn_forward_nhwc_f32(
    const float* const* inputs_host, const int* H_host, const int* W_host, const float* scale_host, int num_levels,
    const float* rois, float* output, int output_nhwc, int32_t* levels_out, int N, int C, int K, int PH, int PW,
    int sampling_ratio, int k_min, int k_max, float canonical_scale, float canonical_level, float eps, void* workspace,
    size_t workspace_bytes, detops_stream_t stream) {
  if (bad_dims(N, C, K, PH, PW) || !inputs_host) return DETOPS_EINVAL;
  Levels L;
  const int rc = nhwc_levels(L, inputs_host, nullptr, H_host, W_host, scale_host, num_levels);
  if (rc) return rc;
  L.k_min = k_min; L.k_max = k_max; L.s0 = canonical_scale; L.lvl0 = canonical_level; L.eps = eps;
  if (K == 0 || C == 0) return 0;
  if (N == 0 || !rois || !output) return DETOPS_EINVAL;
  hipStream_t st = as_stream(stream);
  const int bins = PH * PW;
  int32_t* order = nullptr;
  if (workspace && K >= kOrderMinK && K <= kOrderMaxK && workspace_bytes >= sizeof(int32_t) * static_cast<size_t>(K) &&
      detops_tuning().roi_fwd_order != 1) {
    order = static_cast<int32_t*>(workspace);
    const int blocks = static_cast<int>(ceil_div64(static_cast<int64_t>(K) * kOrderLanes, kOrderBlock));
    hipLaunchKernelGGL(roi_nhwc_order_kernel, dim3(blocks), dim3(kOrderBlock), 0, st, L, rois, static_cast<const int32_t*>(nullptr), K,
                       order);
  }
  const int V = (C % 4 == 0) ? 4 : 1;
  const int cchunks = static_cast<int>(ceil_div64(C, kWave * V));
  const int gbins = std::min(bins, kNfGroupBins);
  const int groups = static_cast<int>(ceil_div64(bins, gbins));
  const int64_t grid = static_cast<int64_t>(K) * groups * cchunks;
  if (grid > 0x7fffffff) return DETOPS_EUNSUPPORTED;
  const size_t lds = output_nhwc ? 0 : sizeof(float) * static_cast<size_t>(kWave) * V * gbins;
#define NF_LAUNCH(VV, NHWC, BLK)                                                                                          \
  hipLaunchKernelGGL((roi_align_fwd_nhwc_kernel<VV, NHWC, BLK>), dim3(static_cast<unsigned>(grid)), dim3(BLK), lds, st, L, rois, \
                     static_cast<const int32_t*>(nullptr), levels_out, static_cast<const int32_t*>(order), output, C, K, PH, PW,  \
                     sampling_ratio, groups, gbins, cchunks)
  const int blk = detops_tuning().roi_fwd_ct;          // A/B switch: threads per workgroup of the [K, C, PH, PW]-output form (0 = default)
  if (V == 4) {
    if (output_nhwc) NF_LAUNCH(4, true, kNfBlockNhwc);
    else if (blk == 256) NF_LAUNCH(4, false, 256);
    else if (blk == 512) NF_LAUNCH(4, false, 512);
    else NF_LAUNCH(4, false, kNfBlockNchw);
  } else {
    if (output_nhwc) NF_LAUNCH(1, true, kNfBlockNhwc); else NF_LAUNCH(1, false, kNfBlockNchw);
  }
#undef NF_LAUNCH
  return launch_status();
}

DETOPS_API size_t detops_roi_align_fpn_backward_nhwc_workspace_bytes(const int* H_host, const int* W_host, int num_levels,
                                                                     int N, int C, int K, int PH, int PW) {
  if (!H_host || !W_host || num_levels < 1 || num_levels > DETOPS_MAX_LEVELS || bad_dims(N, C, K, PH, PW) || K == 0 ||
      C == 0 || N == 0)
    return 0;
  Levels L{};
  L.num = num_levels;
  for (int i = 0; i < num_levels; ++i) {
    if (H_host[i] <= 0 || W_host[i] <= 0) return 0;
    L.lv[i] = Level{nullptr, nullptr, H_host[i], W_host[i], 1.f};
  }
  NbPlan P; NbLayout lay;
  return nb_plan(L, N, C, K, PH, PW, P, lay) ? lay.total : 0;
}

DETOPS_API int detops_roi_align_fpn_backward_nhwc_f32(
    const float* grad_out, int grad_out_nhwc, const float* rois, const int32_t* levels, float* const* grad_inputs_host,
    const int* H_host, const int* W_host, const float* scale_host, int num_levels, int N, int C, int K, int PH, int PW,
    int sampling_ratio, int zero_grad_in, void* workspace, size_t workspace_bytes, detops_stream_t stream) {
  if (bad_dims(N, C, K, PH, PW) || !grad_inputs_host) return DETOPS_EINVAL;
  Levels L;
  const int rc = nhwc_levels(L, nullptr, grad_inputs_host, H_host, W_host, scale_host, num_levels);
  if (rc) return rc;
  if (C == 0 || N == 0) return 0;
  hipStream_t st = as_stream(stream);
  if (K == 0) {
    if (!zero_grad_in) return 0;
    for (int i = 0; i < num_levels; ++i)
      DETOPS_HIP_TRY(hipMemsetAsync(L.lv[i].gin, 0, sizeof(float) * static_cast<size_t>(N) * C * L.lv[i].H * L.lv[i].W, st));
    return 0;
  }
  if (!grad_out || !rois || (num_levels > 1 && !levels) || !workspace) return DETOPS_EINVAL;
  NbPlan P; NbLayout lay;
  if (!nb_plan(L, N, C, K, PH, PW, P, lay)) return DETOPS_EUNSUPPORTED;
  if (workspace_bytes < lay.total) return DETOPS_EINVAL;
  P.accumulate = zero_grad_in ? 0 : 1;
  unsigned char* base = static_cast<unsigned char*>(workspace);
  NbWs ws{reinterpret_cast<int4*>(base), reinterpret_cast<int*>(base + lay.off_counts), reinterpret_cast<int4*>(base + lay.off_lists),
          reinterpret_cast<float*>(base + lay.off_tabs)};
  const int list_blocks = static_cast<int>(ceil_div64(P.num_tiles, kNbPrepTiles));
  hipLaunchKernelGGL(roi_bwd_nhwc_prep_kernel, dim3(P.rec_blocks + list_blocks), dim3(kBlock), 0, st, L, P, ws, rois, levels, K, PH, PW,
                     sampling_ratio);
  int e = launch_status();
  if (e) return e;
  const int bins = PH * PW;
  const int PP = std::max(P.PPH, P.PPW);
  // staged through LDS as [channel][bin]: conflict-free reads need an odd row stride, the 16-byte DMA pieces (C * bins) % 4 == 0
  const bool stage = !grad_out_nhwc && bins <= kNbStageBins && (bins & 1) && (static_cast<int64_t>(C) * bins) % 4 == 0;
  const size_t wave_f = (kNbTH + kNbTW) * static_cast<size_t>(PP) + (stage ? static_cast<size_t>((kNbCh * bins + 255) & ~255) : 0);
  const size_t lds = sizeof(float) * std::max(kNbWaves * wave_f, static_cast<size_t>(kNbWaves) * kNbTH * kNbTW * kNbCh);
  const int units = static_cast<int>(static_cast<int64_t>(P.num_tiles) * P.cchunks);
  // persistent grid: the workgroups the chip holds at once (advisory: any grid size is correct)
  int resident = -1;
  if (grad_out_nhwc) resident = detops_resident_workgroups(roi_align_bwd_nhwc_kernel<true, false>, kNbBlock, lds);
  else if (stage) resident = detops_resident_workgroups(roi_align_bwd_nhwc_kernel<false, true>, kNbBlock, lds);
  else resident = detops_resident_workgroups(roi_align_bwd_nhwc_kernel<false, false>, kNbBlock, lds);
  if (resident <= 0) resident = 2 * kNumCU;
  const dim3 grid(static_cast<unsigned>(std::min(units, resident)));
  if (grad_out_nhwc)
    hipLaunchKernelGGL((roi_align_bwd_nhwc_kernel<true, false>), grid, dim3(kNbBlock), lds, st, L, P, ws, grad_out, C, PH, PW, units);
  else if (stage)
    hipLaunchKernelGGL((roi_align_bwd_nhwc_kernel<false, true>), grid, dim3(kNbBlock), lds, st, L, P, ws, grad_out, C, PH, PW, units);
  else
    hipLaunchKernelGGL((roi_align_bwd_nhwc_kernel<false, false>), grid, dim3(kNbBlock), lds, st, L, P, ws, grad_out, C, PH, PW, units);
  return launch_status();
}
