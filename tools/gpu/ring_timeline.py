"""Timeline of the ring backward's workgroups (roi_bwd_debug bit 64): when each unit started / ended (100 MHz wall clock),
how many hits it walked.  python tools/gpu/ring_timeline.py [set] [tune k=v,...]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "maskrcnn-benchmark_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
import synth
from opbench import tune, _t
from maskrcnn_benchmark import _C as C, _lib
lib, ptr, stream_of = _lib.lib, C.ptr, C.stream_of
name = sys.argv[1] if len(sys.argv) > 1 else "model-random-init"
for kv in filter(None, (sys.argv[2] if len(sys.argv) > 2 else "").split(",")):
    k, v = kv.split("="); tune(k, int(v))
tune("roi_bwd_debug", 64)
rois = synth.roi_sets()[name]["box"]
K = rois.shape[0]
lv = synth.level_map(rois)
tr, tl = _t(rois), _t(lv)
shapes = [(2, 256, h, w) for (h, w) in synth.fpn_shapes()[:4]]
scales = [1.0 / s for s in synth.FPN_STRIDES[:4]]
g = torch.randn(K, 256, 7, 7, device="cuda")
gins = [torch.empty(s, device="cuda") for s in shapes]
ptrs, Hs, Ws, sc = C._host_arrays(gins, scales)
nbytes = int(lib.detops_roi_align_backward_workspace_bytes(Hs, Ws, 4, 2, 256, K, 7, 7))
ws = torch.zeros((nbytes,), dtype=torch.uint8, device="cuda")
for it in range(3):
    rc = lib.detops_roi_align_fpn_backward_ws_f32(ptr(g), ptr(tr), ptr(tl), ptrs, Hs, Ws, sc, 4, 2, 256, K, 7, 7, 2, 1, ptr(ws), nbytes, stream_of(g))
    assert rc == 0
torch.cuda.synchronize()
chunks = 256 // (16 if _lib.tuning_get("roi_bwd_ct") == 16 else 32)
tiles = sum(2 * -(-h // 8) * -(-w // 32) for (_, _, h, w) in shapes)
extra_cap = min(_lib.tuning_get("roi_bwd_extras") or 128, max(8, tiles // 4))
n_units = extra_cap + tiles
tl_bytes = 32 * n_units * chunks
t = ws[nbytes - ((tl_bytes + 255) // 256) * 256:][:tl_bytes].cpu().numpy().view(np.int64).reshape(n_units, chunks, 4)
ran = t[..., 1] > 0
t0 = t[..., 0][ran].min()
start = (t[..., 0] - t0) / 100.0   # us
end = (t[..., 1] - t0) / 100.0
hits = t[..., 2]
print("units that ran:", int(ran.sum()), "of", n_units * chunks, "| kernel span %.1f us" % end[ran].max())
heavy = ran & (hits >= 16)
print("heavy units (>= 16 hits):", int(heavy.sum()), "hits total", int(hits[ran].sum()), "in heavy", int(hits[heavy].sum()))
for lo, hi in ((0, 0), (1, 15), (16, 31), (32, 47), (48, 63), (64, 999)):
    m = ran & (hits >= lo) & (hits <= hi)
    if m.any():
        d = end[m] - start[m]
        print("  hits %3d-%3d: %5d units  start %6.1f..%6.1f us  duration mean %6.2f max %6.2f  end max %6.1f  us/hit %.2f" % (
            lo, hi, m.sum(), start[m].min(), start[m].max(), d.mean(), d.max(), end[m].max(), d.sum() / max(hits[m].sum(), 1)))
# occupancy over time
edges = np.linspace(0, end[ran].max(), 21)
for a, b in zip(edges[:-1], edges[1:]):
    live = ran & (start < b) & (end > a)
    print("  t %5.1f-%5.1f us: %5d units live (%4d heavy)" % (a, b, live.sum(), (live & heavy).sum()))
