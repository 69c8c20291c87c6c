"""Segmented-NMS timing probes on the model's RPN segments in score order (what the detector passes):
work scaling (S = 1 .. 10 segments, 20 segments) and the IoU-arithmetic ablation (nms_debug=1; wrong results, timing only)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "maskrcnn-benchmark_amd")):
    sys.path.insert(0, p)
import synth
from maskrcnn_benchmark import _C, _lib

def dev_time_us(fn, iters=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters

base = synth.rpn_nms_segments()
def case(segs):
    so = [np.argsort(-s, kind="stable") for _, s in segs]
    boxes = torch.from_numpy(np.concatenate([b[o] for (b, _), o in zip(segs, so)])).cuda()
    scores = torch.from_numpy(np.concatenate([s[o] for (_, s), o in zip(segs, so)])).cuda()
    offs = torch.from_numpy(np.cumsum([0] + [len(s) for _, s in segs]).astype(np.int32)).cuda()
    return boxes, scores, offs
big = [x for x in base if len(x[1]) == 2000]
for name, segs in [("S=1 x2000", big[:1]), ("S=2 x2000", big[:2]), ("S=4 x2000", big[:4]), ("S=8 x2000", big[:8]), ("model: 8x2000+2x819", base),
                   ("S=16 x2000", big + big), ("S=1 x819", [x for x in base if len(x[1]) != 2000][:1])]:
    b, s, o = case(segs)
    row = []
    for dbg in (0, 1):
        _lib.tuning_set("nms_debug", dbg)
        row.append(dev_time_us(lambda: _C.nms_batched_mask(b, s, o, 2000, 0.7)))
    _lib.tuning_set("nms_debug", 0)
    tiles = sum(((len(x[1]) + 63) // 64) * ((len(x[1]) + 63) // 64 + 1) // 2 for x in segs)
    print("%-22s tiles %5d   full %7.2f us   without the IoU loop %7.2f us" % (name, tiles, row[0], row[1]))

# ---- timeline of one launch (segment 0 of the model's segments; nms_debug & 4), with and without the IoU loop
import ctypes
def timeline(dbg):
    b, s, o = case(base)
    _lib.tuning_set("nms_debug", 4 | dbg)
    for _ in range(3):
        _C.nms_batched_mask(b, s, o, 2000, 0.7)
    torch.cuda.synchronize()
    n = 128 + 3 * 2112
    buf = (ctypes.c_int64 * n)()
    rc = _lib.lib.detops_debug_nms_timeline(buf, n)
    _lib.tuning_set("nms_debug", 0)
    assert rc == 0, rc
    t = np.array(buf[:], dtype=np.int64)
    us = lambda x: (x - t[0]) / 100.0
    print("== nms_debug=%d: sort start 0.0 | token published %.1f | scan wg start %.1f | scan saw the token %.1f | chain end %.1f | scan wg end %.1f (us)"
          % (dbg, us(t[1]), us(t[2]), us(t[3]), us(t[4]), us(t[5])))
    nb = 32
    tiles = t[128:128 + 3 * 528].reshape(528, 3)
    st, go, dn = us(tiles[:, 0]), us(tiles[:, 1]), us(tiles[:, 2])
    idx, line = 0, []
    for rb in range(nb):           # tiles of row block rb: a run of nb - rb in the row-major upper triangle
        line.append("%.1f" % dn[idx:idx + nb - rb].max())
        idx += nb - rb
    print("   tiles: start %.1f..%.1f  go %.1f..%.1f  counted %.1f..%.1f" % (st.min(), st.max(), go.min(), go.max(), dn.min(), dn.max()))
    print("   last tile of row block rb counted at (us): " + " ".join(line))
for dbg in (0, 1):
    timeline(dbg)

# ---- A/B on one box: input in score order, presorted detection on / off
b, s, o = case(base)
for rep in range(3):
    row = []
    for off in (0, 1):
        _lib.tuning_set("nms_no_presorted", off)
        row.append(dev_time_us(lambda: _C.nms_batched_mask(b, s, o, 2000, 0.7), 300))
    print("A/B model segments in score order: network skipped %.2f us | network forced %.2f us" % tuple(row))
_lib.tuning_set("nms_no_presorted", 0)
