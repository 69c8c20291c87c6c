"""Per-hit latency of the ring backward's hit chain: K ROIs stacked on ONE 8 x 32 tile of the stride-16 level, hit list never
split -> device time / K = one hit's serial latency; ablations through roi_bwd_debug (1: no walk, 2: no staging)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "maskrcnn-benchmark_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
import synth
from opbench import dev_time_us, tune, _t
from maskrcnn_benchmark import _C as C
shapes = [(2, 256, h, w) for (h, w) in synth.fpn_shapes()[:4]]
scales = [1.0 / s for s in synth.FPN_STRIDES[:4]]
rng = np.random.RandomState(0)
for K in (64, 256):
    # ROIs of ~14 px on P4 (224 px in the image), all inside tile (rows 8..15, cols 32..63) of image 0
    cx = rng.uniform(16 * 40, 16 * 56, K); cy = rng.uniform(16 * 10, 16 * 13, K)
    rois = np.stack([np.zeros(K), cx - 112, cy - 112, cx + 112, cy + 112], 1).astype(np.float32)
    lv = synth.level_map(rois)
    assert (lv == 2).all(), np.bincount(lv)
    g = torch.randn(K, 256, 7, 7, device="cuda")
    tr, tl = _t(rois), _t(lv)
    for ct in (16, 32):
        for ring in (2, 3, 4):
            for dbg in (0, 1, 2, 3):
                tune("roi_bwd_seg", 1 << 20); tune("roi_bwd_ct", ct); tune("roi_bwd_ring", ring); tune("roi_bwd_debug", dbg)
                us = dev_time_us(lambda: C.roi_align_fpn_backward(g, tr, tl, shapes, scales, 7, 7, 2), 20)
                print("K=%d ct=%d ring=%d debug=%d: %.1f us  -> %.3f us per hit" % (K, ct, ring, dbg, us, us / K))
