"""cfg-1 (BASELINE configs[0]) ROIAlign backward: device time per (bins, groups) and, under rocprofv3, its kernels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "maskrcnn-benchmark_amd")):
    sys.path.insert(0, p)
import torch
import synth
from opbench import dev_time_us, tune, _t
from maskrcnn_benchmark import _C as C
inp, rois, scale = synth.cfg1_roi_align()
tr = _t(rois)
K, Cc = rois.shape[0], inp.shape[1]
groups = [int(g) for g in (sys.argv[1] if len(sys.argv) > 1 else "0").split(",")]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
for ph, sr in ((7, 2), (14, 2), (7, 0)):
    g = torch.randn(K, Cc, ph, ph, device="cuda")
    alg = 4 * K * Cc * ph * ph + 4 * inp.size + 20 * K
    for grp in groups:
        tune("roi_bwd_groups", grp)
        us = dev_time_us(lambda: C.roi_align_backward(g, tr, scale, ph, ph, 1, Cc, 14, 14, sr), iters)
        print("cfg1 bwd %dx%d sr%d groups=%d: %.2f us  %.1f GB/s (%.3f of 8 TB/s)" % (ph, ph, sr, grp, us, alg / us / 1e3, alg / us / 8e6))
    tune("roi_bwd_groups", 0)
