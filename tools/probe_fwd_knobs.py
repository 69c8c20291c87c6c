"""Sweep of the LDS-DMA forward's launch knobs (channels per workgroup, buffer size, register budget)."""
import os, sys, itertools
sys.path[:0] = ["tools", "maskrcnn-benchmark_amd", "."]
import numpy as np, torch, synth
from maskrcnn_benchmark import _C as C
from opbench import dev_time_us
feats = [torch.randn(2, 256, h, w, device="cuda") for (h, w) in synth.fpn_shapes()[:4]]
scales = [1.0 / s for s in synth.FPN_STRIDES[:4]]
GRID = {(1024, 7): (("32",), ("9", "10", "11", "12", "13", "15"), ("5", "6", "8")),
        (256, 14): (("16", "32"), ("13", "15", "19", "26", "31", "39"), ("5", "6", "8"))}
for (K, ph), (cts, kbs, wpss) in GRID.items():
    tr = torch.from_numpy(synth.fpn_rois(per_image=K // 2)).cuda()
    for ct, kb, wps in itertools.product(cts, kbs, wpss):
        os.environ.update(DETOPS_ROIALIGN_FWD_CT=ct, DETOPS_ROIALIGN_FWD_BUF_KB=kb, DETOPS_ROIALIGN_FWD_WPS=wps)
        us = min(dev_time_us(lambda: C.roi_align_fpn_forward(feats, tr, scales, ph, ph, 2, 2, 5), 20) for _ in range(2))
        print(f"K={K} {ph}x{ph} CT={ct:3s} buf=2x{kb:2s}KB wps={wps}: {us:7.2f} us")
