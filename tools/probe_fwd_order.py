"""A/B: ROIAlign forward with ROIs pre-sorted on the host by various keys (device ranking pre-pass disabled)."""
import os, sys
sys.path[:0] = ["tools", "maskrcnn-benchmark_amd", "."]
import numpy as np, torch, synth
os.environ["DETOPS_ROIALIGN_FWD_ORDER"] = "0"
from maskrcnn_benchmark import _C as C
from opbench import dev_time_us

feats = [torch.randn(2, 256, h, w, device="cuda") for (h, w) in synth.fpn_shapes()[:4]]
scales = [1.0 / s for s in synth.FPN_STRIDES[:4]]
U = np.uint64
for K, ph in ((1024, 7), (256, 14)):
    rois = synth.fpn_rois(per_image=K // 2)
    lv = synth.level_map(rois)
    st = np.array(synth.FPN_STRIDES[:4], np.float32)[lv]
    cx = (rois[:, 1] + rois[:, 3]) * 0.5 / st; cy = (rois[:, 2] + rois[:, 4]) * 0.5 / st
    x1 = rois[:, 1] / st; y1 = rois[:, 2] / st
    img = rois[:, 0].astype(U)
    keys = {"random": None}
    for band in (4, 8, 16, 32, 64):
        keys[f"lvl,img,cy/{band},cx"] = (lv.astype(U) << U(40)) | (img << U(36)) | ((cy / band).astype(U) << U(16)) | cx.astype(U)
    keys["lvl,img,y1/16,x1"] = (lv.astype(U) << U(40)) | (img << U(36)) | ((y1 / 16).astype(U) << U(16)) | x1.astype(U)
    b = (cy / 16).astype(np.int64)
    snake = np.where(b % 2 == 0, cx, 4095 - cx).astype(U)
    keys["lvl,img,cy/16,snake cx"] = (lv.astype(U) << U(40)) | (img << U(36)) | (b.astype(U) << U(16)) | snake
    keys["img,lvl,cy/16,cx"] = (img << U(44)) | (lv.astype(U) << U(40)) | ((cy / 16).astype(U) << U(16)) | cx.astype(U)
    keys["lvl desc,img,cy/16,cx"] = ((3 - lv).astype(U) << U(40)) | (img << U(36)) | ((cy / 16).astype(U) << U(16)) | cx.astype(U)
    keys["lvl,img,cx/16,cy (column bands)"] = (lv.astype(U) << U(40)) | (img << U(36)) | ((cx / 16).astype(U) << U(16)) | cy.astype(U)
    for name, key in keys.items():
        r = rois if key is None else rois[np.argsort(key, kind="stable")]
        tr = torch.from_numpy(np.ascontiguousarray(r)).cuda()
        us = min(dev_time_us(lambda: C.roi_align_fpn_forward(feats, tr, scales, ph, ph, 2, 2, 5), 30) for _ in range(2))
        print(f"K={K} {ph}x{ph} order={name:34s}: {us:7.2f} us")
