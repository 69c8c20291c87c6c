"""Work statistics of the ROIAlign backward kernel on the REAL box-head / mask-head workloads, from the
host emulation's counters (tests/emu, DETOPS_STAT): workgroups, ROI-scan rounds, hit ROIs, batches,
wave-level ROI tasks and FMA bodies for the union walk (default) and the per-lane walk (experimental).
CPU only; C is reduced (the per-channel-chunk structure repeats), map sizes and ROIs are the full ones.

    python tools/emu_workstats.py            # ~1-2 minutes
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import emu  # noqa: E402
import synth  # noqa: E402


def run(tag, K, ph, walk):
    # one 16-channel chunk (every chunk of the 256 repeats this work); gather3 owns G chunks per workgroup
    C = 64 if walk == "gather3" else 16
    shapes = [(2, C, h, w) for (h, w) in synth.fpn_shapes()[:4]]
    scales = [1.0 / s for s in synth.FPN_STRIDES[:4]]
    rois = synth.fpn_rois(per_image=K // 2)
    lv = synth.level_map(rois)
    g = np.random.RandomState(0).randn(K, C, ph, ph).astype(np.float32)
    os.environ["DETOPS_ROIALIGN_BWD"] = "gather3" if walk == "gather3" else "gather"
    os.environ["DETOPS_ROIALIGN_BWD_CT"] = "16"
    if walk == "gather3":
        os.environ["DETOPS_ROIALIGN_BWD_G"] = "4" if ph == 7 else "2"
    else:
        os.environ.pop("DETOPS_ROIALIGN_BWD_G", None)
    if walk == "lane":
        os.environ["DETOPS_ROIALIGN_BWD_WALK"] = "lane"
    else:
        os.environ.pop("DETOPS_ROIALIGN_BWD_WALK", None)
    emu.stats(reset=True)
    t = time.time()
    emu.roi_align_fpn_backward(g, rois, lv, shapes, scales, ph, ph, 2)
    st = emu.stats(reset=True)
    print("%s  walk=%s  (%.0f s)" % (tag, walk, time.time() - t))
    for k in sorted(st):
        print("    %-28s %12.0f" % (k, st[k]))
    if walk == "gather3":
        print("    (64 channels = 4 chunks of the default kernel at 7x7, 2 x 2 chunks at 14x14)")
        return
    wg, tasks = st.get("bwd.workgroups", 1), max(st.get("bwd.wave_roi_tasks", 1), 1)
    bodies = st.get("bwd.bodies_lane_walk", 0) + st.get("bwd.bodies_union_walk", 0)
    print("    per chunk: %.0f workgroups, %.2f hits / workgroup, %.1f bodies / wave-ROI task, lane utilisation %.0f %%" % (
        wg, st.get("bwd.hits", 0) / wg, bodies / tasks, 100.0 * st.get("bwd.active_lane_bodies", 0) / max(bodies * 64, 1)))


if __name__ == "__main__":
    for tag, K, ph in (("box head 1024 x 7x7", 1024, 7), ("mask head 256 x 14x14", 256, 14)):
        for walk in ("union", "lane", "gather3"):
            run(tag, K, ph, walk)
