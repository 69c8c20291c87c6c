"""HBM traffic per launch from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE need separate passes:
TCC has 4 slots, FETCH_SIZE takes 3 and WRITE_SIZE 2 — /opt/skills/guides/MI355X_MICROARCH.md).

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out/fetch -o x -- python tools/opbench.py ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out/write -o x -- python tools/opbench.py ...
    python tools/pmc_traffic.py out/fetch out/write profiles/r01_traffic.json

Corrections applied (guide §HBM): counters are in KiB (x1024); on gfx950 FETCH_SIZE reports HALF the
bytes of a wide coalesced read (x2) — uncalibrated for narrow gathers, so the read side of a gather
kernel is an upper bound after the x2.  Output: {kernel substring: {fetch_bytes, write_bytes,
hbm_bytes, launches}} averaged per launch, for the hand-written kernels only."""
import collections
import csv
import glob
import json
import os
import sys

OURS = ("roi_align_fwd", "roi_align_bwd", "nms_", "focal_kernel", "im2col_kernel", "col2im", "frozen_bn", "roi_pool",
        "nchw_to_nhwc", "im2col_nhwc", "coord_nhwc", "sampleT", "dcn_fused_fwd")


def load(dirname, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for path in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if r.get("Counter_Name") != counter:
                continue
            name = r["Kernel_Name"]
            if not any(k in name for k in OURS):
                continue
            key = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].strip()
            key += "|grid=%s" % r.get("Grid_Size", r.get("Grid_Size_X", ""))
            acc[key][0] += float(r["Counter_Value"])
            acc[key][1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in acc.items() if v[1]}


def tag_fused_launches(res):
    """opbench's FPN-fused launches have the model's shapes (box head: 1024 ROIs 7x7 over P2-P5 of
    2x800x1344; mask head: 256 ROIs 14x14): they are the largest grid of their kernel family, and get
    the name bench.py's KernelTimer uses so that `roofline.traffic` can be looked up."""
    import re
    best = {}
    for k in res:
        m = re.match(r"roi_align_(fwd|bwd)\w*<(\d+), (\d+)", k)
        if not m or m.group(2) not in ("7", "14"):
            continue
        grid = int(k.rsplit("grid=", 1)[1] or 0)
        fam = (m.group(1), m.group(2), k.split("<")[0])
        if fam not in best or grid > best[fam][0]:
            best[fam] = (grid, k)
    for (d, ph, _), (_, k) in best.items():
        K = 1024 if ph == "7" else 256
        res[k]["tag"] = "roi_align_fpn_%s[K=%d,C=256,%sx%s]" % (d, K, ph, ph)
    # forward over a channels-last pyramid (csrc/roi_align_nhwc.hip): <V, kOutNhwc, threads> — the box head's 7 x 7 call returns
    # [K, C, PH, PW] (kOutNhwc = false), the mask head's 14 x 14 call a channels-last tensor; largest grid of each flavour
    nh = {}
    for k in res:
        m = re.match(r"roi_align_fwd_nhwc_kernel<\d+, (true|false), \d+>\|grid=(\d+)", k)
        if m and (m.group(1) not in nh or int(m.group(2)) > nh[m.group(1)][0]):
            nh[m.group(1)] = (int(m.group(2)), k)
    for flavour, (_, k) in nh.items():
        res[k]["tag"] = "roi_align_fpn_fwd[K=1024,C=256,7x7]" if flavour == "false" else "roi_align_fpn_fwd[K=256,C=256,14x14]"
    # opbench --only frozen_bn runs the res2-sized activation [2, 256, 200, 336] fp32 (n = 34,406,400; N*C = 512): the
    # residual forward is the largest FrozenBN member of a training step (bench.py names it the same way)
    for k in res:
        m = re.match(r"frozen_bn_(fwd|bwd)_kernel<float, 4, (true|false), (true|false)>\|grid=", k)
        if m:
            s_, r_ = int(m.group(2) == "true"), int(m.group(3) == "true")    # <T, V, kRelu, kRes>
            res[k]["tag"] = ("frozen_bn_fwd[n=34406400,nc=512,e=4,res=%d]" % r_ if m.group(1) == "fwd"
                             else "frozen_bn_bwd[n=34406400,nc=512,e=4,res=%d,relu=%d]" % (r_, s_))


def main(fetch_dir, write_dir, out):
    f = load(fetch_dir, "FETCH_SIZE")
    w = load(write_dir, "WRITE_SIZE")
    res = {}
    for k in sorted(set(f) | set(w)):
        fb = f.get(k, (0.0, 0))[0] * 1024 * 2
        wb = w.get(k, (0.0, 0))[0] * 1024
        res[k] = {"fetch_bytes": int(fb), "write_bytes": int(wb), "hbm_bytes": int(fb + wb),
                  "launches": max(f.get(k, (0, 0))[1], w.get(k, (0, 0))[1])}
        print("%-90s fetch %8.1f MB  write %8.1f MB" % (k[:90], fb / 1e6, wb / 1e6))
    tag_fused_launches(res)
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)


if __name__ == "__main__":
    main(*sys.argv[1:4])
