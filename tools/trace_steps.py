"""Per-training-step kernel breakdown from a rocprofv3 --kernel-trace CSV of bench.py.

    python tools/trace_steps.py gpurun_out/prof_bench/bench_kernel_trace.csv [steps=4] [top=40]

The --stats summary of a whole bench.py process is dominated by MIOpen's algorithm search during
warm-up; this tool cuts the trace at the once-per-step box-head ROIAlign forward launch and
aggregates only the last `steps` full iterations."""
import collections
import csv
import sys


def main(path, steps=4, top=40):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the box head's ROIAlign forward: the NCHW kernel carries its 7 x 7 bins in the name, the channels-last kernel
    # (roi_align_fwd_nhwc_kernel<V, kOutNhwc = false, ...>) is the box head's when its output is not channels-last
    def box_fwd(name):
        return "roi_align_fwd" in name and ("7, 7" in name or ("nhwc_kernel<" in name and ", false" in name))
    marks = [i for i, r in enumerate(rows) if box_fwd(r["Kernel_Name"])]
    if len(marks) < steps + 1:
        raise SystemExit("not enough steps in the trace (%d markers)" % len(marks))
    a, b = marks[-steps - 1], marks[-1]
    sel = rows[a:b]
    wall = (int(rows[b]["Start_Timestamp"]) - int(sel[0]["Start_Timestamp"])) / 1e6 / steps
    agg = collections.defaultdict(lambda: [0, 0])
    for r in sel:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        agg[r["Kernel_Name"][:120]][0] += d
        agg[r["Kernel_Name"][:120]][1] += 1
    busy = sum(v[0] for v in agg.values()) / 1e6 / steps
    print("steps=%d  wall %.2f ms/step  kernel-busy %.2f ms/step  launches/step %.0f" % (steps, wall, busy, len(sel) / steps))
    print("%10s %10s %10s  kernel" % ("ms/step", "calls/step", "avg_us"))
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:top]:
        print("%10.3f %10.1f %10.1f  %s" % (v[0] / 1e6 / steps, v[1] / steps, v[0] / v[1] / 1e3, k))
    # idle time of the device inside the window: gaps between the end of everything launched so far and the next start
    gaps, end = [], int(sel[0]["End_Timestamp"])
    for prev, r in zip(sel, sel[1:]):
        st = int(r["Start_Timestamp"])
        if st > end:
            gaps.append((st - end, prev["Kernel_Name"][:60], r["Kernel_Name"][:60]))
        end = max(end, int(r["End_Timestamp"]))
    tot = sum(g[0] for g in gaps)
    hist = collections.Counter(min(int(g[0] / 1000) // 5 * 5, 50) for g in gaps)
    print("\ndevice idle inside the window: %.2f ms/step in %d gaps/step; > 20 us: %.2f ms/step in %d gaps/step" % (
        tot / 1e6 / steps, len(gaps) / steps, sum(g[0] for g in gaps if g[0] > 20000) / 1e6 / steps,
        sum(1 for g in gaps if g[0] > 20000) / steps))
    print("gap histogram (us bucket: count/step): " + ", ".join("%d+: %.0f" % (k, v / steps) for k, v in sorted(hist.items())))
    print("largest gaps (us, after kernel -> before kernel):")
    for g in sorted(gaps, reverse=True)[:12]:
        print("%9.1f  %s  ->  %s" % (g[0] / 1e3, g[1], g[2]))


if __name__ == "__main__":
    main(sys.argv[1], *(int(a) for a in sys.argv[2:]))
