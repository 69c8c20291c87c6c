"""Per (kernel, grid) mean duration from a rocprofv3 --kernel-trace CSV:  python tools/kernel_times.py <dir> [substr]"""
import collections, csv, glob, os, sys
d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(list)
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if sub and sub not in name:
            continue
        acc[(name[:70], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", ""))].append(
            (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0)
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print("%-72s grid=%-9s wg=%-5s n=%-4d mean=%8.1f us  median=%8.1f us" % (k[0], k[1], k[2], len(v), sum(v) / len(v), v[len(v) // 2]))
