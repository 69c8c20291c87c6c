"""Per-kernel summary of rocprofv3 PMC passes (any counters): mean counter value per launch for the
hand-written kernels, one row per kernel, one column per counter, plus a few ratios when their
inputs are present (all per launch):

    rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU \\
              SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d out/sq -o x -- <cmd>
    rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum ...
    rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum ...
    python tools/pmc_diag.py out/sq out/tcp out/tcc > profiles/rNN_pmc_diag.txt

Counters go in separate passes (per-block slot limits, /opt/skills/guides/MI355X_MICROARCH.md
"rocprofv3 PMC slots"); never combine --pmc with the hip/hsa/memory trace domains."""
import collections
import csv
import glob
import os
import sys

OURS = ("roi_align", "nms_", "focal_kernel", "im2col_kernel", "col2im", "frozen_bn", "roi_pool", "psroi", "roi_order")


def load(dirs):
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for d in dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(path)):
                name = r["Kernel_Name"]
                if not any(k in name for k in OURS):
                    continue
                key = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].strip()
                key += "|grid=%s" % r.get("Grid_Size", "")
                cell = acc[key][r["Counter_Name"]]
                cell[0] += float(r["Counter_Value"])
                cell[1] += 1
    return {k: {c: v[0] / v[1] for c, v in cs.items() if v[1]} for k, cs in acc.items()}


def ratios(c):
    out = {}
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c and c["TCC_HIT_sum"] + c["TCC_MISS_sum"] > 0:
        out["L2_hit"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    if "TCP_TCC_READ_REQ_sum" in c and c.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0) > 0:
        out["L1_miss"] = c["TCP_TCC_READ_REQ_sum"] / c["TCP_TOTAL_CACHE_ACCESSES_sum"]
    if c.get("SQ_WAVE_CYCLES", 0) > 0:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if k in c:
                out[k + "/WAVE_CYCLES"] = c[k] / c["SQ_WAVE_CYCLES"]
    if c.get("SQ_LDS_IDX_ACTIVE", 0) > 0 and "SQ_LDS_BANK_CONFLICT" in c:
        out["LDS_conflict_frac"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
    return out


def main(dirs):
    table = load(dirs)
    counters = sorted({c for cs in table.values() for c in cs})
    for k in sorted(table):
        print(k)
        for c in counters:
            if c in table[k]:
                print("    %-36s %16.1f" % (c, table[k][c]))
        for name, v in ratios(table[k]).items():
            print("    %-36s %16.4f" % (name, v))


if __name__ == "__main__":
    main(sys.argv[1:])
