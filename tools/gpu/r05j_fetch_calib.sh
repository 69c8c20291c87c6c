# round 5: ONE calibrated HBM-traffic number per flagship kernel.
#   1. tools/gpu/fetch_calib.hip under --pmc FETCH_SIZE / TCC counters: known line fills -> the factor for the row-piece gather
#   2. the ROIAlign launches of tools/opbench.py (both ROI sets, 2 and 4 images) under FETCH_SIZE and WRITE_SIZE
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; O=gpurun_out/r05j; mkdir -p $O
C=$GRAFT_REPO_ROOT/tools/gpu/ab/fetch_calib
for pass in "f:FETCH_SIZE" "w:WRITE_SIZE" "t:TCC_MISS_sum TCC_HIT_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  n=${pass%%:*}; c=${pass#*:}; rm -rf /tmp/cal_$n
  timeout 120 rocprofv3 --pmc $c --output-format csv -d /tmp/cal_$n -o x -- $C > $O/calib_$n.log 2>&1
  f=$(find /tmp/cal_$n -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $O/calib_$n.csv
done
tail -1 $O/calib_f.log
python - <<'PY'
import csv, collections, glob
for n in "fwt":
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open("gpurun_out/r05j/calib_%s.csv" % n)):
        acc[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(n, k, "rows", len(v), "values", [round(x, 1) for x in v[:6]])
PY
for imgs in 2 4; do
PF="python tools/opbench.py --only roi_sets --iters 5 --sets model-random-init,synthetic-loguniform --images $imgs"
for pass in "f:FETCH_SIZE" "w:WRITE_SIZE"; do
  n=${pass%%:*}; c=${pass#*:}; rm -rf /tmp/roi_${n}_$imgs
  timeout 200 rocprofv3 --pmc $c --output-format csv -d /tmp/roi_${n}_$imgs -o x -- $PF < /dev/null > $O/roi_${n}_$imgs.log 2>&1
  f=$(find /tmp/roi_${n}_$imgs -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python - "$f" $O/roi_${n}_$imgs.csv <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "roi_align" in r["Kernel_Name"] or "roi_fwd" in r["Kernel_Name"] or "roi_bwd" in r["Kernel_Name"]]
w = csv.DictWriter(open(sys.argv[2], "w"), fieldnames=["Dispatch_Id", "Kernel_Name", "Grid_Size", "Counter_Name", "Counter_Value"])
w.writeheader()
for r in rows:
    w.writerow({k: (r[k].split("(")[0] if k == "Kernel_Name" else r[k]) for k in w.fieldnames})
print(sys.argv[2], len(rows), "rows")
PY
done; done
du -sh $O
