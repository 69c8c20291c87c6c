# Round 6, experiment U: the ring backward's channels-last store epilogue through LDS (whole 128-byte pieces per pixel) against
# the direct form (roi_bwd_debug=256: 16 bytes per lane at a 1 KB stride); PMC traffic of the NHWC forward / ring-NHWC backward.
O=gpurun_out/r06u; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_ops_gpu.py -q -x -k "channels_last" -p no:cacheprovider < /dev/null > $O/pytest.log 2>&1; tail -2 $O/pytest.log
OB="python tools/opbench.py --only roi_sets --sets model-random-init --layout both --iters 50"
for dbg in 0 256; do
  timeout 300 $OB --dir bwd --tune roi_bwd_debug=$dbg < /dev/null > $O/opbench_bwd_$dbg.log 2>&1; echo "== roi_bwd_debug=$dbg"; grep roi_align $O/opbench_bwd_$dbg.log | cut -c1-170
done
timeout 300 $OB --dir fwd < /dev/null > $O/opbench_fwd.log 2>&1; grep roi_align $O/opbench_fwd.log | cut -c1-170
for d in fwd bwd; do
  PM="python tools/opbench.py --only roi_sets --sets model-random-init --layout nhwc --heads box --dir $d --iters 5"
  for pass in "fw:FETCH_SIZE" "ww:WRITE_SIZE" "tcc:TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
    n=${pass%%:*}; c=${pass#*:}; rm -rf /tmp/pmc_${d}_$n
    timeout 150 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_${d}_$n -o x -- $PM < /dev/null > $O/pmc_${d}_$n.log 2>&1
  done
  python tools/pmc_traffic.py /tmp/pmc_${d}_fw /tmp/pmc_${d}_ww $O/traffic_nhwc_$d.json 2>&1 | cut -c1-170 > $O/traffic_nhwc_$d.txt; grep -i "roi_align\|roi_nhwc\|ring" $O/traffic_nhwc_$d.txt | head
  python - <<PY
import csv,glob,collections
for f in glob.glob('/tmp/pmc_${d}_tcc/**/*counter_collection.csv', recursive=True):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:60]
        if 'roi' not in k: continue
        agg[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[(k,r['Counter_Name'])]+=1
    for k,v in agg.items():
        print(k, {c: round(x/cnt[(k,c)]) for c,x in v.items()})
PY
done
