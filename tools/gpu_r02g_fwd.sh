# LDS-DMA ROIAlign forward + ROI ranking pre-pass: parity on the device, A/B, kernel stats
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "roi_align or pooler or golden or fixture" 2>&1 | tail -4
OPBENCH_FWD_KB=${FWD_KB:-16} OPBENCH_FWD_WPS=${FWD_WPS:-5} timeout 400 python tools/opbench.py --iters 30 --only roi_align_fwd --json gpurun_out/opbench_fwd.json > gpurun_out/opbench_fwd.log 2>&1
grep -v "^/opt" gpurun_out/opbench_fwd.log | grep "roi_align_fwd" | cut -c1-190
rm -rf gpurun_out/prof_fwd
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_fwd -o fwd -- python tools/opbench.py --iters 20 --only roi_align_fpn > gpurun_out/prof_fwd.log 2>&1
python - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/prof_fwd/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:10]:
        print(r['Name'][:90], r['Calls'], r['AverageNs'], r['Percentage'])
PY
find gpurun_out/prof_fwd -name "*kernel_trace.csv" -delete
