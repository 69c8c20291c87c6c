# r03a: where does the in-model ROIAlign backward lose its 30 %?  Model ROI sets -> opbench, per-launch trace of bench.py
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
mkdir -p gpurun_out; export MIOPEN_LOG_LEVEL=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python tools/dump_model_rois.py --steps 6 --out gpurun_out/model_rois.npz > gpurun_out/dump_rois.log 2>&1; tail -3 gpurun_out/dump_rois.log; el dump
timeout 200 python tools/opbench.py --only roi_sets --model-rois gpurun_out/model_rois.npz --iters 50 --json gpurun_out/opbench_roi_sets.json > gpurun_out/opbench_roi_sets.log 2>&1; cat gpurun_out/opbench_roi_sets.log | cut -c1-200; el opbench
DETOPS_ROIALIGN_BWD_DEBUG=128 timeout 200 python tools/opbench.py --only roi_sets --model-rois gpurun_out/model_rois.npz --iters 1 > gpurun_out/phase_clocks.log 2>&1; grep "bwd-binned" gpurun_out/phase_clocks.log | awk 'NR%24<8' | tail -40; el phase
timeout 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_bench -o bench -- python bench.py --steps 6 --warmup 4 --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1; grep -E "^\{" gpurun_out/prof_bench.log | cut -c1-300; el prof
python - <<'PY'
import csv, glob
rows = []
for f in glob.glob("gpurun_out/prof_bench/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for i, r in enumerate(rows):
    n = r["Kernel_Name"]
    if "roi_align" in n or "roi_bwd_prep" in n or "roi_order" in n:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        prev = rows[i - 1]
        gap = (int(r["Start_Timestamp"]) - int(prev["End_Timestamp"])) / 1e3
        print("%8.1f us  gap %7.1f  grid %-9s %-60s after %s" % (d, gap, r.get("Grid_Size_X", r.get("Grid_Size", "")), n.replace("(anonymous namespace)::", "")[:60], prev["Kernel_Name"][:50]))
PY
el list
timeout 300 python bench.py --steps 20 --warmup 10 --no-cpu-baseline > gpurun_out/bench_f32.log 2>&1; grep -E "^\{" gpurun_out/bench_f32.log | cut -c1-1500; el bench
