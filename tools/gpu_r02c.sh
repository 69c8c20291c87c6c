# r02c: binned ROIAlign backward — parity tests, opbench A/B (binned CT16/CT32 vs scan), rocprofv3 kernel stats
set -x
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -k "roi_align or pooler or forced_ddp or train_step_finite" > gpurun_out/pytest_gpu_c.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_c.log
tail -12 gpurun_out/pytest_gpu_c.log | cut -c1-220; el pytest
timeout 200 python tools/opbench.py --iters 30 --only roi_align --json gpurun_out/opbench_c.json > gpurun_out/opbench_c.log 2>&1
grep -v "^/opt" gpurun_out/opbench_c.log | grep "roi_align" | cut -c1-200; el opbench
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bwd -o bwd -- python tools/opbench.py --iters 20 --only roi_align_fpn > gpurun_out/prof_bwd.log 2>&1
python - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/prof_bwd/**/*kernel_stats.csv', recursive=True):
    rows=list(csv.DictReader(open(f)))
    for r in rows[:12]:
        print(r['Name'][:90], r['Calls'], r['AverageNs'], r['Percentage'])
PY
el prof
