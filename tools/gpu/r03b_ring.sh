# r03b: first run of the ring backward on the device — parity subset, opbench on the three ROI sets, bench
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
mkdir -p gpurun_out; export MIOPEN_LOG_LEVEL=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -p no:cacheprovider -k "roi_align" > gpurun_out/pytest_roi.log 2>&1; tail -5 gpurun_out/pytest_roi.log | cut -c1-300; el pytest-roi
timeout 200 python tools/opbench.py --only roi_sets --iters 50 --json gpurun_out/opbench_roi_sets.json > gpurun_out/opbench_roi_sets.log 2>&1; grep -E "roi_align_bwd|Error|error" gpurun_out/opbench_roi_sets.log | cut -c1-200; el opbench
for S in 16 24 48 1000000; do
  DETOPS_TUNING="roi_bwd_seg=$S" timeout 100 python tools/opbench.py --only roi_sets --iters 30 2>&1 | grep -E "roi_align_bwd.*box" | sed "s/^/seg=$S /" | cut -c1-150
done; el seg-sweep
timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline > gpurun_out/bench_f32.log 2>&1; grep -E "^\{" gpurun_out/bench_f32.log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d['value'], d['ms_per_step'], d.get('roofline')); print({k:(v['mean_us']) for k,v in d['kernels'].items()})"; tail -3 gpurun_out/bench_f32.log | cut -c1-300; el bench
