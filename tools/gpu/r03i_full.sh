# r03i: full GPU parity suite + smoke + bench after the ring-backward rewrite; seg sweep
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
mkdir -p gpurun_out; export MIOPEN_LOG_LEVEL=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
grep -E "passed|failed|rc=|Error" gpurun_out/pytest_gpu.log | tail -5 | cut -c1-220; el pytest
for S in 16 24 48 64; do
  DETOPS_TUNING="roi_bwd_seg=$S" timeout 100 python tools/opbench.py --only roi_sets --dir bwd --iters 30 2>&1 | grep -E "roi_align_bwd" | sed "s/^/seg=$S /" | cut -c1-135
done; el seg-sweep
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1; tail -2 gpurun_out/smoke.log; el smoke
timeout 500 python bench.py --steps 40 --warmup 15 --no-cpu-baseline > gpurun_out/bench_f32.log 2>&1; grep -E "^\{" gpurun_out/bench_f32.log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print(d['value'], d['ms_per_step'], d.get('roofline')); print({k:(v['mean_us']) for k,v in d['kernels'].items()})"; el bench
