# r02b: full GPU parity suite after promoting the A/B winners + the new cfg-5 / P2 / fp16 / forced-DDP tests,
# then the default bench line (100 timed / 20 warm-up steps) and the forced-DDP line.
set -x
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
mkdir -p gpurun_out
export MIOPEN_LOG_LEVEL=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=15 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -30 gpurun_out/pytest_gpu.log | cut -c1-220; el pytest
timeout 400 python bench.py > gpurun_out/bench_f32.log 2>&1; grep -E "^\{" gpurun_out/bench_f32.log | cut -c1-3000; el bench
timeout 300 python bench.py --force-ddp --steps 30 --warmup 10 --no-cpu-baseline > gpurun_out/bench_f32_forceddp.log 2>&1; grep -E "^\{" gpurun_out/bench_f32_forceddp.log | cut -c1-600; tail -3 gpurun_out/bench_f32_forceddp.log | cut -c1-300; el bench-ddp
