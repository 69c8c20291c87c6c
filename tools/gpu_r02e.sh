# r02e: target-assignment kernels (f2/f3) — GPU parity suite, bench, per-step breakdown
set -x
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
mkdir -p gpurun_out
export MIOPEN_LOG_LEVEL=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -12 gpurun_out/pytest_gpu.log | cut -c1-220; el pytest
timeout 400 python bench.py > gpurun_out/bench_f32.log 2>&1; grep -E "^\{" gpurun_out/bench_f32.log | cut -c1-2500; el bench
rm -rf gpurun_out/prof_bench
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -o bench -- python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-timing > gpurun_out/prof_bench.log 2>&1
python tools/trace_steps.py gpurun_out/prof_bench 8 > gpurun_out/step_breakdown.txt 2>&1 || true
head -60 gpurun_out/step_breakdown.txt; el prof
