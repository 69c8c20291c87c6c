# r02i: final state of round 2 — full GPU parity suite, smoke, bench (default flags), bf16 extra, forced-DDP line
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
mkdir -p gpurun_out
export MIOPEN_LOG_LEVEL=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
grep -E "passed|failed|rc=" gpurun_out/pytest_gpu.log | tail -3 | cut -c1-220; el pytest
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1; tail -2 gpurun_out/smoke.log; el smoke
timeout 500 python bench.py > gpurun_out/bench_f32.log 2>&1; grep -E "^\{" gpurun_out/bench_f32.log | cut -c1-400; el bench
timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --dtype bfloat16 > gpurun_out/bench_bf16.log 2>&1; grep -E "^\{" gpurun_out/bench_bf16.log | cut -c1-300; el bf16
timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --force-ddp > gpurun_out/bench_ddp.log 2>&1; grep -E "^\{" gpurun_out/bench_ddp.log | cut -c1-300; el ddp
