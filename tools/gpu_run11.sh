# r01e: final defaults — full GPU parity suite, PMC traffic (fresh, read by bench.py), default bench line,
# full opbench (+ rocprofv3 kernel stats of it), per-step rocprofv3 breakdown of bench.py, smoke().
set -x
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
mkdir -p gpurun_out
export MIOPEN_LOG_LEVEL=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 240 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log | cut -c1-220; el pytest
timeout 90 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o x -- python tools/opbench.py --iters 3 --only roi_align_fpn > gpurun_out/pmc_fetch.log 2>&1
timeout 90 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o x -- python tools/opbench.py --iters 3 --only roi_align_fpn > gpurun_out/pmc_write.log 2>&1
python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/traffic.json 2>&1 | cut -c1-200
cp gpurun_out/traffic.json profiles/r01e_roi_align_traffic.json
find gpurun_out/pmc_fetch gpurun_out/pmc_write -name "*kernel_trace.csv" -delete; el pmc
timeout 200 python bench.py > gpurun_out/bench_f32.log 2>&1; grep -E "^\{" gpurun_out/bench_f32.log | cut -c1-3500; el bench
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_opbench -o opbench -- python tools/opbench.py --iters 20 --json gpurun_out/opbench.json > gpurun_out/opbench.log 2>&1
find gpurun_out/prof_opbench -name "*kernel_trace.csv" -delete; grep -c . gpurun_out/opbench.log; el opbench
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -o bench -- python bench.py --steps 5 --warmup 3 --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1
T=$(find gpurun_out/prof_bench -name "*kernel_trace.csv" | head -1); python tools/trace_steps.py $T 4 70 > gpurun_out/step_breakdown.txt 2>&1; head -12 gpurun_out/step_breakdown.txt | cut -c1-160
find gpurun_out/prof_bench -name "*kernel_trace.csv" -delete; el rocprof
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -3 gpurun_out/smoke.log | cut -c1-300; el smoke
timeout 100 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --config e2e_faster_rcnn_R_50_FPN_1x.yaml > gpurun_out/bench_faster.log 2>&1; grep -E "^\{" gpurun_out/bench_faster.log | cut -c1-400; el faster
du -sm gpurun_out
