# r02h: full GPU parity suite, smoke, opbench, kernel stats, PMC traffic + diagnosis, bench (default flags), step breakdown
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
mkdir -p gpurun_out
export MIOPEN_LOG_LEVEL=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
grep -E "passed|failed|rc=" gpurun_out/pytest_gpu.log | tail -3 | cut -c1-220; el pytest
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2; el smoke
timeout 400 python tools/opbench.py --iters 30 --json gpurun_out/opbench.json > gpurun_out/opbench.log 2>&1; grep -c "us " gpurun_out/opbench.log; el opbench
rm -rf gpurun_out/prof_ops gpurun_out/pmc_fetch gpurun_out/pmc_write
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ops -o ops -- python tools/opbench.py --iters 20 --only roi_align_fpn,nms,focal > gpurun_out/prof_ops.log 2>&1
python tools/kernel_times.py gpurun_out/prof_ops > gpurun_out/opbench_kernel_times.txt 2>&1; head -30 gpurun_out/opbench_kernel_times.txt | cut -c1-170
find gpurun_out/prof_ops -name "*kernel_trace.csv" -delete; el kernel-stats
bash tools/gpu_pmc_fwd.sh > /dev/null 2>&1
python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/traffic.json 2>&1 | grep roi_align | cut -c1-200; el pmc
timeout 500 python bench.py > gpurun_out/bench_f32.log 2>&1; grep -E "^\{" gpurun_out/bench_f32.log | cut -c1-3000; el bench
rm -rf gpurun_out/prof_bench
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -o bench -- python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-timing > gpurun_out/prof_bench.log 2>&1
python tools/trace_steps.py gpurun_out/prof_bench/bench_kernel_trace.csv 4 400 > gpurun_out/step_breakdown.txt 2>&1; head -3 gpurun_out/step_breakdown.txt
find gpurun_out/prof_bench -name "*kernel_trace.csv" -delete; el step-breakdown
