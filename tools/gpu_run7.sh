set -x
mkdir -p gpurun_out
export MIOPEN_LOG_LEVEL=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 240 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -6 gpurun_out/pytest_gpu.log | cut -c1-220
timeout 300 python bench.py --steps 10 --warmup 4 > gpurun_out/bench_f32.log 2>&1; grep -E "^\{" gpurun_out/bench_f32.log | cut -c1-3000
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -o bench -- python bench.py --steps 5 --warmup 3 --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1
T=$(find gpurun_out/prof_bench -name "*kernel_trace.csv" | head -1); python tools/trace_steps.py $T 4 70 > gpurun_out/step_breakdown.txt 2>&1; head -30 gpurun_out/step_breakdown.txt | cut -c1-200
find gpurun_out/prof_bench -name "*kernel_trace.csv" -size +30M -delete
timeout 150 python tools/opbench.py --iters 30 --json gpurun_out/opbench.json > gpurun_out/opbench.log 2>&1
cut -c1-200 gpurun_out/opbench.log | tail -60
timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o x -- python tools/opbench.py --iters 3 --only roi_align > gpurun_out/pmc_fetch.log 2>&1
timeout 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o x -- python tools/opbench.py --iters 3 --only roi_align > gpurun_out/pmc_write.log 2>&1
python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/traffic.json 2>&1 | cut -c1-200
find gpurun_out/pmc_fetch gpurun_out/pmc_write -name "*kernel_trace.csv" -delete
timeout 200 python bench.py --steps 10 --warmup 4 --dtype bfloat16 --no-cpu-baseline > gpurun_out/bench_bf16.log 2>&1; grep -E "^\{" gpurun_out/bench_bf16.log | cut -c1-1500
