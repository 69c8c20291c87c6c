# r01d: parity of the new kernels, default bench line, A/B micro-benchmarks, PMC traffic, step profile,
# then (optional, time permitting) MIOpen search mode + tuning-db export, RetinaNet and bf16 benches.
set -x
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
mkdir -p gpurun_out
export MIOPEN_LOG_LEVEL=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log | cut -c1-220; el pytest
timeout 240 python bench.py > gpurun_out/bench_f32.log 2>&1; grep -E "^\{" gpurun_out/bench_f32.log | cut -c1-3500; grep "bench " gpurun_out/bench_f32.log | tail -3; el bench
timeout 200 python tools/opbench.py --iters 30 --json gpurun_out/opbench.json > gpurun_out/opbench.log 2>&1
cut -c1-220 gpurun_out/opbench.log | grep -v "^/opt" | head -90; el opbench
timeout 100 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o x -- python tools/opbench.py --iters 3 --only roi_align_fpn > gpurun_out/pmc_fetch.log 2>&1
timeout 100 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o x -- python tools/opbench.py --iters 3 --only roi_align_fpn > gpurun_out/pmc_write.log 2>&1
python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/traffic.json 2>&1 | cut -c1-200
find gpurun_out/pmc_fetch gpurun_out/pmc_write -name "*kernel_trace.csv" -delete; el pmc
timeout 220 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -o bench -- python bench.py --steps 5 --warmup 3 --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1
T=$(find gpurun_out/prof_bench -name "*kernel_trace.csv" | head -1); python tools/trace_steps.py $T 4 70 > gpurun_out/step_breakdown.txt 2>&1; head -24 gpurun_out/step_breakdown.txt | cut -c1-180
find gpurun_out/prof_bench -name "*kernel_trace.csv" -delete; el rocprof
# ---- optional tail (each bounded; results are extras)
timeout 420 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --miopen-search --export-miopen-db gpurun_out/miopen_db > gpurun_out/bench_f32_search.log 2>&1; grep -E "^\{" gpurun_out/bench_f32_search.log | cut -c1-600; grep "bench " gpurun_out/bench_f32_search.log | tail -2
du -sm gpurun_out/miopen_db gpurun_out/miopen_db/* 2>/dev/null
if [ "$(du -sm gpurun_out/miopen_db 2>/dev/null | cut -f1)" -gt 40 ]; then rm -rf gpurun_out/miopen_db/cache; fi; el search
timeout 150 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --config retinanet/retinanet_R-50-FPN_1x.yaml > gpurun_out/bench_retinanet.log 2>&1; grep -E "^\{" gpurun_out/bench_retinanet.log | cut -c1-1500; el retinanet
timeout 150 python bench.py --steps 10 --warmup 3 --dtype bfloat16 --no-cpu-baseline > gpurun_out/bench_bf16.log 2>&1; grep -E "^\{" gpurun_out/bench_bf16.log | cut -c1-800; el bf16
du -sm gpurun_out
