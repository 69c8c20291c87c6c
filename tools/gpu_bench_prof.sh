# bench line + per-step rocprofv3 breakdown (no test suite)
set -x
mkdir -p gpurun_out
export MIOPEN_LOG_LEVEL=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 120 python -m pytest tests/test_targets_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
timeout 400 python bench.py --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/bench_f32.log 2>&1; grep -E "^\{" gpurun_out/bench_f32.log | cut -c1-600
rm -rf gpurun_out/prof_bench
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -o bench -- python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-timing > gpurun_out/prof_bench.log 2>&1
python tools/trace_steps.py gpurun_out/prof_bench/bench_kernel_trace.csv 4 400 > gpurun_out/step_breakdown.txt 2>&1
head -3 gpurun_out/step_breakdown.txt; grep -i "match_kernel\|sampler\|mask_targ\|prep_kernel\|binned\|fwd_lds\|nms_\|focal" gpurun_out/step_breakdown.txt | cut -c1-140
