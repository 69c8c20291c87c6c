set -x
mkdir -p gpurun_out
export MIOPEN_LOG_LEVEL=1
timeout 300 python tools/opbench.py --iters 30 --only roi_align,dcn --json gpurun_out/opbench.json > gpurun_out/opbench.log 2>&1
grep -E "roi_align|col2im " gpurun_out/opbench.log | cut -c1-200
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -6 gpurun_out/pytest_gpu.log | cut -c1-220
timeout 420 python bench.py --steps 10 --warmup 4 --no-cpu-baseline > gpurun_out/bench_f32.log 2>&1; grep -E "^\[bench|^\{" gpurun_out/bench_f32.log | cut -c1-600
timeout 300 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-miopen-search > gpurun_out/bench_f32_nosearch.log 2>&1; grep -E "^\[bench|^\{" gpurun_out/bench_f32_nosearch.log | cut -c1-600
