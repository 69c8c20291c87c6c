# round-1 third GPU call: parity of the new ROIAlign kernels, opbench, e2e variants
set -x
mkdir -p gpurun_out
export MIOPEN_LOG_LEVEL=1
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -15 gpurun_out/pytest_gpu.log
timeout 300 python tools/opbench.py --iters 30 --only roi_align,nms --json gpurun_out/opbench.json > gpurun_out/opbench.log 2>&1
cat gpurun_out/opbench.log | tail -25
timeout 500 python bench.py --steps 10 --warmup 4 > gpurun_out/bench_f32.log 2>&1; tail -1 gpurun_out/bench_f32.log | cut -c1-400
timeout 400 python bench.py --steps 10 --warmup 4 --channels-last --no-cpu-baseline > gpurun_out/bench_f32_cl.log 2>&1; tail -1 gpurun_out/bench_f32_cl.log | cut -c1-300
timeout 400 python bench.py --steps 10 --warmup 4 --dtype bfloat16 --no-cpu-baseline > gpurun_out/bench_bf16.log 2>&1; tail -1 gpurun_out/bench_bf16.log | cut -c1-300
