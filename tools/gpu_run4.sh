# round-1 fourth GPU call: parity (all), opbench A/B of the ROIAlign forward kernels, DCN, frozen BN; one e2e bench
set -x
mkdir -p gpurun_out
export MIOPEN_LOG_LEVEL=1
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -12 gpurun_out/pytest_gpu.log | cut -c1-220
timeout 300 python tools/opbench.py --iters 30 --only roi_align,frozen_bn,dcn --json gpurun_out/opbench.json > gpurun_out/opbench.log 2>&1
grep -v amdgpu.ids gpurun_out/opbench.log | cut -c1-200
DETOPS_ROIALIGN_FWD=generic timeout 300 python tools/opbench.py --iters 30 --only roi_align > gpurun_out/opbench_generic_fwd.log 2>&1
grep fwd gpurun_out/opbench_generic_fwd.log | cut -c1-200
timeout 500 python bench.py --steps 10 --warmup 4 > gpurun_out/bench_f32.log 2>&1; tail -1 gpurun_out/bench_f32.log | cut -c1-1500
