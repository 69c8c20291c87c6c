# round-1 second GPU call: parity suite + first end-to-end bench line + rocprof of the bench
set -x
mkdir -p gpurun_out
export MIOPEN_LOG_LEVEL=1
python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -3 gpurun_out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
tail -4 gpurun_out/smoke.log
timeout 900 python bench.py --steps 10 --warmup 4 > gpurun_out/bench_f32.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench_f32.log
tail -3 gpurun_out/bench_f32.log
timeout 600 python bench.py --steps 10 --warmup 4 --dtype bfloat16 --no-cpu-baseline > gpurun_out/bench_bf16.log 2>&1
tail -2 gpurun_out/bench_bf16.log
R=$PWD; cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -o bench -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline > $R/gpurun_out/rocprof_bench.log 2>&1
cd $R; tail -2 gpurun_out/rocprof_bench.log; ls gpurun_out/prof_bench | head
