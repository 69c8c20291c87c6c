# experiment: channels-last weights (MIOpen NHWC path) — step time and the conv share of the step
mkdir -p gpurun_out
export MIOPEN_LOG_LEVEL=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --steps 20 --warmup 8 --no-cpu-baseline --no-kernel-timing --channels-last > gpurun_out/bench_cl.log 2>&1; grep -E "^\{" gpurun_out/bench_cl.log | cut -c1-200; tail -3 gpurun_out/bench_cl.log | cut -c1-300
rm -rf gpurun_out/prof_cl
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cl -o bench -- python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-timing --channels-last > gpurun_out/prof_cl.log 2>&1
python tools/trace_steps.py gpurun_out/prof_cl/bench_kernel_trace.csv 4 400 > gpurun_out/step_breakdown_cl.txt 2>&1
head -45 gpurun_out/step_breakdown_cl.txt | cut -c1-150
find gpurun_out/prof_cl -name "*kernel_trace.csv" -delete
