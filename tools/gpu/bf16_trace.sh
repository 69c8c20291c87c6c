#!/bin/bash
# kernel trace of the bf16 step: where a mixed-precision step's device time goes
cd /tmp && export TMPDIR=/tmp
P=/tmp/prof_bf16; rm -rf $P
timeout 500 rocprofv3 --kernel-trace --output-format csv -d $P -o bench -- python /root/repo/bench.py --dtype bfloat16 --steps 6 --warmup 6 --no-cpu-baseline < /dev/null > /tmp/prof_bf16.log 2>&1
T=$(find $P -name "*kernel_trace.csv" | head -1)
python /root/repo/tools/trace_steps.py "$T" 4 70 > /root/repo/gpurun_out/r04ai_bench_bf16_step_breakdown.txt 2>&1
head -75 /root/repo/gpurun_out/r04ai_bench_bf16_step_breakdown.txt | cut -c1-150
