#!/bin/bash
# runtime environment knobs on the whole step: HIP_FORCE_DEV_KERNARG (kernel arguments in device memory)
cd /root/repo
B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline"
one() { "$@" 2>&1 | grep -E "^\{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['dtype'], d['value'], 'img/s', d['ms_per_step'], 'ms host', d.get('host_enqueue_ms_per_step'))"; }
for v in 0 1; do
  echo "== HIP_FORCE_DEV_KERNARG=$v"
  HIP_FORCE_DEV_KERNARG=$v one timeout 300 $B
  HIP_FORCE_DEV_KERNARG=$v one timeout 300 $B --dtype bfloat16
done | tee gpurun_out/r04ak_env_knobs.log
