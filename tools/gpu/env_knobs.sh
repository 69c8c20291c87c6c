#!/bin/bash
# runtime environment knobs on the whole fp32 step
cd /root/repo
B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline"
one() { "$@" 2>&1 | grep -E "^\{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['dtype'], d['value'], 'img/s', d['ms_per_step'], 'ms host', d.get('host_enqueue_ms_per_step'))"; }
echo "== baseline"; one timeout 300 $B
echo "== HSA_NO_SCRATCH_RECLAIM=1"; HSA_NO_SCRATCH_RECLAIM=1 one timeout 300 $B
echo "== GPU_MAX_HW_QUEUES=1"; GPU_MAX_HW_QUEUES=1 one timeout 300 $B
echo "== GPU_MAX_HW_QUEUES=4"; GPU_MAX_HW_QUEUES=4 one timeout 300 $B
