T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
mkdir -p gpurun_out; export MIOPEN_LOG_LEVEL=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "forced_ddp or cfg5 or fused_kernel or train_net" 2>&1 | tail -5; el pytest
run() { timeout 300 python bench.py --steps 40 --warmup 12 --no-cpu-baseline --no-kernel-timing "$@" 2>gpurun_out/r03n_err.log | grep -E "^\{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'], 'host', d['host_enqueue_ms_per_step'], d['ddp'])" || tail -5 gpurun_out/r03n_err.log; }
run; el plain
run --force-ddp; el bdp
run --force-ddp --bucket-mb 8; el bdp-8mb
run; el plain
run --force-ddp; el bdp
run --dtype float16; el f16
run --dtype float16 --force-ddp; el f16-bdp
