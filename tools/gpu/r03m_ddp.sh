T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
mkdir -p gpurun_out; export MIOPEN_LOG_LEVEL=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { timeout 300 python bench.py --steps 40 --warmup 12 --no-cpu-baseline --no-kernel-timing "$@" 2>/dev/null | grep -E "^\{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'], 'host', d['host_enqueue_ms_per_step'], d['ddp'])"; }
run; el plain
DETOPS_DDP_STATIC=1 run --force-ddp; el ddp-static
DETOPS_DDP_VIEW=0 run --force-ddp; el ddp-noview
run --force-ddp; el ddp
