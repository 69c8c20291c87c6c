T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export MIOPEN_LOG_LEVEL=1; O=gpurun_out/round; mkdir -p $O
C5='--config e2e_mask_rcnn_R_101_FPN_1x.yaml --dtype float16 --steps 40 --warmup 12 --no-cpu-baseline'
run() { timeout 300 python bench.py "$@" MODEL.RESNETS.STAGE_WITH_DCN "(False, True, True, True)" < /dev/null 2>/tmp/err.log | grep -E "^\{" > /tmp/line.json; python -c "
import json
d=json.load(open('/tmp/line.json')); print('$TAG', d['value'], d['ms_per_step'], 'host', d['host_enqueue_ms_per_step'], d['loss_finite'], d['miopen']['db'])" || tail -3 /tmp/err.log; }
TAG=shipped run $C5; cp /tmp/line.json $O/bench_cfg5.json; el
mkdir -p /tmp/e1; TAG=nodb MIOPEN_USER_DB_PATH=/tmp/e1 run $C5 --no-kernel-timing; el
TAG=shipped-notimer run $C5 --no-kernel-timing; el
timeout 200 python tools/host_profile.py --config e2e_mask_rcnn_R_101_FPN_1x.yaml --dtype float16 --steps 20 MODEL.RESNETS.STAGE_WITH_DCN "(False, True, True, True)" < /dev/null 2>&1 | grep -v amdgpu.ids | head -12 | cut -c1-150; el
