# Round 6, experiment Y: is the host cheaper per convolution with torch.backends.cudnn.benchmark = True (ATen's per-shape
# algorithm cache in front of MIOpen) than in immediate mode, given that the shipped find-db already holds every problem key?
O=gpurun_out/r06y; mkdir -p $O; export MIOPEN_LOG_LEVEL=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CFG5="--config e2e_mask_rcnn_R_101_FPN_1x.yaml --dtype float16 MODEL.RESNETS.STAGE_WITH_DCN (False,True,True,True)"
run() { timeout 900 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-timing "$@" < /dev/null > $O/$N.log 2>&1
  grep -E "^\{" $O/$N.log | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$N', d['value'], 'img/s', d['ms_per_step'], 'ms host', d['host_enqueue_ms_per_step'], d['miopen'])" 2>/dev/null || tail -3 $O/$N.log; }
for rep in 1 2; do
N=cfg5_immediate_$rep; run $CFG5
N=cfg5_search_$rep; run $CFG5 --miopen-search
N=bf16_immediate_$rep; run --dtype bfloat16
N=bf16_search_$rep; run --dtype bfloat16 --miopen-search
N=f32_immediate_$rep; run
N=f32_search_$rep; run --miopen-search
done
