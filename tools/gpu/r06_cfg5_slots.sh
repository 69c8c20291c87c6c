# cfg-5 (host-bound eager step): mask head slots dynamic vs fixed, alternating
O=gpurun_out/r06cfg5slots; mkdir -p $O; export MIOPEN_LOG_LEVEL=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CFG5="--config e2e_mask_rcnn_R_101_FPN_1x.yaml --dtype float16"
D="MODEL.RESNETS.STAGE_WITH_DCN (False,True,True,True)"
run() { timeout 600 python bench.py --steps 80 --warmup 15 --no-cpu-baseline --no-kernel-timing --no-fixed-quota-line "$@" < /dev/null > $O/$N.log 2>&1
  grep -E "^\{" $O/$N.log | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$N', d['value'], 'img/s', d['ms_per_step'], 'ms host', d['host_enqueue_ms_per_step'], d['loss_finite'], d.get('mask_slots'))" 2>/dev/null || tail -5 $O/$N.log; }
for rep in 1 2 3; do
DETOPS_MASK_SLOTS=fixed N=cfg5_fixed_$rep run $CFG5 $D
DETOPS_MASK_SLOTS=dynamic N=cfg5_dynamic_$rep run $CFG5 $D
DETOPS_MASK_SLOTS=64,48 N=cfg5_forced_$rep run $CFG5 $D
done
