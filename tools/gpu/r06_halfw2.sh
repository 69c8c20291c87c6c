# Round 6 (late): per-STAGE half-weight casts (layers/half_weights.py) — single process and under the data-parallel wrapper
# (--force-ddp: world size 1, the wrapper's hooks / buckets / side stream as at N > 1); A/B with DETOPS_HALF_WEIGHTS=0
O=gpurun_out/r06halfw2; mkdir -p $O; export MIOPEN_LOG_LEVEL=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_model_gpu.py -q -x -k "half_weights or dynamic_slots" -p no:cacheprovider < /dev/null > $O/pytest.log 2>&1; grep -E "passed|failed|Error|assert" $O/pytest.log | tail -4 | cut -c1-220
CFG5="--config e2e_mask_rcnn_R_101_FPN_1x.yaml --dtype float16"
D="MODEL.RESNETS.STAGE_WITH_DCN (False,True,True,True)"
run() { timeout 600 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-kernel-timing --no-fixed-quota-line "$@" < /dev/null > $O/$N.log 2>&1
  grep -E "^\{" $O/$N.log | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$N', d['value'], 'img/s', d['ms_per_step'], 'ms host', d['host_enqueue_ms_per_step'], d['loss_finite'], d.get('ddp'), d.get('ddp_comm'))" 2>/dev/null || tail -5 $O/$N.log; }
for rep in 1 2; do
export DETOPS_HALF_WEIGHTS=0
N=bf16_percast_$rep; run --dtype bfloat16
N=cfg5_percast_$rep; run $CFG5 $D
N=bf16_ddp_percast_$rep; run --dtype bfloat16 --force-ddp
N=cfg5_ddp_percast_$rep; run $CFG5 --force-ddp $D
export DETOPS_HALF_WEIGHTS=1
N=bf16_stage_$rep; run --dtype bfloat16
N=cfg5_stage_$rep; run $CFG5 $D
N=bf16_ddp_stage_$rep; run --dtype bfloat16 --force-ddp
N=cfg5_ddp_stage_$rep; run $CFG5 --force-ddp $D
done
