# Round 6, experiment S: bias(+ReLU) fused pass for the half-precision storage types; bf16 / cfg-5 / RetinaNet-bf16 before-after
# (DETOPS_BIAS_ACT_HALF=0 restores the round's previous behaviour through the env switch read in layers/misc.py).
O=gpurun_out/r06s; mkdir -p $O; export MIOPEN_LOG_LEVEL=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
jl() { grep -E "^\{" "$1" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-28s' % '$2', d['value'], 'img/s', d['ms_per_step'], 'ms', 'loss_finite', d['loss_finite'], d.get('layout'), d['miopen']['db'])" 2>/dev/null || tail -3 "$1"; }
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "bias_act or float64" -p no:cacheprovider < /dev/null > $O/pytest_new.log 2>&1; tail -2 $O/pytest_new.log
CFG5="--config e2e_mask_rcnn_R_101_FPN_1x.yaml --dtype float16 MODEL.RESNETS.STAGE_WITH_DCN (False,True,True,True)"
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-timing"
for h in 0 1; do
  export DETOPS_BIAS_ACT_HALF=$h
  timeout 400 $B --dtype bfloat16 < /dev/null > $O/bf16_$h.log 2>&1; jl $O/bf16_$h.log mask-bf16-half$h
  timeout 400 $B $CFG5 < /dev/null > $O/cfg5_$h.log 2>&1; jl $O/cfg5_$h.log cfg5-half$h
  timeout 400 $B --config retinanet/retinanet_R-50-FPN_1x.yaml --dtype bfloat16 < /dev/null > $O/ret16_$h.log 2>&1; jl $O/ret16_$h.log retinanet-bf16-half$h
done
