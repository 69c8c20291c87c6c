# Round 6, experiment Q: cfg-5 (R-101 + DCN, fp16) with the DCN layers reading / writing channels-last tensors in place;
# shipped find-db (half-precision NHWC keys included); the GPU test suite on the new defaults.
O=gpurun_out/r06q; mkdir -p $O; export MIOPEN_LOG_LEVEL=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
jl() { grep -E "^\{" "$1" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-28s' % '$2', d['value'], 'img/s', d['ms_per_step'], 'ms', 'loss_finite', d['loss_finite'], d.get('layout'), d['miopen']['db'], d.get('kernel_families_ms_per_step'))" 2>/dev/null || tail -3 "$1"; }
CFG5="--config e2e_mask_rcnn_R_101_FPN_1x.yaml --dtype float16 MODEL.RESNETS.STAGE_WITH_DCN (False,True,True,True)"
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline"
for lay in backbone all nchw; do
  timeout 400 $B --layout $lay $CFG5 < /dev/null > $O/cfg5_$lay.log 2>&1; jl $O/cfg5_$lay.log cfg5-$lay
done
timeout 400 $B --dtype bfloat16 < /dev/null > $O/bf16.log 2>&1; jl $O/bf16.log bf16-auto
timeout 400 $B --config retinanet/retinanet_R-50-FPN_1x.yaml < /dev/null > $O/retina.log 2>&1; jl $O/retina.log retinanet-auto
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider < /dev/null > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log | cut -c1-200
