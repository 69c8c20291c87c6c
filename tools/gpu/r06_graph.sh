# Round 6: the training iteration replayed from HIP graphs (engine/graph_step.py, bench.py --hip-graph)
O=gpurun_out/r06graph; mkdir -p $O; export MIOPEN_LOG_LEVEL=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_model_gpu.py -q -x -k "graphed" -p no:cacheprovider < /dev/null > $O/pytest.log 2>&1; grep -E "passed|failed|Error|assert" $O/pytest.log | tail -6 | cut -c1-220
CFG5="--config e2e_mask_rcnn_R_101_FPN_1x.yaml --dtype float16"
D="MODEL.RESNETS.STAGE_WITH_DCN (False,True,True,True)"
run() { timeout 600 python bench.py --steps 40 --warmup 10 --no-cpu-baseline "$@" < /dev/null > $O/$N.log 2>&1
  grep -E "^\{" $O/$N.log | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$N', d['value'], 'img/s', d['ms_per_step'], 'ms host', d['host_enqueue_ms_per_step'], 'loss_finite', d['loss_finite'], d.get('hip_graph'))" 2>/dev/null || tail -5 $O/$N.log; }
N=cfg5_eager; run $CFG5 $D
N=cfg5_graph; run $CFG5 --hip-graph $D
N=bf16_eager; run --dtype bfloat16
N=bf16_graph; run --dtype bfloat16 --hip-graph
N=f32_eager; run
N=f32_graph; run --hip-graph
N=retina_graph; run --config retinanet/retinanet_R-50-FPN_1x.yaml --hip-graph
