# Round 6 (late): layers/half_weights.py — one multi-tensor weight cast per mixed-precision step; A/B with DETOPS_HALF_WEIGHTS=0
O=gpurun_out/r06halfw; mkdir -p $O; export MIOPEN_LOG_LEVEL=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_model_gpu.py -q -x -k "half_weights or graphed" -p no:cacheprovider < /dev/null > $O/pytest.log 2>&1; grep -E "passed|failed|Error|assert" $O/pytest.log | tail -6 | cut -c1-220
CFG5="--config e2e_mask_rcnn_R_101_FPN_1x.yaml --dtype float16"
D="MODEL.RESNETS.STAGE_WITH_DCN (False,True,True,True)"
run() { timeout 600 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-kernel-timing "$@" < /dev/null > $O/$N.log 2>&1
  grep -E "^\{" $O/$N.log | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$N', d['value'], 'img/s', d['ms_per_step'], 'ms host', d['host_enqueue_ms_per_step'], 'loss_finite', d['loss_finite'], d.get('hip_graph'))" 2>/dev/null || tail -5 $O/$N.log; }
for rep in 1 2; do
export DETOPS_HALF_WEIGHTS=0
N=bf16_percast_$rep; run --dtype bfloat16
N=cfg5_percast_$rep; run $CFG5 $D
N=cfg5_graph_percast_$rep; run $CFG5 --hip-graph $D
export DETOPS_HALF_WEIGHTS=1
N=bf16_onecast_$rep; run --dtype bfloat16
N=cfg5_onecast_$rep; run $CFG5 $D
N=cfg5_graph_onecast_$rep; run $CFG5 --hip-graph $D
done
N=bf16_graph_onecast; run --dtype bfloat16 --hip-graph
N=retina_bf16_onecast; run --config retinanet/retinanet_R-50-FPN_1x.yaml --dtype bfloat16
DETOPS_HALF_WEIGHTS=0 N=retina_bf16_percast run --config retinanet/retinanet_R-50-FPN_1x.yaml --dtype bfloat16
P=/tmp/prof_hw; rm -rf $P
timeout 500 rocprofv3 --kernel-trace --output-format csv -d $P -o bench -- python bench.py --dtype bfloat16 --steps 6 --warmup 6 --no-cpu-baseline --no-kernel-timing < /dev/null > $O/trace_bf16.log 2>&1
T=$(find $P -name "*kernel_trace.csv" | head -1)
python tools/trace_steps.py "$T" 4 70 > $O/bf16_onecast_step_breakdown.txt 2>&1; head -3 $O/bf16_onecast_step_breakdown.txt | cut -c1-160
grep -E "copy_kernel|multi_tensor" $O/bf16_onecast_step_breakdown.txt | cut -c1-160
