O=gpurun_out/r06dcnhost; mkdir -p $O; export MIOPEN_LOG_LEVEL=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -q -x -k "deform or dcn or dconv" -p no:cacheprovider < /dev/null > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2
CFG5="--config e2e_mask_rcnn_R_101_FPN_1x.yaml --dtype float16 MODEL.RESNETS.STAGE_WITH_DCN (False,True,True,True)"
for rep in 1 2 3; do
  timeout 400 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-timing $CFG5 < /dev/null > $O/cfg5_$rep.log 2>&1
  grep -E "^\{" $O/cfg5_$rep.log | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cfg5', d['value'], 'img/s', d['ms_per_step'], 'ms host', d['host_enqueue_ms_per_step'])"
done
python tools/host_profile.py --config e2e_mask_rcnn_R_101_FPN_1x.yaml --dtype float16 --steps 10 MODEL.RESNETS.STAGE_WITH_DCN "(False,True,True,True)" 2>&1 | grep -E "per step|DeformConv"
