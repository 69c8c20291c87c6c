# Round 6, experiment X: deformable-conv input gradient as a col2im GATHER of the column gradient (no S_T, no second GEMM)
# against the transposed sampling + GEMM of rounds 3-5 (DETOPS_DCN_INPUT_GRAD=col2im|transposed); one box.
O=gpurun_out/r06x; mkdir -p $O; export MIOPEN_LOG_LEVEL=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -q -x -k "deform or dcn or dconv" -p no:cacheprovider < /dev/null > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -2
CFG5="--config e2e_mask_rcnn_R_101_FPN_1x.yaml --dtype float16 MODEL.RESNETS.STAGE_WITH_DCN (False,True,True,True)"
for form in transposed col2im transposed col2im; do
  export DETOPS_DCN_INPUT_GRAD=$form
  timeout 400 python bench.py --steps 30 --warmup 8 --no-cpu-baseline $CFG5 < /dev/null > $O/cfg5_$form.log 2>&1
  grep -E "^\{" $O/cfg5_$form.log | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$form', d['value'], 'img/s', d['ms_per_step'], 'ms host', d['host_enqueue_ms_per_step'], {k: v for k, v in d['kernel_families_ms_per_step'].items() if k.startswith('dcn')})
[print('   ', k, v['mean_us'], v.get('achieved_GBs')) for k, v in d['kernels'].items() if k.startswith('dcn_col2im_nhwc') or k.startswith('dcn_transposed')]"
done
unset DETOPS_DCN_INPUT_GRAD
timeout 400 python tools/opbench.py --only dcn_block --iters 20 < /dev/null > $O/opbench_dcn.log 2>&1; grep -E "dcn_block" $O/opbench_dcn.log | cut -c1-170 | head -20
