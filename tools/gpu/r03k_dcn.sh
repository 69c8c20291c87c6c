# r03k: channels-last deformable-conv pipeline: parity (all DCN tests), block-level opbench, cfg-5 bench; shipped MIOpen db check
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
mkdir -p gpurun_out; export MIOPEN_LOG_LEVEL=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "deform or dcn or train_net" > gpurun_out/pytest_dcn.log 2>&1; tail -4 gpurun_out/pytest_dcn.log | cut -c1-300; el pytest-dcn
timeout 400 python tools/opbench.py --only dcn_block --iters 20 --json gpurun_out/opbench_dcn_block.json > gpurun_out/opbench_dcn_block.log 2>&1; grep -E "dcn|Error|error" gpurun_out/opbench_dcn_block.log | cut -c1-230; el opbench-dcn
timeout 400 python bench.py --config e2e_mask_rcnn_R_101_FPN_1x.yaml --dtype float16 --steps 30 --warmup 10 --no-cpu-baseline MODEL.RESNETS.STAGE_WITH_DCN "(False, True, True, True)" > gpurun_out/bench_cfg5.log 2>&1; grep -E "^\{" gpurun_out/bench_cfg5.log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('cfg5', d['value'], d['ms_per_step'], d['miopen'], d.get('roofline')); print(d.get('kernel_families_ms_per_step'))"; tail -2 gpurun_out/bench_cfg5.log | cut -c1-300; el bench-cfg5
timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline > gpurun_out/bench_f32.log 2>&1; grep -E "^\{" gpurun_out/bench_f32.log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('f32', d['value'], d['ms_per_step'], d['miopen'])"; el bench-f32
