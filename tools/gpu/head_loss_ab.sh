# First GPU call of the next round:  gpurun --timeout 240 -- 'bash tools/gpu/head_loss_ab.sh'
# The opt-in ROI-head loss kernels (csrc/head_loss.hip, DETOPS_HEAD_LOSS=fused) have only run under the host emulation.
# 1. their device parity test; 2. same-box A/B of the step with and without them (fp32 and bf16, interleaved).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; O=gpurun_out/head_loss; mkdir -p $O; export MIOPEN_LOG_LEVEL=1
DETOPS_TEST_UNMEASURED=1 timeout 120 python -m pytest tests/test_targets_gpu.py -m gpu -q -p no:cacheprovider -k fused_head_losses < /dev/null > $O/pytest.log 2>&1
tail -3 $O/pytest.log | cut -c1-200
B="python bench.py --steps 40 --warmup 12 --no-cpu-baseline"
for dt in float32 bfloat16; do for v in torch fused torch fused; do
  DETOPS_HEAD_LOSS=$v timeout 90 $B --dtype $dt < /dev/null > $O/${dt}_$v.log 2>&1
  grep -E "^\{" $O/${dt}_$v.log | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$dt $v', d['value'], 'img/s', d['ms_per_step'], 'ms host', d.get('host_enqueue_ms_per_step'), {k: v['mean_us'] for k, v in d['kernels'].items() if 'loss' in k})" || tail -3 $O/${dt}_$v.log
done; done
