# Round 6 (late): mask head on the positives only (DETOPS_MASK_SLOTS=dynamic, the default) vs the fixed quota of 128 slots per image
O=gpurun_out/r06slots; mkdir -p $O; export MIOPEN_LOG_LEVEL=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_whole_model_parity.py -q -x -p no:cacheprovider < /dev/null > $O/pytest.log 2>&1; grep -E "passed|failed|Error|assert" $O/pytest.log | tail -4 | cut -c1-220
run() { timeout 600 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-kernel-timing "$@" < /dev/null > $O/$N.log 2>&1
  grep -E "^\{" $O/$N.log | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$N', d['value'], 'img/s', d['ms_per_step'], 'ms host', d['host_enqueue_ms_per_step'], 'loss_finite', d['loss_finite'], d.get('mask_slots'))" 2>/dev/null || tail -5 $O/$N.log; }
for rep in 1 2; do
DETOPS_MASK_SLOTS=fixed N=f32_fixed_$rep run
DETOPS_MASK_SLOTS=dynamic N=f32_dynamic_$rep run
done
for n in 32 64 96 128; do DETOPS_MASK_SLOTS=$n N=f32_forced_$n run; done
DETOPS_MASK_SLOTS=32,64 N=f32_forced_32_64 run
DETOPS_MASK_SLOTS=fixed N=bf16_fixed run --dtype bfloat16
DETOPS_MASK_SLOTS=dynamic N=bf16_dynamic run --dtype bfloat16
