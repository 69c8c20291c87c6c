T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
O=gpurun_out/r04j; mkdir -p $O; export MIOPEN_LOG_LEVEL=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --steps 40 --warmup 10 < /dev/null > $O/bench_f32.log 2>&1; grep -E "^\{" $O/bench_f32.log | tail -1 > $O/bench_f32.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04j/bench_f32.json'))
print(d['value'],'img/s',d['ms_per_step'],'ms host',d['host_enqueue_ms_per_step'], d.get('hip'), d.get('kernel_timers'))
print('roofline', d['roofline']['kernel'], d['roofline']['frac'])
for r in d.get('roofline_path',[]): print('  ', r['kernel'], r['mean_us'], r.get('frac'), 'timed', r['timed'])
print({k:v for k,v in d['kernel_families_ms_per_step'].items()})
PY
el bench
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider < /dev/null > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log | cut -c1-200; el pytest
