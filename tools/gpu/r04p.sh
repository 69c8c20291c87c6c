cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_ops_gpu.py -q -x -p no:cacheprovider -k "pooler or roi_align_backward or whole" < /dev/null 2>&1 | tail -3
export MIOPEN_LOG_LEVEL=1
for st in 20 100; do
timeout 600 python bench.py --steps $st --warmup 10 < /dev/null > gpurun_out/bench_s$st.log 2>&1; grep -E "^\{" gpurun_out/bench_s$st.log | tail -1 > gpurun_out/bench_s$st.json
python - <<PY
import json
d=json.load(open('gpurun_out/bench_s$st.json'))
ks=d['kernels']
print('steps',d['steps'],d['value'],'img/s',d['ms_per_step'],'ms host',d['host_enqueue_ms_per_step'],'samples/step %.1f'%(sum(v['timed'] for v in ks.values())/d['steps']),'min timed',min(v['timed'] for v in ks.values()))
for r in d.get('roofline_path',[]): print('  ', r['kernel'], r['mean_us'], r.get('frac'), 'timed', r['timed'])
print('  prepare:', {k:(v['mean_us'],v['timed']) for k,v in ks.items() if 'prepare' in k})
PY
done
timeout 300 python bench.py --steps 40 --warmup 10 --no-kernel-timing --no-cpu-baseline < /dev/null 2>&1 | grep -E "^\{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('no timers', d['value'], d['ms_per_step'])"
