# Round 6, experiment V: segmented NMS on input that is already in score order (the detector's case) — sort network skipped.
O=gpurun_out/r06v; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -q -x -k "nms or postprocess or eval_forward or rpn" -p no:cacheprovider < /dev/null > $O/pytest.log 2>&1; tail -2 $O/pytest.log
timeout 300 python tools/opbench.py --only nms --iters 200 < /dev/null > $O/opbench_nms.log 2>&1; grep "nms" $O/opbench_nms.log | cut -c1-170
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline"
timeout 400 $B < /dev/null > $O/bench.log 2>&1
grep -E "^\{" $O/bench.log | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], 'img/s', d['ms_per_step'], 'ms'); print(d.get('roofline')); [print(p) for p in d.get('roofline_path', [])]"
