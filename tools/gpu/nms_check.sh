cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 240 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "nms" < /dev/null 2>&1 | grep -E "passed|failed|Error|assert" | tail -4
timeout 120 python tools/opbench.py --only nms --iters 200 < /dev/null 2>&1 | grep "nms" | cut -c1-130
