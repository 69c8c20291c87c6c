T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 240 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "nms" < /dev/null 2>&1 | tail -6; el pytest
timeout 120 python tools/opbench.py --only nms --iters 200 < /dev/null 2>&1 | grep -v "^RCCL\|Warn" | tail -9; el opbench
