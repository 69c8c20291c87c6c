# r01e-a: variants (ROI-list split, col2im XCD remap / smooth offsets, fwd per-shape defaults)
set -x
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 200 python -m pytest tests -m gpu -q -p no:cacheprovider -k "roi_align or deform or pooler" > gpurun_out/pytest_gpu_c.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_c.log
tail -4 gpurun_out/pytest_gpu_c.log | cut -c1-220; el pytest
timeout 120 python tools/opbench.py --iters 20 --only roi_align,dcn --json gpurun_out/opbench_c.json > gpurun_out/opbench_c.log 2>&1
grep -v "^/opt" gpurun_out/opbench_c.log | cut -c1-200 | grep "cfg1\|dcn_col2im\|mask-head 256x14x14 *[0-9]\|box-head 1024x7x7 *[0-9]\|gather (atomic\|total"; el opbench
