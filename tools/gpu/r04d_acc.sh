T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
O=gpurun_out/r04d; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 200 python tools/gpu/cfg1_bwd.py 0,8,16,32,64 50 2>&1 | grep cfg1; el sweep
rm -rf /tmp/kt; timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o x -- python tools/gpu/cfg1_bwd.py 0 20 > $O/trace.log 2>&1
python tools/kernel_times.py /tmp/kt "" | grep -v "at::\|elementwise\|fill" | head -12 | cut -c1-160; el trace
timeout 600 python -m pytest tests/test_ops_gpu.py -q -p no:cacheprovider -k "roi_align_backward" < /dev/null 2>&1 | tail -2
