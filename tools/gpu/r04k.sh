cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 200 python tools/opbench.py --only targets --iters 50 2>&1 | grep "match_boxes\|sample_labels" | cut -c1-150
rm -rf /tmp/kt; timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o x -- python tools/opbench.py --only targets --iters 20 > /dev/null 2>&1
python tools/kernel_times.py /tmp/kt "" | grep -v "at::\|elementwise" | head -12 | cut -c1-160
timeout 900 python -m pytest tests/test_targets_gpu.py tests/test_whole_model_parity.py tests/test_ops_gpu.py -q -p no:cacheprovider -k "targets or whole or nms or sampler or match" < /dev/null 2>&1 | tail -3
