cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python tools/opbench.py --only dcn_block --iters 20 2>&1 | grep "dcn_block" | cut -c1-150
rm -rf /tmp/kt; timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o x -- python tools/opbench.py --only dcn_block --iters 10 > /dev/null 2>&1
python tools/kernel_times.py /tmp/kt "" | grep -v "Cijk\|at::\|rocclr\|elementwise" | head -24 | cut -c1-160
timeout 900 python -m pytest tests/test_ops_gpu.py -q -p no:cacheprovider -k "deform or dcn or psroi" < /dev/null 2>&1 | tail -3
