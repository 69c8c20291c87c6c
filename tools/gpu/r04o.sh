cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -p no:cacheprovider -k "roi_align_forward" < /dev/null 2>&1 | tail -2
OB="python tools/opbench.py --only roi_sets --iters 60 --dir fwd --sets model-random-init,synthetic-loguniform,trained-like"
timeout 300 $OB --sweep "roi_fwd_ring=2|0|2|0" 2>&1 | grep roi_align | cut -c1-140
timeout 200 python tools/opbench.py --only roi_align_fwd --iters 50 2>&1 | grep "cfg1" | cut -c1-120
