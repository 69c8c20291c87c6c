cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -q -x -p no:cacheprovider -k "roi_align_backward or adjoint or pooler" < /dev/null 2>&1 | tail -2
OB="python tools/opbench.py --only roi_sets --iters 40 --dir bwd --sets model-random-init,trained-like,synthetic-loguniform"
timeout 300 $OB 2>&1 | grep roi_align | cut -c1-140
timeout 200 python tools/gpu/ring_timeline.py model-random-init 2>&1 | grep "units that\|heavy units\|hits "
timeout 200 python tools/gpu/cfg1_bwd.py 0 50 2>&1 | grep cfg1
