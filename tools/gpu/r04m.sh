cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OB="python tools/opbench.py --only roi_sets --iters 40 --dir bwd --sets model-random-init,synthetic-loguniform,trained-like"
timeout 300 $OB --sweep "roi_bwd_ct=0|16" 2>&1 | grep roi_align | cut -c1-140
timeout 200 python tools/gpu/ring_timeline.py model-random-init 2>&1 | grep "units that\|hits "
