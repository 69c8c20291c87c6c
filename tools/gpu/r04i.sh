cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OB="python tools/opbench.py --only roi_sets --iters 40 --dir bwd --heads box --sets model-random-init,trained-like"
timeout 300 $OB --sweep "roi_bwd_split=0" 2>&1 | grep roi_align | cut -c1-140
timeout 300 $OB --tune roi_bwd_extras=512,roi_bwd_split=1 --sweep "roi_bwd_seg=32|24|16,roi_bwd_maxseg=8|16" 2>&1 | grep roi_align | cut -c1-160
timeout 300 $OB --tune roi_bwd_extras=512 --sweep "roi_bwd_seg=24|16,roi_bwd_maxseg=8|16" 2>&1 | grep roi_align | cut -c1-160
