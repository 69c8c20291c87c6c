cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OB="python tools/opbench.py --only roi_sets --iters 40 --dir bwd --heads box --sets model-random-init,trained-like,synthetic-loguniform"
timeout 300 $OB --tune roi_bwd_extras=256,roi_bwd_split=1 --sweep "roi_bwd_seg=36|40|44|48|56" 2>&1 | grep roi_align | cut -c1-140
timeout 300 $OB --sweep "roi_bwd_seg=0|28|36|40" 2>&1 | grep roi_align | cut -c1-140
