cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OB="python tools/opbench.py --only roi_sets --iters 40 --dir bwd --heads box --sets model-random-init,trained-like,synthetic-loguniform"
timeout 300 $OB 2>&1 | grep roi_align | cut -c1-140
timeout 300 $OB 2>&1 | grep roi_align | cut -c1-140
