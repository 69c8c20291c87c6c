# ring backward: what does a hit cost?  ablation bits of tuning roi_bwd_debug: 1 = no walk, 2 = no LDS-DMA issue, 3 = neither
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; O=gpurun_out/r05d; mkdir -p $O
timeout 200 python tools/opbench.py --only roi_sets --dir bwd --iters 50 --sets model-random-init,synthetic-loguniform --sweep "roi_bwd_debug=0|1|2|3" < /dev/null > $O/ablate.log 2>&1
grep -E "roi_align|roi_bwd_debug|sweep" $O/ablate.log | cut -c1-170
