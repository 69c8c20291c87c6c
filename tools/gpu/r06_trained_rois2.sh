O=gpurun_out/r06rois; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for set in tools/gpu/model_rois_step320.npz tests/golden/model_rois.npz; do
echo "== $set"
timeout 900 python tools/opbench.py --only roi_sets --sets model-random-init --model-rois $set --layout nhwc --dir bwd --iters 30 \
   --sweep "roi_bwd_seg=32|48|64|96|128|1000" < /dev/null > $O/sweep2.log 2>&1; grep roi_align $O/sweep2.log | cut -c1-150
done
python tools/gpu/ring_timeline.py model-random-init 2>&1 | head -12
