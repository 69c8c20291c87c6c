# Round 6: ROIAlign on the ROI sets of a detector that has TRAINED for 300 steps on the bench workload (proposals cluster around the
# ground truth: heavier hit chains than at random init), and a sweep of the ring backward's split parameters on that set.
O=gpurun_out/r06rois; mkdir -p $O; export MIOPEN_LOG_LEVEL=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python tools/dump_model_rois.py --steps 320 --out $O/model_rois_step320.npz < /dev/null > $O/dump.log 2>&1; tail -3 $O/dump.log
OB="python tools/opbench.py --only roi_sets --sets model-random-init --model-rois $O/model_rois_step320.npz --layout both --iters 50"
timeout 300 $OB < /dev/null > $O/opbench_default.log 2>&1; grep roi_align $O/opbench_default.log | cut -c1-175
timeout 900 python tools/opbench.py --only roi_sets --sets model-random-init --model-rois $O/model_rois_step320.npz --layout nhwc --heads box --dir bwd --iters 30 \
   --sweep "roi_bwd_seg=16|24|32|48,roi_bwd_extras=128|384,roi_bwd_maxseg=8|16" < /dev/null > $O/opbench_sweep.log 2>&1; grep roi_align $O/opbench_sweep.log | cut -c1-200
