# r04b: ROIAlign forward records / static staging, backward 32-channel units: parity first, then the A/B sweep
#   gpurun -- 'bash tools/gpu/r04b_roi_sweep.sh'
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
O=gpurun_out/r04b; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -p no:cacheprovider -k "roi_align" < /dev/null > $O/pytest_roi.log 2>&1; echo "rc=$?" >> $O/pytest_roi.log
tail -4 $O/pytest_roi.log | cut -c1-200; el pytest
OB="python tools/opbench.py --only roi_sets --iters 40"
timeout 300 $OB --sets model-random-init,synthetic-loguniform --dir fwd --sweep "roi_fwd_records=1|2|0" < /dev/null > $O/fwd_modes.log 2>&1; grep roi_align $O/fwd_modes.log | cut -c1-170; el fwd
timeout 300 $OB --sets model-random-init --dir fwd --heads box --sweep "roi_fwd_ct=16|32|64" < /dev/null > $O/fwd_ct.log 2>&1; grep roi_align $O/fwd_ct.log | cut -c1-170; el fwd-ct
timeout 400 $OB --sets model-random-init,synthetic-loguniform --dir bwd --heads box --sweep "roi_bwd_ct=0|32,roi_bwd_ring=2|3" < /dev/null > $O/bwd_ct.log 2>&1; grep roi_align $O/bwd_ct.log | cut -c1-170; el bwd-ct
timeout 300 $OB --sets model-random-init --dir bwd --heads box --sweep "roi_bwd_ct=32,roi_bwd_seg=16|24|48|64" < /dev/null > $O/bwd_seg.log 2>&1; grep roi_align $O/bwd_seg.log | cut -c1-170; el bwd-seg
timeout 300 $OB --sets model-random-init --images 4 --iters 20 < /dev/null > $O/l3_variant.log 2>&1; grep roi_align $O/l3_variant.log | cut -c1-170; el l3
timeout 200 python tools/opbench.py --only roi_align --iters 30 < /dev/null > $O/roi_align_all.log 2>&1; grep "cfg1" $O/roi_align_all.log | cut -c1-170; el cfg1
