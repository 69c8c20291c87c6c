# r04c: acc backward (cfg-1) parity + timing; forward after the barrier fix
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
O=gpurun_out/r04c; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py -q -p no:cacheprovider -k "roi_align" < /dev/null > $O/pytest_roi.log 2>&1; echo "rc=$?" >> $O/pytest_roi.log
tail -5 $O/pytest_roi.log | cut -c1-200; el pytest
timeout 200 python tools/opbench.py --only roi_align --iters 50 < /dev/null > $O/roi_align_all.log 2>&1; grep "cfg1\|fpn-fused" $O/roi_align_all.log | cut -c1-170; el cfg1
for g in 8 16 32 64; do timeout 100 python tools/opbench.py --only roi_align --iters 50 --tune roi_bwd_groups=$g < /dev/null 2>&1 | grep "cfg1.*split over" | cut -c1-150 | sed "s/^/groups=$g /"; done; el groups
OB="python tools/opbench.py --only roi_sets --iters 60"
timeout 300 $OB --sets model-random-init,synthetic-loguniform --dir fwd --sweep "roi_fwd_records=1|2|0|1|2|0" < /dev/null > $O/fwd_modes.log 2>&1; grep roi_align $O/fwd_modes.log | cut -c1-150; el fwd
