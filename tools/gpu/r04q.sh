cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for t in "roi_bwd_split=0" "roi_bwd_split=1,roi_bwd_seg=44,roi_bwd_extras=256" "roi_bwd_split=1,roi_bwd_seg=36,roi_bwd_extras=256" "roi_bwd_ct=16,roi_bwd_split=1,roi_bwd_seg=44,roi_bwd_extras=256"; do
echo "=== $t"; timeout 200 python tools/gpu/ring_timeline.py model-random-init $t 2>&1 | grep -v amdgpu.ids | head -9
timeout 200 python tools/gpu/ring_timeline.py model-random-init $t 2>&1 | grep "t  " | awk 'NR%2==1' | head -12
done
