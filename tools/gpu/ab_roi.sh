# same-box A/B of the ROIAlign kernels: base = tools/gpu/ab/libdetops_base.so (tools/gpu/ab_build.sh), new = the tree's library
#   gpurun --timeout 300 -- 'bash tools/gpu/ab_roi.sh [fwd,bwd] [reps]'
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; O=gpurun_out/ab_roi; mkdir -p $O
D=${1:-bwd}; R=${2:-2}
timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -p no:cacheprovider -k "roi_align" < /dev/null > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -1
for rep in $(seq 1 $R); do for v in base new; do
  L=""; [ $v = base ] && L=$GRAFT_REPO_ROOT/tools/gpu/ab/libdetops_base.so
  DETOPS_LIB_PATH=$L timeout 120 python tools/opbench.py --only roi_sets --dir $D --iters 50 < /dev/null > $O/${v}_$rep.log 2>&1
  echo "== $v $rep"; grep -E "roi_align" $O/${v}_$rep.log | cut -c1-160
done; done
