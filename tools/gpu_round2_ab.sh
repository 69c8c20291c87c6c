# Round-2 opener: measure the opt-in variants that round 1 could only check in the host emulation
# (GPU minutes ran out).  ~1 minute on the box.
#   DETOPS_ROIALIGN_BWD_WALK=lane   per-lane bin-range walk in the pixel-owner backward
#   DETOPS_ROIALIGN_FWD_ORDER=1     spatially ordered ROI visiting + XCD-contiguous ids in the forward
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 120 python tools/opbench.py --iters 30 --only roi_align --experimental --json gpurun_out/opbench_experimental.json > gpurun_out/opbench_experimental.log 2>&1
grep -v "^/opt" gpurun_out/opbench_experimental.log | grep "fpn-fused" | cut -c1-220
timeout 90 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch_x -o x -- env DETOPS_ROIALIGN_FWD_ORDER=1 python tools/opbench.py --iters 3 --only roi_align_fpn > gpurun_out/pmc_fetch_x.log 2>&1
python tools/pmc_traffic.py gpurun_out/pmc_fetch_x gpurun_out/pmc_fetch_x gpurun_out/traffic_ordered.json 2>&1 | grep fwd | cut -c1-200
find gpurun_out/pmc_fetch_x -name "*kernel_trace.csv" -delete
