# Round-2 opener: measure the opt-in variants that round 1 could only check in the host emulation
# (GPU minutes ran out).  ~1 minute on the box.
#   DETOPS_ROIALIGN_BWD_WALK=lane   per-lane bin-range walk in the pixel-owner backward
#   DETOPS_ROIALIGN_BWD=gather3     chunk-group backward (G x 16 channels, one staging phase per batch)
#   DETOPS_ROIALIGN_FWD_ORDER=1     spatially ordered ROI visiting + XCD-contiguous ids in the forward
#   DETOPS_DCN_COL2IM=ell           fixed-width inverted index for deformable col2im
set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
DETOPS_TEST_EXPERIMENTAL=1 timeout 200 python -m pytest tests/test_experimental_gpu.py -m gpu -q -p no:cacheprovider > gpurun_out/pytest_experimental.log 2>&1; tail -3 gpurun_out/pytest_experimental.log
timeout 180 python tools/opbench.py --iters 30 --only roi_align,dcn --experimental --json gpurun_out/opbench_experimental.json > gpurun_out/opbench_experimental.log 2>&1
grep -v "^/opt" gpurun_out/opbench_experimental.log | grep "fpn-fused\|dcn_col2im" | cut -c1-220
timeout 90 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch_x -o x -- env DETOPS_ROIALIGN_FWD_ORDER=1 python tools/opbench.py --iters 3 --only roi_align_fpn > gpurun_out/pmc_fetch_x.log 2>&1
python tools/pmc_traffic.py gpurun_out/pmc_fetch_x gpurun_out/pmc_fetch_x gpurun_out/traffic_ordered.json 2>&1 | grep fwd | cut -c1-200
find gpurun_out/pmc_fetch_x -name "*kernel_trace.csv" -delete
# ---- where do the slow kernels spend their cycles?  (separate PMC passes: SQ / TCP+TA / TCC)
CMD="python tools/opbench.py --iters 3 --only roi_align_fpn,dcn"
timeout 150 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d gpurun_out/pmc_sq -o x -- $CMD > gpurun_out/pmc_sq.log 2>&1
timeout 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc_sq2 -o x -- $CMD > gpurun_out/pmc_sq2.log 2>&1
timeout 150 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum --kernel-trace --output-format csv -d gpurun_out/pmc_tcp -o x -- $CMD > gpurun_out/pmc_tcp.log 2>&1
timeout 150 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d gpurun_out/pmc_tcc -o x -- $CMD > gpurun_out/pmc_tcc.log 2>&1
python tools/pmc_diag.py gpurun_out/pmc_sq gpurun_out/pmc_sq2 gpurun_out/pmc_tcp gpurun_out/pmc_tcc > gpurun_out/pmc_diag.txt 2>&1
find gpurun_out/pmc_sq gpurun_out/pmc_sq2 gpurun_out/pmc_tcp gpurun_out/pmc_tcc -name "*kernel_trace.csv" -delete
head -80 gpurun_out/pmc_diag.txt
