# r03d: ring kernel v2 (scalar-base DMA, packed FMA, straight-line walk)
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -p no:cacheprovider -k "roi_align" > gpurun_out/pytest_roi.log 2>&1; tail -3 gpurun_out/pytest_roi.log | cut -c1-300; el pytest-roi
OB="python tools/opbench.py --only roi_sets --dir bwd --iters 40"
for T in "" "roi_bwd_ring=2" "roi_bwd_ring=4" "roi_bwd_debug=1" "roi_bwd_debug=3"; do
  DETOPS_TUNING="$T" timeout 100 $OB 2>&1 | grep -E "roi_align_bwd" | sed "s/^/[$T] /" | cut -c1-150
done; el ablations
PM="python tools/opbench.py --only roi_sets --heads box --dir bwd --iters 5 --sets model-random-init"
timeout 120 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pmc_kt -o x -- $PM > gpurun_out/pmc_kt.log 2>&1
python tools/kernel_times.py gpurun_out/pmc_kt roi_ ; el trace
timeout 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmc_sq2 -o x -- $PM > gpurun_out/pmc_sq2.log 2>&1
timeout 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/pmc_sq -o x -- $PM > gpurun_out/pmc_sq.log 2>&1
python tools/pmc_diag.py gpurun_out/pmc_sq gpurun_out/pmc_sq2 > gpurun_out/pmc_diag.txt 2>&1; grep -A30 "ring_kernel" gpurun_out/pmc_diag.txt | head -40; el pmc
