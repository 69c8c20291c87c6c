# r03c: where does the ring kernel spend its time?  ablations, ring depth, PMC on the model ROI set (box head)
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OB="python tools/opbench.py --only roi_sets --heads box --dir bwd --iters 40"
for T in "" "roi_bwd_ring=2" "roi_bwd_ring=4" "roi_bwd_debug=1" "roi_bwd_debug=2" "roi_bwd_debug=3" "roi_bwd_seg=1000000,roi_bwd_debug=3"; do
  DETOPS_TUNING="$T" timeout 100 $OB 2>&1 | grep -E "roi_align_bwd" | sed "s/^/[$T] /" | cut -c1-140
done; el ablations
PM="python tools/opbench.py --only roi_sets --heads box --dir bwd --iters 5 --sets model-random-init"
timeout 120 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pmc_kt -o x -- $PM > gpurun_out/pmc_kt.log 2>&1
python tools/kernel_times.py gpurun_out/pmc_kt roi_ ; el trace
timeout 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/pmc_sq -o x -- $PM > gpurun_out/pmc_sq.log 2>&1
timeout 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/pmc_sq2 -o x -- $PM > gpurun_out/pmc_sq2.log 2>&1
timeout 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d gpurun_out/pmc_tcc -o x -- $PM > gpurun_out/pmc_tcc.log 2>&1
timeout 120 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d gpurun_out/pmc_fw -o x -- $PM > gpurun_out/pmc_fw.log 2>&1
python tools/pmc_diag.py gpurun_out/pmc_sq gpurun_out/pmc_sq2 gpurun_out/pmc_tcc gpurun_out/pmc_fw > gpurun_out/pmc_diag.txt 2>&1; cat gpurun_out/pmc_diag.txt | grep -v "roi_order\|fwd" | head -80; el pmc
