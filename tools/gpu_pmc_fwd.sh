# PMC diagnosis of the FPN-fused ROIAlign launches, forward focus (separate passes: SQ / SQ2 / TCP / TCC / FETCH / WRITE)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python tools/opbench.py --iters 3 --only roi_align_fpn"
for d in pmc_sq pmc_sq2 pmc_tcp pmc_tcc pmc_fetch pmc_write; do rm -rf gpurun_out/$d; done
timeout 150 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d gpurun_out/pmc_sq -o x -- $CMD > gpurun_out/pmc_sq.log 2>&1
timeout 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc_sq2 -o x -- $CMD > gpurun_out/pmc_sq2.log 2>&1
timeout 150 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum --kernel-trace --output-format csv -d gpurun_out/pmc_tcp -o x -- $CMD > gpurun_out/pmc_tcp.log 2>&1
timeout 150 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d gpurun_out/pmc_tcc -o x -- $CMD > gpurun_out/pmc_tcc.log 2>&1
timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o x -- $CMD > gpurun_out/pmc_fetch.log 2>&1
timeout 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o x -- $CMD > gpurun_out/pmc_write.log 2>&1
python tools/pmc_diag.py gpurun_out/pmc_sq gpurun_out/pmc_sq2 gpurun_out/pmc_tcp gpurun_out/pmc_tcc gpurun_out/pmc_fetch gpurun_out/pmc_write > gpurun_out/pmc_diag.txt 2>&1
find gpurun_out/pmc_* -name "*kernel_trace.csv" -delete
grep -A34 "roi_align_fwd" gpurun_out/pmc_diag.txt | head -90
