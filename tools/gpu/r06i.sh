# Round 6, experiment I: record-form ("gather") NHWC backward vs the ring kernel with a channels-last store.
O=gpurun_out/r06l; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
DETOPS_ROIALIGN_NHWC_BWD=native timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "channels_last" -p no:cacheprovider < /dev/null > $O/pytest_native.log 2>&1; tail -2 $O/pytest_native.log
for mode in native ring; do
  DETOPS_ROIALIGN_NHWC_BWD=$mode timeout 300 python tools/opbench.py --only roi_sets --layout nhwc --dir bwd --iters 30 < /dev/null > $O/opbench_$mode.log 2>&1; echo "== $mode"; grep roi_align $O/opbench_$mode.log | cut -c1-150
done
rm -rf /tmp/kt; DETOPS_ROIALIGN_NHWC_BWD=native timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o x -- python tools/opbench.py --only roi_sets --layout nhwc --dir bwd --iters 10 --sets model-random-init < /dev/null > $O/kt.log 2>&1
python tools/kernel_times.py /tmp/kt "" 2>/dev/null | head -12 | cut -c1-160
PM="python tools/opbench.py --only roi_sets --layout nhwc --heads box --dir bwd --iters 5 --sets model-random-init"
for pass in "sq:SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
            "sq2:SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
            "tcc:TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"; do
  n=${pass%%:*}; c=${pass#*:}; rm -rf /tmp/pmc_$n
  DETOPS_ROIALIGN_NHWC_BWD=native timeout 150 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$n -o x -- $PM < /dev/null > $O/pmc_$n.log 2>&1
done
python tools/pmc_diag.py /tmp/pmc_sq /tmp/pmc_sq2 /tmp/pmc_tcc > $O/ng_pmc.txt 2>&1; grep -A30 "bwd_ng_kernel" $O/ng_pmc.txt | head -40
