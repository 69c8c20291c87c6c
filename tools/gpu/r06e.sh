# Round 6, experiment E: ring backward with a channels-last store epilogue, fused bias (+ ReLU) + in-pass bias gradient for channels-last convolutions.
O=gpurun_out/r06g; mkdir -p $O; export MIOPEN_LOG_LEVEL=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
jl() { grep -E "^\{" "$1" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$2', d['value'], 'img/s', d['ms_per_step'], 'ms', 'loss_finite', d['loss_finite'], d.get('layout'), d['miopen']['db'])" 2>/dev/null || tail -3 "$1"; }
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "channels_last" -p no:cacheprovider < /dev/null > $O/pytest_nhwc.log 2>&1; tail -2 $O/pytest_nhwc.log
timeout 300 python tools/opbench.py --only roi_sets --layout nhwc --iters 30 < /dev/null > $O/opbench.log 2>&1; grep roi_align $O/opbench.log | cut -c1-150
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-timing"
timeout 400 $B --layout all < /dev/null > $O/all.log 2>&1; jl $O/all.log all
timeout 400 $B --layout backbone < /dev/null > $O/backbone.log 2>&1; jl $O/backbone.log backbone
timeout 400 $B --layout all < /dev/null > $O/all2.log 2>&1; jl $O/all2.log all-again
timeout 400 $B --layout nchw < /dev/null > $O/nchw.log 2>&1; jl $O/nchw.log nchw
P=/tmp/prof_all; rm -rf $P
timeout 500 rocprofv3 --kernel-trace --output-format csv -d $P -o bench -- python bench.py --steps 6 --warmup 6 --no-cpu-baseline --no-kernel-timing --layout all < /dev/null > $O/prof_all.log 2>&1
T=$(find $P -name "*kernel_trace.csv" | head -1)
[ -n "$T" ] && python tools/trace_steps.py "$T" 4 70 > $O/all_step_breakdown.txt 2>&1 && head -50 $O/all_step_breakdown.txt | cut -c1-150
PM="python tools/opbench.py --only roi_sets --layout nhwc --heads box --dir fwd --iters 5 --sets model-random-init"
for pass in "sq:SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
            "sq2:SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  n=${pass%%:*}; c=${pass#*:}; rm -rf /tmp/pmc_$n
  timeout 150 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$n -o x -- $PM < /dev/null > $O/pmc_$n.log 2>&1
done
python tools/pmc_diag.py /tmp/pmc_sq /tmp/pmc_sq2 > $O/roi_align_nhwc_bwd_pmc.txt 2>&1; grep -v "roi_order\|order_kernel" $O/roi_align_nhwc_bwd_pmc.txt | head -60
