# Round 6, experiment B: ROIAlign over a channels-last pyramid (csrc/roi_align_nhwc.hip) — parity on the device, opbench
# NCHW vs NHWC on the model's ROI set, then the whole step with the heads channels-last (find-db search for the new NHWC keys).
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
O=gpurun_out/r06b; mkdir -p $O; export MIOPEN_LOG_LEVEL=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
jl() { grep -E "^\{" "$1" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$2', d['value'], 'img/s', d['ms_per_step'], 'ms', 'loss_finite', d['loss_finite'], d.get('layout'), d['miopen'])" 2>/dev/null || tail -3 "$1"; }
timeout 900 python -m pytest tests/test_ops_gpu.py -q -x -k "channels_last" -p no:cacheprovider < /dev/null > $O/pytest_nhwc.log 2>&1; tail -3 $O/pytest_nhwc.log; el pytest
timeout 600 python tools/opbench.py --only roi_sets --layout both --iters 30 --json $O/opbench_roi.json < /dev/null > $O/opbench_roi.log 2>&1; grep roi_align $O/opbench_roi.log | cut -c1-160; el opbench
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-timing"
DB=$GRAFT_REPO_ROOT/gpurun_out/r06b/db_all
timeout 1500 python bench.py --steps 4 --warmup 4 --no-cpu-baseline --no-kernel-timing --layout all --miopen-search --export-miopen-db $DB < /dev/null > $O/all_search.log 2>&1; jl $O/all_search.log all-search; el all-search
wc -l $DB/db/*.txt
export MIOPEN_USER_DB_PATH=$DB/db MIOPEN_CUSTOM_CACHE_DIR=$DB/cache
timeout 400 $B --layout all < /dev/null > $O/all_tuned.log 2>&1; jl $O/all_tuned.log all-tuned; el all-tuned
timeout 400 $B --layout backbone < /dev/null > $O/backbone.log 2>&1; jl $O/backbone.log backbone; el backbone
timeout 400 $B --layout nchw < /dev/null > $O/nchw.log 2>&1; jl $O/nchw.log nchw; el nchw
timeout 400 $B --layout all < /dev/null > $O/all_tuned2.log 2>&1; jl $O/all_tuned2.log all-tuned-again; el all-tuned2
P=/tmp/prof_all; rm -rf $P
timeout 500 rocprofv3 --kernel-trace --output-format csv -d $P -o bench -- python bench.py --steps 6 --warmup 6 --no-cpu-baseline --no-kernel-timing --layout all < /dev/null > $O/prof_all.log 2>&1
T=$(find $P -name "*kernel_trace.csv" | head -1)
[ -n "$T" ] && python tools/trace_steps.py "$T" 4 70 > $O/all_step_breakdown.txt 2>&1 && head -60 $O/all_step_breakdown.txt | cut -c1-160
el trace
du -sh gpurun_out | tail -1
