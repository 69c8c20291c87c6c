# Round 6, experiment A: the backbone + FPN on channels-last activations WITH a tuned MIOpen find-db for the NHWC problem keys
# (the r05 probe ran immediate-mode picks only).   gpurun -- 'bash tools/gpu/r06a_nhwc.sh'
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
O=gpurun_out/r06a; mkdir -p $O; export MIOPEN_LOG_LEVEL=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
jl() { grep -E "^\{" "$1" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$2', d['value'], 'img/s', d['ms_per_step'], 'ms', 'loss_finite', d['loss_finite'], d['miopen'])" 2>/dev/null || tail -3 "$1"; }
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-timing"
timeout 300 $B < /dev/null > $O/nchw.log 2>&1; jl $O/nchw.log nchw; el nchw
timeout 400 $B --channels-last < /dev/null > $O/nhwc_immediate.log 2>&1; jl $O/nhwc_immediate.log nhwc-immediate; el nhwc-immediate
DB=$GRAFT_REPO_ROOT/gpurun_out/r06a/db_nhwc
timeout 1500 python bench.py --steps 4 --warmup 4 --no-cpu-baseline --no-kernel-timing --channels-last --miopen-search --export-miopen-db $DB < /dev/null > $O/nhwc_search.log 2>&1; jl $O/nhwc_search.log nhwc-search; el nhwc-search
ls -la $DB/db | head; wc -l $DB/db/*.txt
export MIOPEN_USER_DB_PATH=$DB/db MIOPEN_CUSTOM_CACHE_DIR=$DB/cache
timeout 400 $B --channels-last < /dev/null > $O/nhwc_tuned.log 2>&1; jl $O/nhwc_tuned.log nhwc-tuned; el nhwc-tuned
timeout 400 $B < /dev/null > $O/nchw_again.log 2>&1; jl $O/nchw_again.log nchw-again; el nchw-again
P=/tmp/prof_nhwc; rm -rf $P
timeout 500 rocprofv3 --kernel-trace --output-format csv -d $P -o bench -- python bench.py --steps 6 --warmup 6 --no-cpu-baseline --no-kernel-timing --channels-last < /dev/null > $O/prof_nhwc.log 2>&1
T=$(find $P -name "*kernel_trace.csv" | head -1)
[ -n "$T" ] && python tools/trace_steps.py "$T" 4 70 > $O/nhwc_tuned_step_breakdown.txt 2>&1 && head -50 $O/nhwc_tuned_step_breakdown.txt | cut -c1-160
el trace
rm -rf $DB/cache/*.tmp; du -sh gpurun_out | tail -1
