# Round 6 (late): MIOpen perf-config SEARCH (MIOPEN_FIND_ENFORCE) for the tunable solvers of the fp32 headline step — the
# shipped find-db ranks solvers with each solver's default / heuristic kernel config; this tunes the configs themselves.
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
O=gpurun_out/r06tune; mkdir -p $O; export MIOPEN_LOG_LEVEL=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
jl() { grep -E "^\{" "$1" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$2', d['value'], 'img/s', d['ms_per_step'], 'ms', 'loss_finite', d['loss_finite'], d['miopen'])" 2>/dev/null || tail -3 "$1"; }
B="python bench.py --steps 40 --warmup 12 --no-cpu-baseline --no-kernel-timing"
timeout 300 $B < /dev/null > $O/shipped.log 2>&1; jl $O/shipped.log shipped; el shipped
DB=$GRAFT_REPO_ROOT/gpurun_out/r06tune/db
ENF=${ENF:-4}
MIOPEN_FIND_ENFORCE=$ENF timeout ${TUNE_TIMEOUT:-1800} python bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-kernel-timing --miopen-search --export-miopen-db $DB < /dev/null > $O/search.log 2>&1; jl $O/search.log search; el search
ls -la $DB/db | head; wc -l $DB/db/*.txt
(
export MIOPEN_USER_DB_PATH=$DB/db MIOPEN_CUSTOM_CACHE_DIR=$DB/cache
timeout 400 $B < /dev/null > $O/tuned.log 2>&1; jl $O/tuned.log tuned; el tuned
)
timeout 300 $B < /dev/null > $O/shipped_again.log 2>&1; jl $O/shipped_again.log shipped-again; el shipped-again
(
export MIOPEN_USER_DB_PATH=$DB/db MIOPEN_CUSTOM_CACHE_DIR=$DB/cache
timeout 400 $B < /dev/null > $O/tuned2.log 2>&1; jl $O/tuned2.log tuned-again; el tuned2
)
rm -rf $DB/cache/*.tmp; du -sh gpurun_out | tail -1
