# Round 6, experiment P: channels-last for the half-precision configurations (bf16 autocast; cfg-5 = R-101 + DCN, fp16):
# find-db search for the NHWC half keys, then NCHW vs channels-last with the tuned database.
O=gpurun_out/r06p; mkdir -p $O; export MIOPEN_LOG_LEVEL=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
jl() { grep -E "^\{" "$1" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-28s' % '$2', d['value'], 'img/s', d['ms_per_step'], 'ms', 'loss_finite', d['loss_finite'], d.get('layout'), d['miopen']['db'])" 2>/dev/null || tail -3 "$1"; }
DB=$GRAFT_REPO_ROOT/gpurun_out/r06p/db_half
CFG5="--config e2e_mask_rcnn_R_101_FPN_1x.yaml --dtype float16 MODEL.RESNETS.STAGE_WITH_DCN (False,True,True,True)"
S="python bench.py --steps 4 --warmup 4 --no-cpu-baseline --no-kernel-timing --miopen-search --export-miopen-db $DB"
timeout 1500 $S --dtype bfloat16 --layout all < /dev/null > $O/search_bf16.log 2>&1; jl $O/search_bf16.log search-bf16-all
timeout 1500 $S --layout backbone $CFG5 < /dev/null > $O/search_cfg5.log 2>&1; jl $O/search_cfg5.log search-cfg5-backbone
timeout 1500 $S --layout all $CFG5 < /dev/null > $O/search_cfg5_all.log 2>&1; jl $O/search_cfg5_all.log search-cfg5-all
wc -l $DB/db/*.txt | tail -1
export MIOPEN_USER_DB_PATH=$DB/db MIOPEN_CUSTOM_CACHE_DIR=$DB/cache
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-timing"
for lay in nchw backbone all; do
  timeout 400 $B --dtype bfloat16 --layout $lay < /dev/null > $O/bf16_$lay.log 2>&1; jl $O/bf16_$lay.log bf16-$lay
done
for lay in nchw backbone all; do
  timeout 400 $B --layout $lay $CFG5 < /dev/null > $O/cfg5_$lay.log 2>&1; jl $O/cfg5_$lay.log cfg5-$lay
done
P=/tmp/prof_c5; rm -rf $P
timeout 500 rocprofv3 --kernel-trace --output-format csv -d $P -o bench -- python bench.py --steps 6 --warmup 6 --no-cpu-baseline --no-kernel-timing --layout all $CFG5 < /dev/null > $O/prof_c5.log 2>&1
T=$(find $P -name "*kernel_trace.csv" | head -1)
[ -n "$T" ] && python tools/trace_steps.py "$T" 4 45 > $O/cfg5_all_step_breakdown.txt 2>&1 && head -52 $O/cfg5_all_step_breakdown.txt | cut -c1-150
