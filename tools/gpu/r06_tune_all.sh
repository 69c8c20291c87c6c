# Round 6 (late): the perf-config search of tools/gpu/r06_tune.sh for the other BASELINE configurations (and the eval forward),
# into ONE database that starts from the shipped one; then shipped vs tuned, alternating, per configuration.
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
O=gpurun_out/r06tuneall; mkdir -p $O; export MIOPEN_LOG_LEVEL=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
jl() { grep -E "^\{" "$1" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$2', d['value'], 'img/s', d.get('ms_per_step'), 'ms', d.get('loss_finite'), d.get('miopen'))" 2>/dev/null || tail -3 "$1"; }
DB=$GRAFT_REPO_ROOT/gpurun_out/r06tuneall/db
S="--steps 3 --warmup 3 --no-cpu-baseline --no-kernel-timing --miopen-search --export-miopen-db $DB"
D="MODEL.RESNETS.STAGE_WITH_DCN (False,True,True,True)"
export MIOPEN_FIND_ENFORCE=4
timeout 900 python bench.py $S --dtype bfloat16 < /dev/null > $O/s_bf16.log 2>&1; el search-bf16
timeout 900 python bench.py $S --config e2e_mask_rcnn_R_101_FPN_1x.yaml --dtype float16 $D < /dev/null > $O/s_cfg5.log 2>&1; el search-cfg5
timeout 900 python bench.py $S --config e2e_faster_rcnn_R_50_FPN_1x.yaml < /dev/null > $O/s_faster.log 2>&1; el search-faster
timeout 900 python bench.py $S --config retinanet/retinanet_R-50-FPN_1x.yaml < /dev/null > $O/s_retina.log 2>&1; el search-retina
timeout 900 python bench.py --eval --steps 3 --warmup 3 --miopen-search --export-miopen-db $DB < /dev/null > $O/s_eval.log 2>&1; el search-eval
timeout 900 python bench.py --eval --steps 3 --warmup 3 --dtype bfloat16 --miopen-search --export-miopen-db $DB < /dev/null > $O/s_eval_bf16.log 2>&1; el search-eval-bf16
timeout 900 python bench.py $S < /dev/null > $O/s_f32.log 2>&1; el search-f32
unset MIOPEN_FIND_ENFORCE
wc -l $DB/db/*.txt
B="python bench.py --steps 40 --warmup 12 --no-cpu-baseline --no-kernel-timing"
ab() { N=$1; shift
  for rep in 1 2; do
    timeout 400 $B "$@" < /dev/null > $O/${N}_shipped$rep.log 2>&1; jl $O/${N}_shipped$rep.log ${N}-shipped
    ( export MIOPEN_USER_DB_PATH=$DB/db MIOPEN_CUSTOM_CACHE_DIR=$DB/cache; timeout 400 $B "$@" < /dev/null > $O/${N}_tuned$rep.log 2>&1; jl $O/${N}_tuned$rep.log ${N}-tuned )
  done; el $N; }
ab f32
ab bf16 --dtype bfloat16
ab cfg5 --config e2e_mask_rcnn_R_101_FPN_1x.yaml --dtype float16 $D
ab faster --config e2e_faster_rcnn_R_50_FPN_1x.yaml
ab retina --config retinanet/retinanet_R-50-FPN_1x.yaml
rm -rf $DB/cache/*.tmp; du -sh gpurun_out | tail -1
