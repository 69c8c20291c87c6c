# Round 6 (late): find-db entries for the mask head's batch sizes under DETOPS_MASK_SLOTS=dynamic (2 images x ceil32(positives):
# 64 .. 256 ROIs in steps of 32) — normal Find (cudnn.benchmark) from the shipped database: only the new problem keys are searched.
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
O=gpurun_out/r06slotsdb; mkdir -p $O; export MIOPEN_LOG_LEVEL=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
jl() { grep -E "^\{" "$1" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$2', d['value'], 'img/s', d.get('ms_per_step'), 'ms', d.get('loss_finite'), d.get('mask_slots'), d.get('miopen'))" 2>/dev/null || tail -3 "$1"; }
DB=$GRAFT_REPO_ROOT/gpurun_out/r06slotsdb/db
S="--steps 3 --warmup 3 --no-cpu-baseline --no-kernel-timing --miopen-search --export-miopen-db $DB"
D="MODEL.RESNETS.STAGE_WITH_DCN (False,True,True,True)"
for m in 32 32,64 64 64,96 96 96,128; do
  DETOPS_MASK_SLOTS=$m timeout 600 python bench.py $S < /dev/null > $O/s_f32_$m.log 2>&1
  DETOPS_MASK_SLOTS=$m timeout 600 python bench.py $S --dtype bfloat16 < /dev/null > $O/s_bf16_$m.log 2>&1
  DETOPS_MASK_SLOTS=$m timeout 600 python bench.py $S --config e2e_mask_rcnn_R_101_FPN_1x.yaml --dtype float16 $D < /dev/null > $O/s_cfg5_$m.log 2>&1
  el search-$m
done
wc -l $DB/db/*.txt
B="python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-kernel-timing"
for rep in 1 2; do
  timeout 400 $B < /dev/null > $O/f32_shipped$rep.log 2>&1; jl $O/f32_shipped$rep.log f32-dynamic-shipped-db
  ( export MIOPEN_USER_DB_PATH=$DB/db MIOPEN_CUSTOM_CACHE_DIR=$DB/cache; timeout 400 $B < /dev/null > $O/f32_new$rep.log 2>&1; jl $O/f32_new$rep.log f32-dynamic-new-db )
done
( export MIOPEN_USER_DB_PATH=$DB/db MIOPEN_CUSTOM_CACHE_DIR=$DB/cache
  for m in 32 96; do DETOPS_MASK_SLOTS=$m timeout 400 $B < /dev/null > $O/f32_new_forced$m.log 2>&1; jl $O/f32_new_forced$m.log f32-forced-$m-new-db; done
  timeout 400 $B --dtype bfloat16 < /dev/null > $O/bf16_new.log 2>&1; jl $O/bf16_new.log bf16-dynamic-new-db
  timeout 400 $B --config e2e_mask_rcnn_R_101_FPN_1x.yaml --dtype float16 $D < /dev/null > $O/cfg5_new.log 2>&1; jl $O/cfg5_new.log cfg5-dynamic-new-db )
timeout 400 $B --dtype bfloat16 < /dev/null > $O/bf16_shipped.log 2>&1; jl $O/bf16_shipped.log bf16-dynamic-shipped-db
rm -rf $DB/cache/*.tmp; du -sh gpurun_out | tail -1
