# Round 6 (late): slot granule 16 for the mask head's dynamic batch — find-db entries for the additional batch sizes, then A/B 32 vs 16
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
O=gpurun_out/r06slotsdb16; mkdir -p $O; export MIOPEN_LOG_LEVEL=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
jl() { grep -E "^\{" "$1" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$2', d['value'], 'img/s', d.get('ms_per_step'), 'ms', d.get('loss_finite'), d.get('mask_slots'))" 2>/dev/null || tail -3 "$1"; }
DB=$GRAFT_REPO_ROOT/gpurun_out/r06slotsdb16/db
S="--steps 3 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-fixed-quota-line --miopen-search --export-miopen-db $DB"
D="MODEL.RESNETS.STAGE_WITH_DCN (False,True,True,True)"
for m in 16 16,32 32,48 48,64 64,80 80,96 96,112 112,128; do
  DETOPS_MASK_SLOTS=$m timeout 600 python bench.py $S < /dev/null > $O/s_f32_$m.log 2>&1
  DETOPS_MASK_SLOTS=$m timeout 600 python bench.py $S --dtype bfloat16 < /dev/null > $O/s_bf16_$m.log 2>&1
  DETOPS_MASK_SLOTS=$m timeout 600 python bench.py $S --config e2e_mask_rcnn_R_101_FPN_1x.yaml --dtype float16 $D < /dev/null > $O/s_cfg5_$m.log 2>&1
  el search-$m
done
wc -l $DB/db/*.txt
export MIOPEN_USER_DB_PATH=$DB/db MIOPEN_CUSTOM_CACHE_DIR=$DB/cache
B="python bench.py --steps 100 --warmup 15 --no-cpu-baseline --no-kernel-timing --no-fixed-quota-line"
for rep in 1 2; do
  DETOPS_MASK_SLOT_GRANULE=32 timeout 400 $B < /dev/null > $O/f32_g32_$rep.log 2>&1; jl $O/f32_g32_$rep.log f32-granule-32
  DETOPS_MASK_SLOT_GRANULE=16 timeout 400 $B < /dev/null > $O/f32_g16_$rep.log 2>&1; jl $O/f32_g16_$rep.log f32-granule-16
done
DETOPS_MASK_SLOT_GRANULE=32 timeout 400 $B --dtype bfloat16 < /dev/null > $O/bf16_g32.log 2>&1; jl $O/bf16_g32.log bf16-granule-32
DETOPS_MASK_SLOT_GRANULE=16 timeout 400 $B --dtype bfloat16 < /dev/null > $O/bf16_g16.log 2>&1; jl $O/bf16_g16.log bf16-granule-16
rm -rf $DB/cache/*.tmp; du -sh gpurun_out | tail -1
