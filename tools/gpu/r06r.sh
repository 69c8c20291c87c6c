# Round 6, experiment R: RetinaNet with its head channels-last (find-db search for the new keys), float64 operators,
# cfg-5 / Faster R-CNN on the final defaults.
O=gpurun_out/r06r; mkdir -p $O; export MIOPEN_LOG_LEVEL=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
jl() { grep -E "^\{" "$1" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-28s' % '$2', d['value'], 'img/s', d['ms_per_step'], 'ms', 'loss_finite', d['loss_finite'], d.get('layout'), d['miopen']['db'])" 2>/dev/null || tail -3 "$1"; }
timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "float64 or channels_last or stand_in or non_finite" -p no:cacheprovider < /dev/null > $O/pytest_new.log 2>&1; tail -2 $O/pytest_new.log
DB=$GRAFT_REPO_ROOT/gpurun_out/r06r/db
RET="--config retinanet/retinanet_R-50-FPN_1x.yaml"
timeout 1500 python bench.py --steps 4 --warmup 4 --no-cpu-baseline --no-kernel-timing --miopen-search --export-miopen-db $DB $RET < /dev/null > $O/search_ret.log 2>&1; jl $O/search_ret.log search-retinanet
timeout 1500 python bench.py --steps 4 --warmup 4 --no-cpu-baseline --no-kernel-timing --miopen-search --export-miopen-db $DB $RET --dtype bfloat16 < /dev/null > $O/search_ret16.log 2>&1; jl $O/search_ret16.log search-retinanet-bf16
wc -l $DB/db/*.txt | tail -1
export MIOPEN_USER_DB_PATH=$DB/db MIOPEN_CUSTOM_CACHE_DIR=$DB/cache
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-timing"
for lay in nchw backbone all; do timeout 400 $B $RET --layout $lay < /dev/null > $O/ret_$lay.log 2>&1; jl $O/ret_$lay.log retinanet-$lay; done
timeout 400 $B --config e2e_faster_rcnn_R_50_FPN_1x.yaml < /dev/null > $O/faster.log 2>&1; jl $O/faster.log faster-auto
timeout 400 $B < /dev/null > $O/mask.log 2>&1; jl $O/mask.log mask-auto
