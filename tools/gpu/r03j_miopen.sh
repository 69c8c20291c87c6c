# r03j: new tests (train_net.py subprocess, focal entry points, mask dtypes) + MIOpen find-db build seeded with the r02 export
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
mkdir -p gpurun_out; export MIOPEN_LOG_LEVEL=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -k "train_net or focal_model_entry or mask_targets_bit" > gpurun_out/pytest_new.log 2>&1; tail -4 gpurun_out/pytest_new.log | cut -c1-300; el pytest-new
timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline > gpurun_out/bench_immediate.log 2>&1; grep -E "^\{" gpurun_out/bench_immediate.log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('immediate', d['value'], d['ms_per_step'], d['miopen'], d.get('roofline')); print(d.get('kernel_families_ms_per_step'))"; el bench-immediate
rm -rf gpurun_out/miopen_db_r3; mkdir -p gpurun_out/miopen_db_r3; cp -r maskrcnn-benchmark_amd/miopen_db_seed/* gpurun_out/miopen_db_r3/
timeout 1000 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --miopen-search --export-miopen-db gpurun_out/miopen_db_r3 > gpurun_out/bench_search.log 2>&1; echo "search rc=$?"; grep -E "^\{" gpurun_out/bench_search.log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('search', d['value'], d['ms_per_step'], d['miopen'])"; grep "warm-up step" gpurun_out/bench_search.log | tail -3; du -sh gpurun_out/miopen_db_r3; el bench-search
