# round 5: what is left of the data-parallel wrapper's cost (direct RCCL path)?   gpurun --timeout 600 -- 'bash tools/gpu/r05c_ddp_variants.sh'
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; O=gpurun_out/r05c; mkdir -p $O; export MIOPEN_LOG_LEVEL=1
timeout 200 python -m pytest tests/test_ops_gpu.py -m gpu -q -p no:cacheprovider -k "beside_rccl" < /dev/null > $O/pytest.log 2>&1; tail -2 $O/pytest.log | cut -c1-200
B="python bench.py --steps 40 --warmup 12 --no-cpu-baseline --no-kernel-timing"
show() { grep -E "^\{" $O/$1.log | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], 'img/s', d['ms_per_step'], 'ms host', d.get('host_enqueue_ms_per_step'), d.get('ddp_comm'))" || tail -3 $O/$1.log; }
run() { n=$1; shift; env "$@" timeout 120 $B $EXTRA < /dev/null > $O/$n.log 2>&1; show $n; }
EXTRA=""; run plain_1 X=1
EXTRA="--force-ddp"
run normal_1 DETOPS_DDP_COMM=direct DETOPS_DDP_PRIO=normal
run low_1 DETOPS_DDP_COMM=direct DETOPS_DDP_PRIO=low
run high_1 DETOPS_DDP_COMM=direct DETOPS_DDP_PRIO=high
run main_1 DETOPS_DDP_COMM=direct DETOPS_DDP_PRIO=main
run nooverlap_1 DETOPS_DDP_COMM=direct DETOPS_DDP_PRIO=normal DETOPS_DDP_OVERLAP=0
EXTRA="--force-ddp --bucket-mb 64"; run b64 DETOPS_DDP_COMM=direct DETOPS_DDP_PRIO=normal
EXTRA="--force-ddp --bucket-mb 200"; run b200 DETOPS_DDP_COMM=direct DETOPS_DDP_PRIO=normal
EXTRA=""; run plain_2 X=1
EXTRA="--force-ddp"
run normal_2 DETOPS_DDP_COMM=direct DETOPS_DDP_PRIO=normal
run low_2 DETOPS_DDP_COMM=direct DETOPS_DDP_PRIO=low
run high_2 DETOPS_DDP_COMM=direct DETOPS_DDP_PRIO=high
