# round 5: the data-parallel wrapper with flat buckets + native pack / SGD kernels (csrc/optim.hip) vs the torch multi-tensor ops
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; O=gpurun_out/r05g; mkdir -p $O; export MIOPEN_LOG_LEVEL=1
timeout 300 python -m pytest tests/test_model_gpu.py -m gpu -q -p no:cacheprovider -k "forced_ddp or ddp or data_parallel" < /dev/null > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -1; grep -E "^E " $O/pytest.log | head -5
B="python bench.py --steps 40 --warmup 12 --no-cpu-baseline --no-kernel-timing"
show() { grep -E "^\{" $O/$1.log | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], 'img/s', d['ms_per_step'], 'ms host', d.get('host_enqueue_ms_per_step'), d.get('ddp_comm'), 'loss_finite', d.get('loss_finite'))" || tail -3 $O/$1.log; }
run() { n=$1; shift; env "$@" timeout 120 $B $EXTRA < /dev/null > $O/$n.log 2>&1; show $n; }
for rep in 1 2; do
EXTRA=""; run plain_$rep X=1
EXTRA="--force-ddp"
run native_$rep DETOPS_DDP_COMM=direct
run torchops_$rep DETOPS_DDP_COMM=direct DETOPS_DDP_NATIVE=0
run pg_$rep DETOPS_DDP_COMM=pg
done
EXTRA="--force-ddp --dtype float16"; run native_fp16 DETOPS_DDP_COMM=direct
EXTRA="--force-ddp --bucket-mb 64"; run native_b64 DETOPS_DDP_COMM=direct
