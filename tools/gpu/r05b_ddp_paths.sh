# round 5: where does the data-parallel wrapper's cost at world size 1 go?  One box, interleaved runs of bench.py:
#   plain | --force-ddp over ProcessGroupNCCL | over RCCL called directly on ONE side stream (low / normal priority) |
#   the same side stream without any collective.   gpurun --timeout 900 -- 'bash tools/gpu/r05b_ddp_paths.sh'
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; O=gpurun_out/r05b; mkdir -p $O; export MIOPEN_LOG_LEVEL=1
timeout 300 python -m pytest tests/test_model_gpu.py tests/test_ops_gpu.py tests/test_targets_gpu.py -m gpu -q -p no:cacheprovider -k "forced_ddp or nms_single_launch or injected_nms or fused_head" < /dev/null > $O/pytest.log 2>&1
tail -5 $O/pytest.log | cut -c1-300
B="python bench.py --steps 40 --warmup 12 --no-cpu-baseline"
show() { grep -E "^\{" $O/$1.log | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], 'img/s', d['ms_per_step'], 'ms host', d.get('host_enqueue_ms_per_step'), d.get('ddp_comm'), {k: v['mean_us'] for k, v in d['kernels'].items() if 'roi_align' in k or 'nms' in k})" || tail -3 $O/$1.log; }
for rep in 1 2; do
  timeout 120 $B < /dev/null > $O/plain_$rep.log 2>&1; show plain_$rep
  DETOPS_DDP_COMM=pg timeout 120 $B --force-ddp < /dev/null > $O/pg_$rep.log 2>&1; show pg_$rep
  DETOPS_DDP_COMM=direct timeout 120 $B --force-ddp < /dev/null > $O/direct_low_$rep.log 2>&1; show direct_low_$rep
  [ $rep = 1 ] && { DETOPS_DDP_COMM=direct DETOPS_DDP_PRIO=normal timeout 120 $B --force-ddp < /dev/null > $O/direct_normal_$rep.log 2>&1; show direct_normal_$rep; }
done
DETOPS_DDP_COMM=side-nocoll timeout 120 $B --force-ddp < /dev/null > $O/side_nocoll.log 2>&1; show side_nocoll
DETOPS_DDP_COMM=direct GPU_MAX_HW_QUEUES=1 timeout 120 $B --force-ddp < /dev/null > $O/direct_hwq1.log 2>&1; show direct_hwq1
DETOPS_DDP_COMM=direct timeout 120 $B --force-ddp --no-kernel-timing < /dev/null > $O/direct_notimer.log 2>&1; grep -E "^\{" $O/direct_notimer.log | tail -1 | cut -c1-200
timeout 120 $B --no-kernel-timing < /dev/null > $O/plain_notimer.log 2>&1; grep -E "^\{" $O/plain_notimer.log | tail -1 | cut -c1-200
