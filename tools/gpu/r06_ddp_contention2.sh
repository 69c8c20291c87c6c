# Round 6: follow-up of r06_ddp_contention.sh — the normal-priority side stream SERIALISED with the compute stream (step time
# grew by buckets x stand-in time at any CU count); explicit-priority streams overlapped.  Which setting, and is it the HW queue?
O=gpurun_out/r06_ddp2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
row() { grep -E "^\{" "$1" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d.get('kernels',{})
g=lambda p: next((round(v['mean_us'],1) for n,v in k.items() if n.startswith(p)), None)
print('%-34s %7.3f ms/step  exposed %s ms  roi_bwd_box %s us  roi_bwd_mask %s us  nms %s us  repaired %s' % ('$2', d['ms_per_step'], d.get('exposed_allreduce_ms'), g('roi_align_fpn_bwd[K=1024'), g('roi_align_fpn_bwd[K=256'), g('nms_batched'), d.get('nms_repaired_segments')))" 2>/dev/null || tail -2 "$1"; }
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --kernel-timing-steps 30"
for prio in low high; do for spec in 0:0 16:1000 32:1000 64:1000 32:2000 128:1000; do
  DETOPS_DDP_PRIO=$prio DETOPS_DDP_STANDIN=$spec timeout 300 $B --force-ddp < /dev/null > $O/p_${prio}_$spec.log 2>&1; row $O/p_${prio}_$spec.log "prio=$prio standin=$spec"
done; done
for q in 4 8; do
  GPU_MAX_HW_QUEUES=$q DETOPS_DDP_STANDIN=32:1000 timeout 300 $B --force-ddp < /dev/null > $O/q$q.log 2>&1; row $O/q$q.log "prio=normal hwq=$q standin=32:1000"
  GPU_MAX_HW_QUEUES=$q DETOPS_DDP_PRIO=low DETOPS_DDP_STANDIN=32:1000 timeout 300 $B --force-ddp < /dev/null > $O/q${q}_low.log 2>&1; row $O/q${q}_low.log "prio=low hwq=$q standin=32:1000"
done
