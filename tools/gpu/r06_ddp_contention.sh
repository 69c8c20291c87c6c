# Round 6 (VERDICT r05 #4a): the data-parallel step beside a CONTENTION STAND-IN.  At world size 1 the bucket all-reduce is a
# no-op copy; a real N > 1 ring all-reduce holds a fixed number of CUs (its channels) for the duration of each 25 MB bucket.
# DETOPS_DDP_STANDIN="workgroups:microseconds" launches detops_debug_occupy on the wrapper's side stream behind every bucket's
# (1-rank) all-reduce: `workgroups` 1024-thread workgroups spinning for `microseconds`.  Recorded per setting: ms per step,
# exposed all-reduce wait, ROIAlign backward / NMS entry-point times inside the step, NMS repair count.
O=gpurun_out/r06_ddp; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
row() { grep -E "^\{" "$1" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d.get('kernels',{})
g=lambda p: next((round(v['mean_us'],1) for n,v in k.items() if n.startswith(p)), None)
print('%-22s %7.3f ms/step  exposed %s ms  roi_bwd_box %s us  roi_bwd_mask %s us  nms %s us  repaired %s  comm %s' % ('$2', d['ms_per_step'], d.get('exposed_allreduce_ms'), g('roi_align_fpn_bwd[K=1024'), g('roi_align_fpn_bwd[K=256'), g('nms_batched'), d.get('nms_repaired_segments'), (d.get('ddp_comm') or {}).get('mode')))" 2>/dev/null || tail -2 "$1"; }
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --kernel-timing-steps 30"
timeout 300 $B < /dev/null > $O/plain.log 2>&1; row $O/plain.log "plain (no wrapper)"
timeout 300 $B --force-ddp < /dev/null > $O/ddp_0.log 2>&1; row $O/ddp_0.log "ddp, no stand-in"
for wg in 16 32 64; do for us in 300 1000 2000; do
  DETOPS_DDP_STANDIN=$wg:$us timeout 300 $B --force-ddp < /dev/null > $O/ddp_${wg}_${us}.log 2>&1; row $O/ddp_${wg}_${us}.log "ddp $wg WG x $us us"
done; done
for prio in low high; do
  DETOPS_DDP_PRIO=$prio DETOPS_DDP_STANDIN=32:1000 timeout 300 $B --force-ddp < /dev/null > $O/ddp_prio_$prio.log 2>&1; row $O/ddp_prio_$prio.log "32x1000 prio=$prio"
done
