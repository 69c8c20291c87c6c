# quick A/B of the FPN-fused ROIAlign launches: opbench lines + rocprofv3 kernel stats + the zero-hit floor
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 200 python tools/opbench.py --iters 30 --only roi_align --json gpurun_out/opbench_c.json > gpurun_out/opbench_c.log 2>&1
grep -v "^/opt" gpurun_out/opbench_c.log | grep "fpn-fused" | grep -v "LDS\|tile" | cut -c1-200
rm -rf gpurun_out/prof_bwd
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bwd -o bwd -- python tools/opbench.py --iters 20 --only roi_align_fpn > gpurun_out/prof_bwd.log 2>&1
python - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/prof_bwd/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]:
        print(r['Name'][:80], r['Calls'], r['AverageNs'], r['Percentage'])
PY
python - <<'PY'
import sys; sys.path[:0]=['tools','maskrcnn-benchmark_amd','.']
import torch, numpy as np, synth
from maskrcnn_benchmark import _C as C
from opbench import dev_time_us
shapes=[(2,256,h,w) for (h,w) in synth.fpn_shapes()[:4]]; scales=[1.0/s for s in synth.FPN_STRIDES[:4]]
for K,ph in ((2,7),(1024,7)):
    rois=synth.fpn_rois(per_image=K//2); lv=synth.level_map(rois)
    g=torch.randn(K,256,ph,ph,device='cuda'); tr=torch.from_numpy(rois).cuda(); tl=torch.from_numpy(lv).cuda()
    print("bwd K=%d: %.1f us"%(K, dev_time_us(lambda: C.roi_align_fpn_backward(g,tr,tl,shapes,scales,ph,ph,2),30)))
bufs=[torch.empty(s,device='cuda') for s in shapes]
print("4 memsets of the maps: %.1f us"%dev_time_us(lambda: [b.zero_() for b in bufs],30))
PY
