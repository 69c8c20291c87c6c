mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cat > /tmp/probe.py <<'PY'
import sys, os; sys.path[:0]=['tools','maskrcnn-benchmark_amd','.']
import torch, numpy as np, synth
from maskrcnn_benchmark import _C as C
shapes=[(2,256,h,w) for (h,w) in synth.fpn_shapes()[:4]]; scales=[1.0/s for s in synth.FPN_STRIDES[:4]]
K=int(sys.argv[1]); ph=int(sys.argv[2])
rois=synth.fpn_rois(per_image=K//2); lv=synth.level_map(rois)
g=torch.randn(K,256,ph,ph,device='cuda'); tr=torch.from_numpy(rois).cuda(); tl=torch.from_numpy(lv).cuda()
for _ in range(10): C.roi_align_fpn_backward(g,tr,tl,shapes,scales,ph,ph,2)
torch.cuda.synchronize()
PY
for cfg in "2 7" "1024 7" "256 14"; do
  rm -rf gpurun_out/probe
  timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/probe -o p -- python /tmp/probe.py $cfg > gpurun_out/probe.log 2>&1
  echo "== K ph = $cfg"
  python - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/probe/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:4]:
        print(r['Name'][:70], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
PY
done
