mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cat > /tmp/probe.py <<'PY'
import sys, os; sys.path[:0]=['tools','maskrcnn-benchmark_amd','.']
import torch, numpy as np, synth
from maskrcnn_benchmark import _C as C
from opbench import dev_time_us
shapes=[(2,256,h,w) for (h,w) in synth.fpn_shapes()[:4]]; scales=[1.0/s for s in synth.FPN_STRIDES[:4]]
for K,ph in ((1024,7),(256,14)):
    rois=synth.fpn_rois(per_image=K//2); lv=synth.level_map(rois)
    g=torch.randn(K,256,ph,ph,device='cuda'); tr=torch.from_numpy(rois).cuda(); tl=torch.from_numpy(lv).cuda()
    for dbg in ("0","1"):
        os.environ["DETOPS_ROIALIGN_BWD_DEBUG"]=dbg
        print("K=%d %dx%d debug=%s: %.1f us"%(K,ph,ph,dbg, dev_time_us(lambda: C.roi_align_fpn_backward(g,tr,tl,shapes,scales,ph,ph,2),30)))
PY
python /tmp/probe.py
