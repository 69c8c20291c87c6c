# round 5: fused FPN top-down step (csrc/fpn_topdown.hip) vs interpolate + add; full-size reference target expectations on the device
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; O=gpurun_out/r05h; mkdir -p $O; export MIOPEN_LOG_LEVEL=1
timeout 300 python -m pytest tests/test_targets_fullsize.py tests/test_model_gpu.py tests/test_whole_model_parity.py -m gpu -q -p no:cacheprovider < /dev/null > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -1; grep -E "^E " $O/pytest.log | head -5
timeout 100 python - > $O/topdown_parity.log 2>&1 <<'PY'
import sys, torch
sys.path.insert(0, "maskrcnn-benchmark_amd")
from maskrcnn_benchmark import _C
import torch.nn.functional as F
for dt, tol in ((torch.float32, 1e-6), (torch.bfloat16, 2e-2)):
    for (H, W, h, w) in ((200, 336, 100, 168), (100, 168, 50, 84), (50, 84, 25, 42), (25, 42, 13, 21), (31, 47, 16, 24)):
        lat = torch.randn(2, 256, H, W, device="cuda", dtype=dt, requires_grad=True); top = torch.randn(2, 256, h, w, device="cuda", dtype=dt, requires_grad=True)
        g = torch.randn(2, 256, H, W, device="cuda", dtype=dt)
        o = _C.fpn_topdown(lat, top); o.backward(g)
        l2, t2 = lat.detach().clone().requires_grad_(), top.detach().clone().requires_grad_()
        r = l2 + F.interpolate(t2, size=(H, W), mode="nearest"); r.backward(g)
        print(dt, (H, W, h, w), "fwd equal", torch.equal(o, r), "gtop maxdiff", float((top.grad.float() - t2.grad.float()).abs().max()), "glat equal", torch.equal(lat.grad, l2.grad))
        assert torch.allclose(o.float(), r.float(), rtol=tol, atol=tol) and torch.allclose(top.grad.float(), t2.grad.float(), rtol=tol, atol=4 * tol)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    lat = torch.randn(2, 256, 200, 336, device="cuda", dtype=dt); top = torch.randn(2, 256, 100, 168, device="cuda", dtype=dt)
    for _ in range(3): _C.fpn_topdown(lat, top); lat + F.interpolate(top, size=(200, 336), mode="nearest")
    ev[0].record()
    for _ in range(20): _C.fpn_topdown(lat, top)
    ev[1].record(); ev[2].record()
    for _ in range(20): lat + F.interpolate(top, size=(200, 336), mode="nearest")
    ev[3].record(); torch.cuda.synchronize()
    print(dt, "P2 step: fused %.1f us, interpolate + add %.1f us" % (ev[0].elapsed_time(ev[1]) * 50, ev[2].elapsed_time(ev[3]) * 50))
PY
tail -14 $O/topdown_parity.log | cut -c1-200
B="python bench.py --steps 40 --warmup 12 --no-cpu-baseline --no-kernel-timing"
show() { grep -E "^\{" $O/$1.log | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], 'img/s', d['ms_per_step'], 'ms host', d.get('host_enqueue_ms_per_step'))" || tail -3 $O/$1.log; }
for rep in 1 2; do for v in torch fused; do
  DETOPS_FPN_TOPDOWN=$v timeout 120 $B < /dev/null > $O/f32_${v}_$rep.log 2>&1; show f32_${v}_$rep
done; done
for v in torch fused; do DETOPS_FPN_TOPDOWN=$v timeout 120 $B --dtype bfloat16 < /dev/null > $O/bf16_$v.log 2>&1; show bf16_$v; done
