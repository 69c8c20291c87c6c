# Round 6 (late): longer soak on the last tree — dynamic mask-head slots + per-stage half-weight casts, 1000 iterations each
O=gpurun_out/r06soak2; mkdir -p $O; export MIOPEN_LOG_LEVEL=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T="python tools/train_net.py --skip-test"; OPTS="SOLVER.MAX_ITER 1000 SOLVER.IMS_PER_BATCH 2 SOLVER.BASE_LR 0.0025 SOLVER.CHECKPOINT_PERIOD 100000"
chk() { python - "$1" <<'PY'
import re,sys
bad=0; n=0; last=None
for line in open(sys.argv[1]):
    m=re.search(r"iter: (\d+) .*?loss: ([0-9.naninf]+) \(", line)
    if m:
        n+=1; last=line.strip()[-0:]
        v=m.group(2)
        if "nan" in v or "inf" in v: bad+=1
print(sys.argv[1].split("/")[-1], "log lines", n, "non-finite", bad)
if last: print("   ", re.sub(r"^.*?iter:", "iter:", last)[:230])
PY
}
timeout 1200 $T --config-file e2e_mask_rcnn_R_50_FPN_1x.yaml $OPTS DTYPE bfloat16 OUTPUT_DIR /tmp/o_bf16 < /dev/null > $O/bf16.log 2>&1; chk $O/bf16.log
timeout 1200 $T --config-file e2e_mask_rcnn_R_50_FPN_1x.yaml $OPTS DTYPE float16 OUTPUT_DIR /tmp/o_f16 < /dev/null > $O/f16.log 2>&1; chk $O/f16.log
timeout 1200 $T --config-file e2e_mask_rcnn_R_50_FPN_1x.yaml $OPTS OUTPUT_DIR /tmp/o_f32 < /dev/null > $O/f32.log 2>&1; chk $O/f32.log
grep -ciE "traceback|error" $O/*.log
