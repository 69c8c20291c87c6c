# Round 6, experiment T: inference lines (bench.py --eval), step breakdown of the fp32 and bf16 steps on the current defaults.
O=gpurun_out/r06t; mkdir -p $O; export MIOPEN_LOG_LEVEL=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
jl() { grep -E "^\{" "$1" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-28s' % '$2', d['value'], 'img/s', d['ms_per_step'], 'ms', d.get('detections_per_image'), d.get('detections_finite'), d.get('layout'))" 2>/dev/null || tail -5 "$1"; }
E="python bench.py --eval --steps 30 --warmup 8"
timeout 300 $E < /dev/null > $O/eval_mask.log 2>&1; jl $O/eval_mask.log eval-mask
timeout 300 $E MODEL.ROI_HEADS.SCORE_THRESH 0.0 < /dev/null > $O/eval_mask_t0.log 2>&1; jl $O/eval_mask_t0.log eval-mask-thresh0
timeout 300 $E --layout nchw < /dev/null > $O/eval_mask_nchw.log 2>&1; jl $O/eval_mask_nchw.log eval-mask-nchw
timeout 300 $E --dtype bfloat16 < /dev/null > $O/eval_mask_bf16.log 2>&1; jl $O/eval_mask_bf16.log eval-mask-bf16
timeout 300 $E --config retinanet/retinanet_R-50-FPN_1x.yaml < /dev/null > $O/eval_ret.log 2>&1; jl $O/eval_ret.log eval-retinanet
timeout 300 $E --config retinanet/retinanet_R-50-FPN_1x.yaml MODEL.RETINANET.INFERENCE_TH 0.0 < /dev/null > $O/eval_ret_t0.log 2>&1; jl $O/eval_ret_t0.log eval-retinanet-thresh0
for dt in float32 bfloat16; do
  P=/tmp/prof_$dt; rm -rf $P
  timeout 500 rocprofv3 --kernel-trace --output-format csv -d $P -o bench -- python bench.py --steps 6 --warmup 6 --no-cpu-baseline --no-kernel-timing --dtype $dt < /dev/null > $O/prof_$dt.log 2>&1
  T=$(find $P -name "*kernel_trace.csv" | head -1)
  [ -n "$T" ] && python tools/trace_steps.py "$T" 4 70 > $O/step_breakdown_$dt.txt 2>&1; head -4 $O/step_breakdown_$dt.txt
done
