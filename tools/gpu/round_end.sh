# Reproduces the committed round profiles on one MI355X box:  gpurun -- 'bash tools/gpu/round_end.sh [parts]'
# parts (default all): tests bench trace opbench pmc extra
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
PARTS="${*:-tests bench trace opbench pmc extra}"
has() { case " $PARTS " in *" $1 "*) return 0;; esac; return 1; }
O=gpurun_out/round; mkdir -p $O; export MIOPEN_LOG_LEVEL=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
jline() { grep -E "^\{" "$1" | tail -1; }
brief() { jline "$1" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}
print('$2', d['value'], 'img/s', d['ms_per_step'], 'ms host', d.get('host_enqueue_ms_per_step'), d.get('ddp'), '| roofline', r.get('kernel'), r.get('frac'), 'rocprof', r.get('rocprof_us'), 'traffic', r.get('traffic'))
print('  families', d.get('kernel_families_ms_per_step'))" 2>/dev/null || tail -3 "$1"; }
if has tests; then
  timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider < /dev/null > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
  grep -E "passed|failed|rc=|Error" $O/pytest_gpu.log | tail -4 | cut -c1-200; el pytest
  timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" < /dev/null > $O/smoke.log 2>&1; tail -1 $O/smoke.log; el smoke
fi
if has bench; then
  timeout 600 python bench.py < /dev/null > $O/bench_f32.log 2>&1; jline $O/bench_f32.log > $O/bench_f32.json; brief $O/bench_f32.log f32; el bench
fi
if has trace; then
  P=/tmp/prof_bench; rm -rf $P
  timeout 500 rocprofv3 --kernel-trace --output-format csv -d $P -o bench -- python bench.py --steps 6 --warmup 6 --no-cpu-baseline --no-fixed-quota-line < /dev/null > $O/prof_bench.log 2>&1
  T=$(find $P -name "*kernel_trace.csv" | head -1)
  if [ -n "$T" ]; then
    python tools/trace_steps.py "$T" 4 60 > $O/bench_f32_step_breakdown.txt 2>&1
    python tools/kernel_times.py $P "" 2>/dev/null | grep -E "roi_align|roi_nhwc|roi_bwd_prep|nms_|frozen_bn|topdown|bias_act|bias_grad|column_sum|focal|match_|sampler|mask_targets|rpn_loss|rpn_decode|dcn|col2im|im2col|sampleT|coord_nhwc|nchw_to_nhwc" > $O/bench_kernel_times.txt
    head -3 $O/bench_f32_step_breakdown.txt; head -12 $O/bench_kernel_times.txt | cut -c1-150
  fi; el trace
fi
if has opbench; then
  timeout 900 python tools/opbench.py --iters 50 --layout both --json $O/opbench.json < /dev/null > $O/opbench.log 2>&1; grep -E "roi_align_(fwd|bwd) (fpn|cfg1)|nms batched|frozen_bn|focal|match_boxes|sample_labels|dcn_block|roi_pool|psroi" $O/opbench.log | cut -c1-170 | head -70; el opbench
  # the > L3 variant (4 img/GPU: 365.6 MB of maps against the 256 MiB Infinity Cache)
  timeout 300 python tools/opbench.py --only roi_sets --sets model-random-init --images 4 --iters 30 < /dev/null > $O/roi_align_l3_variant.log 2>&1; grep roi_align $O/roi_align_l3_variant.log | cut -c1-150; el l3-variant
  timeout 200 python tools/gpu/cfg1_bwd.py 0 50 < /dev/null > $O/cfg1_bwd.log 2>&1; grep cfg1 $O/cfg1_bwd.log; el cfg1
  timeout 200 python tools/gpu/ring_timeline.py model-random-init < /dev/null > $O/ring_timeline.txt 2>&1; head -8 $O/ring_timeline.txt; el timeline
fi
if has pmc; then
  PM="python tools/opbench.py --only roi_sets --heads box --dir bwd --iters 5 --sets model-random-init --layout nhwc"
  for pass in "sq:SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
              "sq2:SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
              "tcc:TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "fw:FETCH_SIZE" "ww:WRITE_SIZE"; do
    n=${pass%%:*}; c=${pass#*:}; rm -rf /tmp/pmc_$n
    timeout 150 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$n -o x -- $PM < /dev/null > $O/pmc_$n.log 2>&1
  done
  python tools/pmc_diag.py /tmp/pmc_sq /tmp/pmc_sq2 /tmp/pmc_tcc /tmp/pmc_fw /tmp/pmc_ww > $O/roi_align_bwd_ring_pmc.txt 2>&1; grep -v "roi_order" $O/roi_align_bwd_ring_pmc.txt | head -40; el pmc-ring
  for d in fwd; do
    PF="python tools/opbench.py --only roi_sets --heads box --dir $d --iters 5 --sets model-random-init --layout nhwc"
    for pass in "sq:SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
                "sq2:SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
                "tcc:TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "fw:FETCH_SIZE" "ww:WRITE_SIZE"; do
      n=${pass%%:*}; c=${pass#*:}; rm -rf /tmp/pmcf_$n
      timeout 150 rocprofv3 --pmc $c --output-format csv -d /tmp/pmcf_$n -o x -- $PF < /dev/null > $O/pmcf_$n.log 2>&1
    done
    python tools/pmc_diag.py /tmp/pmcf_sq /tmp/pmcf_sq2 /tmp/pmcf_tcc /tmp/pmcf_fw /tmp/pmcf_ww > $O/roi_align_fwd_pmc.txt 2>&1; grep -A28 "roi_align_fwd_nhwc" $O/roi_align_fwd_pmc.txt | head -32; el pmc-fwd
  done
  # HBM traffic of the ROIAlign launches over the channels-last pyramid (the layout the step runs; both heads, both directions):
  # separate FETCH_SIZE / WRITE_SIZE passes -> traffic_roi_nhwc.json (bench.py looks the newest profiles/*traffic*.json up)
  TRR="python tools/opbench.py --only roi_sets --sets model-random-init --layout nhwc --iters 5"
  rm -rf /tmp/trr_f /tmp/trr_w
  timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/trr_f -o x -- $TRR < /dev/null > $O/traffic_roi_fetch.log 2>&1
  timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/trr_w -o x -- $TRR < /dev/null > $O/traffic_roi_write.log 2>&1
  python tools/pmc_traffic.py /tmp/trr_f /tmp/trr_w $O/traffic_roi_nhwc.json 2>&1 | cut -c1-150 > $O/traffic_roi_nhwc.txt; cat $O/traffic_roi_nhwc.txt | head; el pmc-roi-traffic
  # (the NCHW launches' calibrated table is tools/gpu/r05j_fetch_calib.sh -> profiles/r05j_traffic.*; the pass below keeps the
  #  FrozenBN / NMS / deformable-conv rows)
  TR="python tools/opbench.py --only frozen_bn,nms,dcn_block --iters 5"
  rm -rf /tmp/tr_f /tmp/tr_w
  timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/tr_f -o x -- $TR < /dev/null > $O/traffic_fetch.log 2>&1
  timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/tr_w -o x -- $TR < /dev/null > $O/traffic_write.log 2>&1
  python tools/pmc_traffic.py /tmp/tr_f /tmp/tr_w $O/traffic.json 2>&1 | cut -c1-150 > $O/traffic.txt; grep -v "nms_sort\|nms_scan\|nms_mask" $O/traffic.txt | head -40; el pmc-traffic
  rm -rf /tmp/kt_dcn
  timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_dcn -o x -- python tools/opbench.py --only dcn_block --iters 10 < /dev/null > $O/dcn_block_trace.log 2>&1
  python tools/kernel_times.py /tmp/kt_dcn "" 2>/dev/null | grep -v "Cijk\|at::\|rocclr\|elementwise" > $O/dcn_block_kernel_times.txt; head -24 $O/dcn_block_kernel_times.txt | cut -c1-150; el dcn-trace
fi
if has extra; then
  B="python bench.py --steps 40 --warmup 12 --no-cpu-baseline"
  timeout 300 $B --dtype bfloat16 < /dev/null > $O/bench_bf16.log 2>&1; jline $O/bench_bf16.log > $O/bench_bf16.json; brief $O/bench_bf16.log bf16; el bf16
  timeout 400 $B --config e2e_mask_rcnn_R_101_FPN_1x.yaml --dtype float16 MODEL.RESNETS.STAGE_WITH_DCN "(False, True, True, True)" < /dev/null > $O/bench_cfg5.log 2>&1; jline $O/bench_cfg5.log > $O/bench_cfg5.json; brief $O/bench_cfg5.log cfg5; el cfg5
  timeout 400 $B --hip-graph --config e2e_mask_rcnn_R_101_FPN_1x.yaml --dtype float16 MODEL.RESNETS.STAGE_WITH_DCN "(False, True, True, True)" < /dev/null > $O/bench_cfg5_graph.log 2>&1; jline $O/bench_cfg5_graph.log > $O/bench_cfg5_graph.json; brief $O/bench_cfg5_graph.log cfg5-hip-graph; el cfg5-graph
  timeout 300 $B < /dev/null > $O/bench_plain40.log 2>&1; jline $O/bench_plain40.log > $O/bench_plain40.json; brief $O/bench_plain40.log plain-40-steps; el plain40
  timeout 300 $B --force-ddp < /dev/null > $O/bench_forceddp.log 2>&1; jline $O/bench_forceddp.log > $O/bench_forceddp.json; brief $O/bench_forceddp.log force-ddp-direct; el force-ddp
  DETOPS_DDP_COMM=pg timeout 300 $B --force-ddp < /dev/null > $O/bench_forceddp_pg.log 2>&1; jline $O/bench_forceddp_pg.log > $O/bench_forceddp_pg.json; brief $O/bench_forceddp_pg.log force-ddp-pg; el force-ddp-pg
  timeout 300 $B --config e2e_faster_rcnn_R_50_FPN_1x.yaml < /dev/null > $O/bench_faster.log 2>&1; jline $O/bench_faster.log > $O/bench_faster.json; brief $O/bench_faster.log faster; el faster
  timeout 300 $B --config retinanet/retinanet_R-50-FPN_1x.yaml < /dev/null > $O/bench_retinanet.log 2>&1; jline $O/bench_retinanet.log > $O/bench_retinanet.json; brief $O/bench_retinanet.log retinanet; el retinanet
  E="python bench.py --eval --steps 40 --warmup 8"
  timeout 300 $E < /dev/null > $O/bench_eval_f32.log 2>&1; jline $O/bench_eval_f32.log > $O/bench_eval_f32.json
  timeout 300 $E --dtype bfloat16 < /dev/null > $O/bench_eval_bf16.log 2>&1; jline $O/bench_eval_bf16.log > $O/bench_eval_bf16.json
  timeout 300 $E --config retinanet/retinanet_R-50-FPN_1x.yaml MODEL.RETINANET.INFERENCE_TH 0.0 < /dev/null > $O/bench_eval_retinanet.log 2>&1; jline $O/bench_eval_retinanet.log > $O/bench_eval_retinanet.json
  for f in eval_f32 eval_bf16 eval_retinanet; do python -c "
import json,sys
d=json.load(open('$O/bench_$f.json')); print('$f', d['value'], 'img/s', d['s_per_image'], 's/im detections', d['detections_per_image'], d['detections_finite'])" 2>/dev/null; done; el eval
  timeout 200 python tools/gpu/nms_probe.py < /dev/null > $O/nms_probe.txt 2>&1; tail -8 $O/nms_probe.txt | cut -c1-200; el nms-probe
fi
du -sh gpurun_out | tail -1
