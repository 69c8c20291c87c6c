# Round 6: soak runs on the final tree — the reference's training command line for 300 iterations on three configurations
# (finite losses, no repaired NMS segments, the data-parallel wrapper at world size 1), and a 300-step bench of the headline config.
O=gpurun_out/r06soak; mkdir -p $O; export MIOPEN_LOG_LEVEL=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T="python tools/train_net.py --skip-test"; OPTS="SOLVER.MAX_ITER 300 SOLVER.IMS_PER_BATCH 2 SOLVER.BASE_LR 0.0025 SOLVER.CHECKPOINT_PERIOD 100000"
timeout 900 $T --config-file e2e_mask_rcnn_R_50_FPN_1x.yaml $OPTS OUTPUT_DIR /tmp/out_mask_f32 < /dev/null > $O/train_mask_f32.log 2>&1; grep -E "iter: (20|100|200|300) |comm_mode|nms|Error|Traceback" $O/train_mask_f32.log | cut -c1-260 | tail -8
timeout 900 $T --config-file e2e_mask_rcnn_R_50_FPN_1x.yaml $OPTS DTYPE bfloat16 OUTPUT_DIR /tmp/out_mask_bf16 < /dev/null > $O/train_mask_bf16.log 2>&1; grep -E "iter: (20|100|200|300) |Error|Traceback" $O/train_mask_bf16.log | cut -c1-260 | tail -6
timeout 900 $T --config-file e2e_mask_rcnn_R_101_FPN_1x.yaml $OPTS DTYPE float16 MODEL.RESNETS.STAGE_WITH_DCN "(False,True,True,True)" OUTPUT_DIR /tmp/out_cfg5 < /dev/null > $O/train_cfg5.log 2>&1; grep -E "iter: (20|100|200|300) |Error|Traceback" $O/train_cfg5.log | cut -c1-260 | tail -6
timeout 900 $T --config-file retinanet/retinanet_R-50-FPN_1x.yaml $OPTS OUTPUT_DIR /tmp/out_retinanet < /dev/null > $O/train_retinanet.log 2>&1; grep -E "iter: (20|100|200|300) |Error|Traceback" $O/train_retinanet.log | cut -c1-260 | tail -6
true; grep -E "^\{" $O/bench300.log | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench 300 steps', d['value'], 'img/s', d['ms_per_step'], 'ms loss_finite', d['loss_finite'], d['losses'], 'nms repaired', d['nms_repaired_segments'])"
