# round 5: cfg-5 (R-101-FPN + DCN, fp16): is the step host-bound?  eager vs HIP-graph replay of the whole step (tools/graph_probe.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; O=gpurun_out/r05i; mkdir -p $O; export MIOPEN_LOG_LEVEL=1
timeout 200 python -m pytest tests/test_model_gpu.py -m gpu -q -p no:cacheprovider -k "forced_ddp or bucket_pack" < /dev/null > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -1; grep -E "^E " $O/pytest.log | head -5
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 GPU_MAX_HW_QUEUES=2
timeout 300 python tools/graph_probe.py --config e2e_mask_rcnn_R_101_FPN_1x.yaml --dtype float16 --steps 20 MODEL.RESNETS.STAGE_WITH_DCN "(False, True, True, True)" < /dev/null > $O/cfg5_graph.log 2>&1
grep -E "eager|graph replay|CAPTURE|Error|error" $O/cfg5_graph.log | head
timeout 200 python tools/graph_probe.py --dtype bfloat16 --steps 20 < /dev/null > $O/bf16_graph.log 2>&1
grep -E "eager|graph replay|CAPTURE|Error|error" $O/bf16_graph.log | head
timeout 200 python tools/graph_probe.py --dtype float32 --steps 20 < /dev/null > $O/f32_graph.log 2>&1
grep -E "eager|graph replay|CAPTURE|Error|error" $O/f32_graph.log | head
