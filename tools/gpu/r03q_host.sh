cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export MIOPEN_LOG_LEVEL=1
timeout 400 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "dcn or deform or frozen or cfg5" < /dev/null 2>&1 | tail -4
timeout 500 python tools/host_profile.py --config e2e_mask_rcnn_R_101_FPN_1x.yaml --dtype float16 --steps 10 MODEL.RESNETS.STAGE_WITH_DCN "(False, True, True, True)" < /dev/null > gpurun_out/r03q_host_cfg5.txt 2>&1
grep -v "amdgpu.ids" gpurun_out/r03q_host_cfg5.txt | head -12 | cut -c1-170
