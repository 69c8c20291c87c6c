O=gpurun_out/r06w; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 python tools/gpu/nms_probe.py < /dev/null > $O/nms_probe.txt 2>&1; cat $O/nms_probe.txt | tail -12
