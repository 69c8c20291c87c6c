cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 200 python tools/gpu/cfg1_bwd.py 0,16,32,64 50 2>&1 | grep cfg1
for d in 1 32 33; do echo "== debug=$d"; DETOPS_TUNING=roi_bwd_debug=$d timeout 100 python tools/gpu/cfg1_bwd.py 32 40 2>&1 | grep "7x7 sr2"; done
timeout 600 python -m pytest tests/test_ops_gpu.py -q -p no:cacheprovider -k "roi_align_backward" < /dev/null 2>&1 | tail -2
