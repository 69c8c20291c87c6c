cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for d in 0 1 2 3 4 7; do echo "== debug=$d"; DETOPS_TUNING=roi_bwd_debug=$d timeout 100 python tools/gpu/cfg1_bwd.py 16,32,64 40 2>&1 | grep "7x7 sr2"; done
