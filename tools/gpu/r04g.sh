cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for d in 0 8 16 32 33 35; do echo "== debug=$d"; DETOPS_TUNING=roi_bwd_debug=$d timeout 100 python tools/gpu/cfg1_bwd.py 32 40 2>&1 | grep "7x7 sr2"; done
