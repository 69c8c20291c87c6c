# round 5: where does the HOST time of the fp32 step go (host enqueue 36.6 of 37.6 ms)?  and: cudnn.benchmark with the shipped find-db
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; O=gpurun_out/r05k; mkdir -p $O; export MIOPEN_LOG_LEVEL=1
timeout 200 python tools/host_profile.py --steps 10 < /dev/null > $O/host_profile.log 2>&1; head -70 $O/host_profile.log | cut -c1-180
B="python bench.py --steps 40 --warmup 12 --no-cpu-baseline --no-kernel-timing"
timeout 120 $B < /dev/null > $O/plain.log 2>&1; grep -E "^\{" $O/plain.log | tail -1 | cut -c1-330
timeout 400 $B --miopen-search < /dev/null > $O/search.log 2>&1; grep -E "^\{" $O/search.log | tail -1 | cut -c1-330; tail -3 $O/search.log | cut -c1-200
