T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
mkdir -p gpurun_out; export MIOPEN_LOG_LEVEL=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P=/tmp/r03n_prof; rm -rf $P
timeout 300 rocprofv3 --kernel-trace --stats -d $P -o bdp -- python bench.py --steps 8 --warmup 6 --no-cpu-baseline --no-kernel-timing --force-ddp > gpurun_out/r03n_prof.log 2>&1 < /dev/null; el "rocprof rc=$?"
tail -3 gpurun_out/r03n_prof.log
f=$(find $P -name "*kernel_stats.csv" | head -1)
t=$(find $P -name "*kernel_trace.csv" | head -1)
echo "stats=$f trace=$t"
if [ -n "$f" ]; then cp "$f" gpurun_out/r03n_bdp_kernel_stats.csv; grep -i "ccl\|allreduce\|OneRank\|multi_tensor\|fused_sgd\|foreach" "$f" | cut -c1-200 | head -12; fi
if [ -n "$t" ]; then head -1 "$t" > gpurun_out/r03n_bdp_trace_tail.csv; tail -n 9000 "$t" >> gpurun_out/r03n_bdp_trace_tail.csv; fi
el done
