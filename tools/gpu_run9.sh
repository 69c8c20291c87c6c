set -x
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_dcn -o dcn -- python tools/opbench.py --iters 5 --only dcn > gpurun_out/prof_dcn.log 2>&1
head -30 gpurun_out/prof_dcn/dcn_kernel_stats.csv | cut -c1-220
find gpurun_out/prof_dcn -name "*kernel_trace.csv" -delete
