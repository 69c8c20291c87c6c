# r02d: binned ROIAlign backward (v3) — full GPU parity suite, opbench roi_align, kernel stats, PMC traffic, bench
set -x
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
mkdir -p gpurun_out
export MIOPEN_LOG_LEVEL=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -8 gpurun_out/pytest_gpu.log | cut -c1-220; el pytest
timeout 200 python tools/opbench.py --iters 30 --only roi_align --json gpurun_out/opbench_c.json > gpurun_out/opbench_c.log 2>&1
grep -v "^/opt" gpurun_out/opbench_c.log | grep "roi_align" | cut -c1-200; el opbench
rm -rf gpurun_out/prof_bwd gpurun_out/pmc_fetch gpurun_out/pmc_write
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bwd -o bwd -- python tools/opbench.py --iters 20 --only roi_align_fpn > gpurun_out/prof_bwd.log 2>&1
timeout 90 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o x -- python tools/opbench.py --iters 3 --only roi_align_fpn > gpurun_out/pmc_fetch.log 2>&1
timeout 90 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o x -- python tools/opbench.py --iters 3 --only roi_align_fpn > gpurun_out/pmc_write.log 2>&1
python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/traffic.json 2>&1 | cut -c1-200
find gpurun_out/pmc_fetch gpurun_out/pmc_write -name "*kernel_trace.csv" -delete; el pmc
timeout 400 python bench.py > gpurun_out/bench_f32.log 2>&1; grep -E "^\{" gpurun_out/bench_f32.log | cut -c1-2500; el bench
