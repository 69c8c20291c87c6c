set -x
mkdir -p gpurun_out
python -c "import torch; print(torch.cuda.get_device_name(0))" > gpurun_out/dev.log 2>&1
python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
python tools/opbench.py --iters 30 --json gpurun_out/opbench.json > gpurun_out/opbench.log 2>&1
R=$PWD; cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o opbench -- python $R/tools/opbench.py --iters 10 > $R/gpurun_out/rocprof.log 2>&1
cd $R; tail -5 gpurun_out/pytest_gpu.log; tail -50 gpurun_out/opbench.log
