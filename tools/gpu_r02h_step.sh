# step-level experiments: bench as is; NHWC weight-gradient igemm off; MIOpen search restricted to the pre-built (dynamic) kernels
mkdir -p gpurun_out
export MIOPEN_LOG_LEVEL=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T0=$(date +%s); el() { echo "[t+$(( $(date +%s) - T0 ))s] $*"; }
line() { grep -E "^\{" $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step','miopen')}, {k:round(v['mean_us'],1) for k,v in d.get('kernels',{}).items()})" 2>&1 | cut -c1-900; }
timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/bench_a.log 2>&1; line gpurun_out/bench_a.log; el base
MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_WRW_GTC_XDLOPS_NHWC=0 timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing > gpurun_out/bench_b.log 2>&1; line gpurun_out/bench_b.log; el no-nhwc-wrw
MIOPEN_FIND_MODE=5 timeout 420 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing --miopen-search > gpurun_out/bench_c.log 2>&1; line gpurun_out/bench_c.log; tail -3 gpurun_out/bench_c.log | cut -c1-300; el find-mode-5
