# Round 6 (late): kernel traces of the mixed-precision steps on the final tree — where the small launches are
O=gpurun_out/r06trace2; mkdir -p $O; export MIOPEN_LOG_LEVEL=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CFG5="--config e2e_mask_rcnn_R_101_FPN_1x.yaml --dtype float16"
tr() { N=$1; shift; P=/tmp/prof_$N; rm -rf $P
  timeout 500 rocprofv3 --kernel-trace --output-format csv -d $P -o bench -- python bench.py --steps 6 --warmup 6 --no-cpu-baseline --no-kernel-timing "$@" < /dev/null > $O/$N.log 2>&1
  T=$(find $P -name "*kernel_trace.csv" | head -1)
  python tools/trace_steps.py "$T" 4 70 > $O/${N}_step_breakdown.txt 2>&1; head -3 $O/${N}_step_breakdown.txt | cut -c1-160; }
tr cfg5 $CFG5 MODEL.RESNETS.STAGE_WITH_DCN "(False,True,True,True)"
tr bf16 --dtype bfloat16
