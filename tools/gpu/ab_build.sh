# Build the library of another commit next to the current one, for same-box A/B runs on the GPU:
#   bash tools/gpu/ab_build.sh [commit=HEAD]   ->  tools/gpu/ab/libdetops_base.so   (git-ignored, travels with gpurun)
#   on the box:  DETOPS_LIB_PATH=$GRAFT_REPO_ROOT/tools/gpu/ab/libdetops_base.so python tools/opbench.py ...
set -e
C=${1:-HEAD}; R=$(git rev-parse --show-toplevel); T=$(mktemp -d)
git -C $R archive $C maskrcnn-benchmark_amd/csrc include | tar -x -C $T
mkdir -p $T/maskrcnn-benchmark_amd/maskrcnn_benchmark/lib $R/tools/gpu/ab
make -s -C $T/maskrcnn-benchmark_amd/csrc -j8 > /dev/null
cp $T/maskrcnn-benchmark_amd/maskrcnn_benchmark/lib/libdetops_gfx950.so $R/tools/gpu/ab/libdetops_base.so
rm -rf $T; echo "built $C -> tools/gpu/ab/libdetops_base.so"
