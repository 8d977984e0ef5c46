# per-dispatch summary (tools/trace_summary.py) of chosen kernels of a bench.py run
# usage (on the GPU box): bash tools/prof_trace.sh <tag> "<kernel name substrings>" [bench.py arguments]
cd /tmp && export TMPDIR=/tmp
TAG=$1
PATS=$2
shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_$TAG
rm -rf $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o r -- python $GRAFT_REPO_ROOT/bench.py --no-extras "$@" > $GRAFT_REPO_ROOT/gpurun_out/trace_$TAG.log 2>&1
python $GRAFT_REPO_ROOT/tools/trace_summary.py $OUT/r_kernel_trace.csv $PATS > $GRAFT_REPO_ROOT/gpurun_out/trace_$TAG.txt 2>&1
cp $OUT/r_kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/trace_${TAG}_kernel_stats.csv
rm -rf $OUT
