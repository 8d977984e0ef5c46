# rocprofv3 kernel-trace summary of tools/scale_check.py <args>; keeps only the stats CSVs
# usage (on the GPU box): bash tools/prof_scale.sh c5 --small
cd /tmp && export TMPDIR=/tmp
TAG=$(echo "$@" | tr -d ' -')
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
rm -rf $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o r -- python $GRAFT_REPO_ROOT/tools/scale_check.py "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log 2>&1
rm -f $OUT/r_kernel_trace.csv $OUT/*.db
tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log | cut -c1-400
