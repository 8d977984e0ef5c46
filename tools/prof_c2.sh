cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_c2
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o r -- python $GRAFT_REPO_ROOT/tools/scale_check.py c2 > $GRAFT_REPO_ROOT/gpurun_out/prof_c2.log 2>&1
rm -f $OUT/r_kernel_trace.csv $OUT/*.db
tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_c2.log | cut -c1-300
