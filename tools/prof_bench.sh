# rocprofv3 kernel-trace summary of the default bench (config 3); keeps only the stats CSVs
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_bench
rm -rf $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-extras > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1
rm -f $OUT/r_kernel_trace.csv $OUT/*.db
tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log | cut -c1-160
