# rocprofv3 kernel-trace summary of a bench.py run; keeps only the stats CSVs
# usage (on the GPU box): bash tools/prof_bench.sh [tag] [bench.py arguments]      (default: config 3)
cd /tmp && export TMPDIR=/tmp
TAG=${1:-bench}
shift
ARGS=${@:---steps 10 --warmup 2}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
rm -rf $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o r -- python $GRAFT_REPO_ROOT/bench.py --no-extras $ARGS > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log 2>&1
rm -f $OUT/r_kernel_trace.csv $OUT/*.db
tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log | cut -c1-160
