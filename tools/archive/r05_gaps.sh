# idle-time analysis of one workload:  bash tools/r05_gaps.sh <tag> <workload> [ENV=..]
TAG=$1; W=$2; shift 2
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
case $W in
  c4_*) extra="--workload c4 --nbatch ${W#c4_}";;
  *) extra="--workload $W";;
esac
(cd /tmp && export TMPDIR=/tmp && env "$@" timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/gaps_$TAG -o r -- python $R/bench.py $extra --cpu-steps 0 --steps 6 --warmup 1 --no-extras > $O/gaps_$TAG.log 2>&1)
python $R/tools/trace_gaps.py $O/gaps_$TAG/r_kernel_trace.csv > $O/${TAG}_gaps_$W.txt 2>&1
cat $O/${TAG}_gaps_$W.txt | cut -c1-250
ls $O/gaps_$TAG | head
rm -rf $O/gaps_$TAG
