# dispatch-order listing of one step:  bash tools/r05_step.sh <tag> <workload> [ENV=..]
TAG=$1; W=$2; shift 2
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
case $W in
  c4_*) extra="--workload c4 --nbatch ${W#c4_}";;
  *) extra="--workload $W";;
esac
(cd /tmp && export TMPDIR=/tmp && env "$@" timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/step_$TAG -o r -- python $R/bench.py $extra --cpu-steps 0 --steps 4 --warmup 1 --no-extras > $O/step_$TAG.log 2>&1)
python $R/tools/trace_step.py $O/step_$TAG/r_kernel_trace.csv $MARKER > $O/${TAG}_step_$W.txt 2>&1
head -5 $O/${TAG}_step_$W.txt | cut -c1-200
rm -rf $O/step_$TAG
