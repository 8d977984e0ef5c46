# dispatch-order listing of one step of a workload (folded runs only):  bash tools/r06_step_w.sh <tag> <workload> [ENV=..]
TAG=$1; W=$2; shift 2
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
(cd /tmp && export TMPDIR=/tmp && env "$@" timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/step_$TAG -o r -- python $R/bench.py --workload $W --cpu-steps 0 --steps 3 --warmup 1 --no-extras > $O/step_$TAG.log 2>&1)
python $R/tools/trace_step.py $O/step_$TAG/r_kernel_trace.csv k_scatter_init 2>&1 | sed -n '/^dispatch by dispatch/q;p' > $O/${TAG}_step_$W.txt
rm -rf $O/step_$TAG
