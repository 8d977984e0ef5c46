# raw dispatch list (start, duration, gap to the previous END on the device, kernel) of the last step of a workload:
#   bash tools/r06_trace_raw.sh <tag> <workload> [ENV=..]     (overlapping dispatches show as negative gaps)
TAG=$1; W=$2; shift 2
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
(cd /tmp && export TMPDIR=/tmp && env "$@" timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/raw_$TAG -o r -- python $R/bench.py --workload $W --cpu-steps 0 --steps 3 --warmup 1 --no-extras > $O/raw_$TAG.log 2>&1)
python $R/tools/trace_step.py $O/raw_$TAG/r_kernel_trace.csv k_scatter_init 2>&1 | sed -n '/^dispatch by dispatch/,$p' > $O/${TAG}_raw_$W.txt
rm -rf $O/raw_$TAG
