# dispatch-order listing of one config-3 step with the gaps between dispatches:  bash tools/r06_step.sh <tag> [ENV=..]
TAG=$1; shift 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
(cd /tmp && export TMPDIR=/tmp && env "$@" timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/step_$TAG -o r -- python $R/bench.py --workload c3 --cpu-steps 0 --steps 4 --warmup 1 --no-extras > $O/step_$TAG.log 2>&1)
python $R/tools/trace_step.py $O/step_$TAG/r_kernel_trace.csv k_sym_scale_write > $O/${TAG}_step_c3.txt 2>&1
cat $O/${TAG}_step_c3.txt | cut -c1-160
rm -rf $O/step_$TAG
