# HBM traffic per kernel from two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE; no other trace domains) of a short
# bench.py run.  usage (on the GPU box):  bash tools/pmc_traffic.sh [workload: auto | c2 | c5 | c4] [extra bench args]
# then:  python tools/pmc_summarize.py gpurun_out profiles/<tag>_pmc_traffic[_<workload>].json <workload>
WL=${1:-auto}
shift
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_$C
timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$C -o p -- python $GRAFT_REPO_ROOT/bench.py --workload $WL --steps 2 --warmup 1 --no-extras --cpu-steps 0 "$@" > $GRAFT_REPO_ROOT/gpurun_out/pmc_$C.log 2>&1
ls $GRAFT_REPO_ROOT/gpurun_out/pmc_$C
done
