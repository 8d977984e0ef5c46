cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --pmc $C --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$C -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-extras > $GRAFT_REPO_ROOT/gpurun_out/pmc_$C.log 2>&1
ls $GRAFT_REPO_ROOT/gpurun_out/pmc_$C
done
