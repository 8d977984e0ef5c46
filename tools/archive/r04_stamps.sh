# stamps of every workgroup of the step kernels on the 128-tree share of config 4:  bash tools/r04_stamps.sh <tag>
TAG=${1:-r04_e}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
CHIP_IR_DEBUG=3 CHIP_IR_DEBUG_FILE=$O/${TAG}_stamps.bin timeout 300 python bench.py --workload c4 --nbatch 128 --no-extras --steps 2 --warmup 1 > /dev/null 2> $O/${TAG}_stamps.err
python tools/ir_skew.py $O/${TAG}_stamps.bin > $O/${TAG}_solve_skew.txt 2>&1
python tools/ir_skew.py $O/${TAG}_stamps.bin.factor > $O/${TAG}_factor_skew.txt 2>&1
rm -f $O/${TAG}_stamps.bin $O/${TAG}_stamps.bin.factor
head -24 $O/${TAG}_solve_skew.txt
head -18 $O/${TAG}_factor_skew.txt
