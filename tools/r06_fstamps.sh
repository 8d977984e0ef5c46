# stamps of every workgroup of the bundle factorisation on config 3:  bash tools/r06_fstamps.sh <tag> [ENV=VAL ...]
TAG=$1; shift 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
env "$@" CHIP_IR_DEBUG=3 CHIP_IR_DEBUG_FILE=$O/${TAG}_stamps.bin timeout 300 python bench.py --workload c3 --no-extras --cpu-steps 0 --steps 2 --warmup 1 > /dev/null 2> $O/${TAG}_stamps.err
python tools/ir_skew.py $O/${TAG}_stamps.bin.factor > $O/${TAG}_c3_factor_skew.txt 2>&1
rm -f $O/${TAG}_stamps.bin $O/${TAG}_stamps.bin.factor
head -22 $O/${TAG}_c3_factor_skew.txt
