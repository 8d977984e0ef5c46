# One round's evidence in one call (on the GPU box):  bash tools/prof_round.sh r03_x [quick]
#   <tag>_bench.json                     bench.py default (config 3 + parity + cpu rows + batched config 4)
#   <tag>_kernel_stats.csv               rocprofv3 --kernel-trace --stats of the same command (--no-extras)
#   <tag>_pmc_traffic.json               two --pmc passes (FETCH_SIZE, WRITE_SIZE), summarised per kernel
#   <tag>_bench_c2 / c4_128 / c5 .json   the other workloads' lines; kernel stats of c2 and c5 (not with `quick`)
TAG=${1:-r03}
QUICK=$2
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
bash tools/prof_bench.sh $TAG > /dev/null 2>&1
cp $O/prof_$TAG/r_kernel_stats.csv $O/${TAG}_kernel_stats.csv
bash tools/pmc_traffic.sh > /dev/null 2>&1
python tools/pmc_summarize.py $O $O/${TAG}_pmc_traffic.json > $O/${TAG}_pmc_traffic.txt 2>&1
find $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE -name "*.csv" -size +1M -delete 2>/dev/null
rm -f $O/pmc_*/*.db $O/pmc_*/*/*.db 2>/dev/null
python bench.py --workload c4 --nbatch 128 --cpu-steps 0 > $O/${TAG}_bench_c4_128.json 2>> $O/${TAG}_bench.err
python bench.py --workload c4 --nbatch 256 --cpu-steps 0 > $O/${TAG}_bench_c4_256.json 2>> $O/${TAG}_bench.err
if [ -z "$QUICK" ]; then
python bench.py --workload c2 > $O/${TAG}_bench_c2.json 2>> $O/${TAG}_bench.err
python bench.py --workload c5 --steps 10 > $O/${TAG}_bench_c5.json 2>> $O/${TAG}_bench.err
bash tools/prof_bench.sh ${TAG}_c2 --workload c2 --steps 5 --warmup 2 > /dev/null 2>&1
cp $O/prof_${TAG}_c2/r_kernel_stats.csv $O/${TAG}_c2_kernel_stats.csv
bash tools/prof_bench.sh ${TAG}_c5 --workload c5 --steps 5 --warmup 2 > /dev/null 2>&1
cp $O/prof_${TAG}_c5/r_kernel_stats.csv $O/${TAG}_c5_kernel_stats.csv
fi
for f in $O/${TAG}_bench*.json; do python - $f <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    r = d.get("roofline") or {}
    print(sys.argv[1].split("/")[-1], "value", d["value"], "ms", d["ms_per_step"], "frac", r.get("frac"), "parity", (d.get("parity") or {}).get("rel_err_vs_oracle"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
