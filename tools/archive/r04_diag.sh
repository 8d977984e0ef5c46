# Round-4 diagnostics in one GPU call:  bash tools/r04_diag.sh <tag>
#   <tag>_pytest.log                 the -m gpu suite
#   <tag>_c4_<nb>.json               bench lines of a rank's share of config 4
#   <tag>_c4_128_skew.txt            per-phase stamps of every workgroup of the fused launch (tools/ir_skew.py)
#   <tag>_c4_128_kernel_stats.csv    rocprofv3 kernel summary of the 128-tree share
TAG=${1:-r04_a}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/${TAG}_pytest.log 2>&1
tail -3 $O/${TAG}_pytest.log
for nb in 128 256; do
  timeout 300 python bench.py --workload c4 --nbatch $nb --cpu-steps 0 > $O/${TAG}_c4_$nb.json 2> $O/${TAG}_c4_$nb.err
done
CHIP_IR_DEBUG=2 CHIP_IR_DEBUG_FILE=$O/${TAG}_stamps.bin timeout 300 python bench.py --workload c4 --nbatch 128 --no-extras --steps 2 --warmup 1 > /dev/null 2> $O/${TAG}_stamps.err
python tools/ir_skew.py $O/${TAG}_stamps.bin > $O/${TAG}_c4_128_skew.txt 2>&1
rm -f $O/${TAG}_stamps.bin
bash tools/prof_bench.sh ${TAG}_c4_128 --workload c4 --nbatch 128 --steps 10 --warmup 2 > /dev/null 2>&1
cp $O/prof_${TAG}_c4_128/r_kernel_stats.csv $O/${TAG}_c4_128_kernel_stats.csv
for f in $O/${TAG}_c4_*.json; do python - $f <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    r = d.get("roofline") or {}
    print(sys.argv[1].split("/")[-1], "ms", d["ms_per_step"], "ir_us", r.get("avg_launch_us"), "parity", (d.get("parity") or {}).get("rel_err_vs_oracle"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
head -30 $O/${TAG}_c4_128_skew.txt
