# Round-4 GPU check of the grouped-fold step kernels:  bash tools/r04_step.sh <tag> [full]
TAG=${1:-r04_b}
FULL=$2
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
if [ -n "$FULL" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q > $O/${TAG}_pytest.log 2>&1
else
  timeout 600 python -m pytest tests -m gpu -x -q -k "grouped_fold or fused or c4 or repeated_solve or device_resident or bench_sharded" > $O/${TAG}_pytest.log 2>&1
fi
tail -15 $O/${TAG}_pytest.log
timeout 300 python bench.py --workload c4 --nbatch 128 --cpu-steps 0 > $O/${TAG}_c4_128.json 2> $O/${TAG}_c4_128.err
CHIP_NO_STEP_KERNEL=1 timeout 300 python bench.py --workload c4 --nbatch 128 --no-extras > $O/${TAG}_c4_128_old.json 2> $O/${TAG}_c4_128_old.err
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
tail -3 $O/${TAG}_c4_128.err
head -32 $O/${TAG}_c4_128_skew.txt
python - <<'PY'
import csv, re, os
p = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "%s_c4_128_kernel_stats.csv" % os.environ.get("TAGX", ""))
PY
python - $O/${TAG}_c4_128_kernel_stats.csv <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(k_[a-z_0-9A-Z]+|__amd[a-zA-Z_]+)", r["Name"])
    print("%-30s calls %4s avg %8.1f us" % (m.group(1) if m else r["Name"][:30], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
