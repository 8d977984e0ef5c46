# round 6: config 3 line + stamps of every workgroup of k_bundle_ir (CHIP_IR_DEBUG=2):  bash tools/r06_base.sh <tag>
TAG=${1:-r06_a}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 600 python bench.py --workload c3 --cpu-steps 0 --no-extras > $O/${TAG}_bench_c3.json 2> $O/${TAG}_bench_c3.err
python - $O/${TAG}_bench_c3.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d.get("roofline") or {}
print("it/s", d["value"], "ms", d["ms_per_step"], "kernel", r.get("kernel"), "us", r.get("avg_launch_us"), "frac", r.get("frac"), "parity", (d.get("parity") or {}).get("rel_err_vs_oracle"))
PY
CHIP_IR_DEBUG=2 CHIP_IR_DEBUG_FILE=$O/${TAG}_stamps.bin timeout 300 python bench.py --workload c3 --no-extras --cpu-steps 0 --steps 2 --warmup 1 > /dev/null 2> $O/${TAG}_stamps.err
python tools/ir_skew.py $O/${TAG}_stamps.bin > $O/${TAG}_c3_ir_skew.txt 2>&1
rm -f $O/${TAG}_stamps.bin
head -64 $O/${TAG}_c3_ir_skew.txt | tail -42
