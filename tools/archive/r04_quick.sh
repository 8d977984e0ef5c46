# quick GPU check:  bash tools/r04_quick.sh <tag> "<pytest -k expression>"
TAG=${1:-r04_q}
KEXPR=${2:-grouped_fold or fused or c4 or update_scaled}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q -k "$KEXPR" > $O/${TAG}_pytest.log 2>&1
tail -12 $O/${TAG}_pytest.log | cut -c1-220
timeout 300 python bench.py --workload c4 --nbatch 128 --cpu-steps 0 > $O/${TAG}_c4_128.json 2> $O/${TAG}_c4_128.err
timeout 300 python bench.py --workload c3 --cpu-steps 0 > $O/${TAG}_c3.json 2> $O/${TAG}_c3.err
for f in $O/${TAG}_c4_128.json $O/${TAG}_c3.json; do python - $f <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    r = d.get("roofline") or {}
    print(sys.argv[1].split("/")[-1], "it/s", d["value"], "ms", d["ms_per_step"], "ir_us", r.get("avg_launch_us"), "parity", (d.get("parity") or {}).get("rel_err_vs_oracle"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
tail -2 $O/${TAG}_c3.err
