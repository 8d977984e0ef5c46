# config 5 check:  bash tools/r04_c5.sh <tag> "<pytest -k expression>" [env assignments for the bench run]
TAG=${1:-r04_c5}
KEXPR=${2:-assembled or chain_supernodes or c5_24 or spread}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
if [ "$KEXPR" != "none" ]; then timeout 900 python -m pytest tests -m gpu -x -q -k "$KEXPR" > $O/${TAG}_pytest.log 2>&1; tail -8 $O/${TAG}_pytest.log | cut -c1-220; fi
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof -o c5 -- python $R/bench.py --workload c5 --cpu-steps 0 --steps 5 --warmup 2 > $O/${TAG}_c5.json 2> $O/${TAG}_c5.err
cd $R
python - $O/${TAG}_c5.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    r = d.get("roofline") or {}
    print("c5 it/s", d["value"], "ms", d["ms_per_step"], "setup", d["config"].get("setup_s"), "frac", r.get("frac"), "parity", (d.get("parity") or {}).get("rel_err_vs_oracle"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
tail -2 $O/${TAG}_c5.err
f=$(find $O/${TAG}_prof -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp $f $O/${TAG}_c5_kernel_stats.csv && head -14 $f | cut -c1-60,100-400 | cut -c1-170; rm -rf $O/${TAG}_prof
