# the supernodal host comparator in the c2 / c5 lines:  bash tools/r04_cpu_mt.sh <tag>
TAG=${1:-r04_mt}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
nproc
for wl in ${2:-c2 c5}; do
  SECONDS=0; timeout 900 python bench.py --workload $wl --steps 5 --warmup 2 > $O/${TAG}_$wl.json 2> $O/${TAG}_$wl.err
  echo "wall ${SECONDS}s"; tail -2 $O/${TAG}_$wl.err
  python - $O/${TAG}_$wl.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[1].split("/")[-1], "it/s", d["value"], "ms", d["ms_per_step"], "parity", (d.get("parity") or {}).get("rel_err_vs_oracle"))
    print("  cpu_baseline", json.dumps(d.get("cpu_baseline"))[:300])
    print("  cpu_baseline_mt", json.dumps(d.get("cpu_baseline_mt"))[:900])
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
