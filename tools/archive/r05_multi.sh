# bench lines of several workloads (no tests):  bash tools/r05_multi.sh <tag> "<workloads>" ["ENV=1 ENV2=1"]
TAG=${1:-r05_m}
WLS=${2:-c3}
ENVS=${3:-}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
for w in $WLS; do
  case $w in
    c4_*) extra="--workload c4 --nbatch ${w#c4_}";;
    *) extra="--workload $w";;
  esac
  env $ENVS timeout 900 python bench.py $extra --cpu-steps 0 --no-extras > $O/${TAG}_bench_$w.json 2> $O/${TAG}_bench_$w.err
  python - $O/${TAG}_bench_$w.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    r = d.get("roofline") or {}
    print(sys.argv[1].split("/")[-1], "it/s", d["value"], "ms", d["ms_per_step"], "kernel", (r.get("kernel") or "")[:30], "us", r.get("avg_launch_us"), "frac", r.get("frac"), "setup_s", d.get("setup_s"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
  tail -2 $O/${TAG}_bench_$w.err | cut -c1-300
done
