# GPU check of round 5:  bash tools/r05_quick.sh <tag> "<pytest -k expression or ALL>" "<workloads: c2 c3 c5 c4_128 ...>" [prof workloads]
TAG=${1:-r05_q}
KEXPR=${2:-ALL}
WLS=${3:-c2}
PROFS=${4:-}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
if [ "$KEXPR" = "ALL" ]; then
  timeout 1200 python -m pytest tests -m gpu -q > $O/${TAG}_pytest.log 2>&1
else
  timeout 1200 python -m pytest tests -m gpu -q -k "$KEXPR" > $O/${TAG}_pytest.log 2>&1
fi
grep -E "^(FAILED|ERROR)|passed|failed" $O/${TAG}_pytest.log | tail -25 | cut -c1-250
for w in $WLS; do
  case $w in
    c4_*) extra="--workload c4 --nbatch ${w#c4_}";;
    *) extra="--workload $w";;
  esac
  timeout 600 python bench.py $extra --cpu-steps 0 > $O/${TAG}_bench_$w.json 2> $O/${TAG}_bench_$w.err
  python - $O/${TAG}_bench_$w.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    r = d.get("roofline") or {}
    print(sys.argv[1].split("/")[-1], "it/s", d["value"], "ms", d["ms_per_step"], "kernel", r.get("kernel"), "us", r.get("avg_launch_us"), "frac", r.get("frac"),
          "parity", (d.get("parity") or {}).get("rel_err_vs_oracle"), "setup_s", d.get("setup_s"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
  tail -2 $O/${TAG}_bench_$w.err | cut -c1-300
done
for w in $PROFS; do
  case $w in
    c4_*) extra="--workload c4 --nbatch ${w#c4_}";;
    *) extra="--workload $w";;
  esac
  (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_$w -o r -- python $R/bench.py $extra --cpu-steps 0 --steps 6 --warmup 1 --no-extras > $O/${TAG}_prof_$w.log 2>&1)
  f=$(ls $O/${TAG}_prof_$w/*kernel_stats.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then cp $f $O/${TAG}_${w}_kernel_stats.csv; head -16 $f | cut -c1-200; fi
  rm -rf $O/${TAG}_prof_$w
done
