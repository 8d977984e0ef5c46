# per-kernel average durations of config 3 under several settings, same box:  bash tools/r06_kstats.sh <tag> "ENV=V[,ENV2=V2] ..." [workload]
TAG=$1; SETS=$2; W=${3:-c3}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
for st in $SETS; do
  e=$(echo $st | tr ',' ' '); if [ "$st" = "default" ]; then e=""; fi
  (cd /tmp && export TMPDIR=/tmp && env $e timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_$TAG -o r -- python $R/bench.py --workload $W --cpu-steps 0 --steps 6 --warmup 1 --no-extras > $O/ks_$TAG.log 2>&1)
  echo "== $st"
  python - $O/ks_$TAG/r_kernel_stats.csv <<'PY'
import csv, sys, re
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    m = re.search(r"(k_\w+(<[^>]*>)?)", n)
    if float(r["Percentage"]) > 0.4: print("   %-34s calls %5s  avg %9.1f us  %5.1f %%" % ((m.group(1) if m else n)[:34], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
  rm -rf $O/ks_$TAG
done
