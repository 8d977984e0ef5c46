# rocprofv3 kernel summaries of c3 and the 128-tree share:  bash tools/r04_prof.sh <tag>
TAG=${1:-r04_p}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
bash tools/prof_bench.sh ${TAG}_c3 --workload c3 --steps 10 --warmup 2 > /dev/null 2>&1
cp $O/prof_${TAG}_c3/r_kernel_stats.csv $O/${TAG}_c3_kernel_stats.csv
bash tools/prof_bench.sh ${TAG}_c4_128 --workload c4 --nbatch 128 --steps 10 --warmup 2 > /dev/null 2>&1
cp $O/prof_${TAG}_c4_128/r_kernel_stats.csv $O/${TAG}_c4_128_kernel_stats.csv
for f in $O/${TAG}_c3_kernel_stats.csv $O/${TAG}_c4_128_kernel_stats.csv; do echo $f; python - $f <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(k_[a-z_0-9A-Z]+|__amd[a-zA-Z_]+)", r["Name"])
    print("%-30s calls %4s avg %8.1f us" % (m.group(1) if m else r["Name"][:30], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
tail -1 $O/prof_${TAG}_c3.log | cut -c1-200
