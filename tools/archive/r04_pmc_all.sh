# PMC traffic of the three profiled workloads, summarised into gpurun_out/<tag>_pmc_traffic*.json
TAG=${1:-r04_zz}
R=$GRAFT_REPO_ROOT
for wl in auto c2 c5; do
  bash $R/tools/pmc_traffic.sh $wl > /dev/null 2>&1
  sfx=""; [ $wl != auto ] && sfx="_$wl"
  python $R/tools/pmc_summarize.py $R/gpurun_out $R/gpurun_out/${TAG}_pmc_traffic$sfx.json $wl | head -6
  rm -rf $R/gpurun_out/pmc_FETCH_SIZE $R/gpurun_out/pmc_WRITE_SIZE
done
