# PMC traffic (FETCH_SIZE / WRITE_SIZE passes) of the profiled workloads -> gpurun_out/<tag>_pmc_traffic*.json
# usage: bash tools/r05_pmc_all.sh <tag> "<workloads: auto c2 c5 c4_128>"
TAG=${1:-r05_zz}
WLS=${2:-auto c2 c5 c4_128}
R=$GRAFT_REPO_ROOT
for wl in $WLS; do
  case $wl in
    c4_*) bash $R/tools/pmc_traffic.sh c4 --nbatch ${wl#c4_} > /dev/null 2>&1;;
    *) bash $R/tools/pmc_traffic.sh $wl > /dev/null 2>&1;;
  esac
  sfx=""; [ $wl != auto ] && sfx="_$wl"
  python $R/tools/pmc_summarize.py $R/gpurun_out $R/gpurun_out/${TAG}_pmc_traffic$sfx.json $wl | head -8
  rm -rf $R/gpurun_out/pmc_FETCH_SIZE $R/gpurun_out/pmc_WRITE_SIZE
done
