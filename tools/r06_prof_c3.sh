# round 6 evidence for config 3 in one call:  bash tools/r06_prof_c3.sh <tag>
#   <tag>_bench_c3.json, <tag>_c3_kernel_stats.csv (rocprofv3 --kernel-trace --stats), <tag>_pmc_traffic.json (two --pmc passes),
#   <tag>_step_c3_dispatch_order.txt, <tag>_c3_ir_skew.txt (stamps of every workgroup of the fused solve)
TAG=${1:-r06_x}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
python bench.py --workload c3 --cpu-steps 0 --no-extras > $O/${TAG}_bench_c3.json 2> $O/${TAG}_bench_c3.err
bash tools/prof_bench.sh ${TAG}_c3 --workload c3 --steps 10 --warmup 2 > /dev/null 2>&1
cp $O/prof_${TAG}_c3/r_kernel_stats.csv $O/${TAG}_c3_kernel_stats.csv
bash tools/pmc_traffic.sh c3 > /dev/null 2>&1
python tools/pmc_summarize.py $O $O/${TAG}_pmc_traffic.json auto > $O/${TAG}_pmc_traffic.txt 2>&1
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/prof_${TAG}_c3
bash tools/r06_step.sh ${TAG} > /dev/null 2>&1
bash tools/r06_stamps.sh ${TAG} > /dev/null 2>&1
cat $O/${TAG}_pmc_traffic.txt
head -12 $O/${TAG}_c3_kernel_stats.csv | cut -c1-150
python -c "
import json; d=json.load(open('$O/${TAG}_bench_c3.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], r['avg_launch_us'], r['frac'], r.get('traffic'), r.get('traffic_profiled_in'))"
