# bundle size (CHIP_BUNDLE_MAX_WORK) on config 2, same box:  bash tools/r05_bmw.sh <tag>
TAG=${1:-r05_bmw}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
for g in 0 500000 250000 2000000 0; do
  CHIP_BUNDLE_MAX_WORK=$g timeout 600 python bench.py --workload c2 --cpu-steps 0 --no-extras --steps 20 --warmup 3 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('max_work %-8s ms/step %.4f  it/s %.2f  setup %s nnzL %s' % ('$g', d['ms_per_step'], d['value'], d['config'].get('setup_s'), d['config'].get('nnz_L')))
" | tee -a $O/${TAG}_c2.txt
done
