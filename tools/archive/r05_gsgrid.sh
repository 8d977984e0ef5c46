# grid size of the persistent sweeps on config 2, same box:  bash tools/r05_gsgrid.sh <tag>
TAG=${1:-r05_gsg}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
for g in 0 32 64 96 192 0; do
  CHIP_GSWEEP_GRID=$g timeout 600 python bench.py --workload c2 --cpu-steps 0 --no-extras --steps 20 --warmup 3 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('grid %-6s ms/step %.4f  it/s %.2f  step_ms %s' % ('$g', d['ms_per_step'], d['value'], d.get('step_ms')))
" | tee -a $O/${TAG}_c2.txt
done
