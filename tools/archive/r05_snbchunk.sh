# chunk size of the bundle contributions into supernode members, config 2 (and 5), same box:  bash tools/r05_snbchunk.sh <tag>
TAG=${1:-r05_sc}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
for w in c2 c5; do
for g in 0 4096 16384 65536 0; do
  [ $w = c5 ] && [ $g = 4096 ] && continue
  CHIP_SNB_CHUNK=$g timeout 600 python bench.py --workload $w --cpu-steps 0 --no-extras --steps 20 --warmup 3 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$w chunk %-6s ms/step %.4f  it/s %.2f  parity %s step_ms %s' % ('$g', d['ms_per_step'], d['value'], (d.get('parity') or {}).get('rel_err_vs_oracle'), d.get('step_ms')))
" | tee -a $O/${TAG}.txt
done
done
