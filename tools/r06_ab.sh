# same-box A/B on config 3:  bash tools/r06_ab.sh <tag> "ENV=VAL[,ENV2=VAL2] ..." [workload]   (each setting twice, interleaved)
# AB_PARITY=1: the runs keep their parity leg (config 5: against the committed fixture, seconds; configs 2 / 3: one oracle solve) --
# a variant must be RIGHT before its time is believed
TAG=$1; SETS=$2; W=${3:-c3}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
for rep in 1 2; do
for st in $SETS; do
  e=$(echo $st | tr ',' ' ')
  if [ "$st" = "default" ]; then e=""; fi
  env $e timeout 600 python bench.py --workload $W --cpu-steps 0 $([ -n "$AB_PARITY" ] || echo --no-extras) --steps 20 --warmup 3 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d.get('roofline') or {}
p=d.get('parity') or {}
print('%-40s ms/step %.4f  it/s %8.2f  launch_us %s  parity %s %s' % ('$st', d['ms_per_step'], d['value'], r.get('avg_launch_us'), p.get('ok'), p.get('rel_err_vs_oracle')))
" | tee -a $O/${TAG}_ab.txt
done
done
