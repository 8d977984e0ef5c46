# same-box A/B of switches on one workload:  bash tools/r05_ab.sh <tag> <workload> "SWITCH1 SWITCH2 ..."   (each run: default + one switch set)
TAG=$1; W=$2; SWS=$3
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
case $W in
  c4_*) extra="--workload c4 --nbatch ${W#c4_}";;
  *) extra="--workload $W";;
esac
for sw in default $SWS default; do
  if [ "$sw" = "default" ]; then e=""; else e="$sw=1"; fi
  env $e timeout 600 python bench.py $extra --cpu-steps 0 --no-extras --steps 20 --warmup 3 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-28s ms/step %.4f  it/s %.2f  step_ms %s' % ('$sw', d['ms_per_step'], d['value'], d.get('step_ms')))
" | tee -a $O/${TAG}_ab_$W.txt
done
