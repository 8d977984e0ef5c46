# same-box A/B with arbitrary env / args:  bash tools/r05_ab2.sh <tag> <workload> "<extra args>" "ENVSET1" "ENVSET2" ...   (ENVSET: "A=1 B=2" or "-")
TAG=$1; W=$2; XARGS=$3; shift 3
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
case $W in
  c4_*) extra="--workload c4 --nbatch ${W#c4_}";;
  *) extra="--workload $W";;
esac
for es in "$@"; do
  if [ "$es" = "-" ]; then e=""; else e="$es"; fi
  env $e timeout 600 python bench.py $extra --cpu-steps 0 --no-extras --steps 20 --warmup 3 $XARGS 2> $O/${TAG}_ab2.err | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1])
    print('%-40s ms/step %.4f  it/s %.2f  step_ms %s' % ('$es', d['ms_per_step'], d['value'], d.get('step_ms')))
except Exception as ex:
    print('$es', 'ERR', ex)
" | tee -a $O/${TAG}_ab2_$W.txt
  tail -1 $O/${TAG}_ab2.err | cut -c1-200
done
