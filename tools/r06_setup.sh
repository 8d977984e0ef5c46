# setup time + step of a workload:  bash tools/r06_setup.sh "<workloads>"
R=$GRAFT_REPO_ROOT; cd $R
for w in $1; do
python bench.py --workload $w --cpu-steps 0 --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', 'ms/step', d['ms_per_step'], 'setup_s', d['config']['setup_s'])"
done
