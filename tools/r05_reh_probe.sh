cd $GRAFT_REPO_ROOT
for args in "--coresident 8:60" "--coresident 8:1" "--force-comm --coresident 8:60" "--coresident 1:60" "--coresident 64:60"; do
  timeout 300 python bench.py --workload c4 --nbatch 128 $args --cpu-steps 0 --no-extras --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$args', d['ms_per_step'], d['fused_launch_repeats'], d.get('rehearsal',{}).get('fused_fallbacks'))
"
done
