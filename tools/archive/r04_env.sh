# A/B of one switch on a bench workload:  bash tools/r04_env.sh <workload> "<VAR=val>" "<VAR=val>" ...
WL=$1; shift
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for e in "$@"; do
  env $e timeout 600 python bench.py --workload $WL --cpu-steps 0 --steps 5 --warmup 2 > gpurun_out/env_tmp.json 2> gpurun_out/env_tmp.err
  python - "$e" <<'PY'
import json, sys
try:
    d = json.load(open("gpurun_out/env_tmp.json"))
    print(sys.argv[1], "ms", d["ms_per_step"], "it/s", d["value"], "parity", (d.get("parity") or {}).get("rel_err_vs_oracle"))
except Exception as ex:
    print(sys.argv[1], "ERR", ex)
PY
done
