# the round's evidence in one call:  bash tools/r05_final.sh <tag> [parts: tests bench lines stats pmc rehearsal]
TAG=${1:-r05_final}
PARTS=${2:-tests bench lines stats pmc rehearsal}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
has() { case " $PARTS " in *" $1 "*) return 0;; *) return 1;; esac; }
line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    r = d.get("roofline") or {}
    print(sys.argv[1].split("/")[-1], "it/s", d["value"], "ms", d["ms_per_step"], "kernel", (r.get("kernel") or "")[:28], "us", r.get("avg_launch_us"), "frac", r.get("frac"),
          "parity", (d.get("parity") or {}).get("rel_err_vs_oracle"), "other", (d.get("config") or {}).get("other_solve_policy"), "rehearsal", d.get("rehearsal") and {k: d["rehearsal"][k] for k in ("ms_per_step", "fused_launch_repeats", "fused_fallbacks")})
    for k in ("c2", "c5", "batched_c4", "l1_dropin"):
        o = d.get(k)
        if o:
            print("   ", k, {kk: o.get(kk) for kk in ("value", "ms_per_step", "setup_s", "error")}, "frac", (o.get("roofline") or {}).get("frac"), "parity", (o.get("parity") or {}).get("rel_err_vs_oracle"),
                  "mt", (o.get("cpu_baseline_mt") or {}).get("value"), "fast", (o.get("fast_path") or {}).get("value"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
}
if has tests; then
  timeout 1500 python -m pytest tests -m gpu -q > $O/${TAG}_pytest.log 2>&1
  grep -E "^(FAILED|ERROR)|passed|failed" $O/${TAG}_pytest.log | tail -12 | cut -c1-200
fi
if has bench; then
  timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
  line $O/${TAG}_bench.json; tail -2 $O/${TAG}_bench.err | cut -c1-200
fi
if has lines; then
  for w in c2 c5 c4_128 c4_256 c4_512; do
    case $w in
      c4_*) extra="--workload c4 --nbatch ${w#c4_}";;
      *) extra="--workload $w";;
    esac
    timeout 1200 python bench.py $extra > $O/${TAG}_bench_$w.json 2> $O/${TAG}_bench_$w.err
    line $O/${TAG}_bench_$w.json; tail -1 $O/${TAG}_bench_$w.err | cut -c1-200
  done
fi
if has stats; then
  for w in c3 c2 c5 c4_128; do
    case $w in
      c4_*) extra="--workload c4 --nbatch ${w#c4_}";;
      *) extra="--workload $w";;
    esac
    (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_$w -o r -- python $R/bench.py $extra --cpu-steps 0 --steps 6 --warmup 1 --no-extras > $O/${TAG}_prof_$w.log 2>&1)
    f=$(ls $O/${TAG}_prof_$w/*kernel_stats.csv 2>/dev/null | head -1)
    if [ -n "$f" ]; then cp $f $O/${TAG}_${w}_kernel_stats.csv; echo "== $w"; head -9 $f | cut -c1-150; fi
    rm -rf $O/${TAG}_prof_$w
  done
fi
if has pmc; then
  bash $R/tools/r05_pmc_all.sh $TAG "auto c2 c5 c4_128"
fi
if has rehearsal; then
  timeout 600 python bench.py --workload c4 --nbatch 128 --force-comm --cpu-steps 0 --no-extras > $O/${TAG}_rehearsal_comm_only.json 2> $O/${TAG}_rehearsal.err
  line $O/${TAG}_rehearsal_comm_only.json
  for cfg in 8:60 16:60 32:120; do
    timeout 600 python bench.py --workload c4 --nbatch 128 --force-comm --coresident $cfg --cpu-steps 0 --no-extras > $O/${TAG}_rehearsal_${cfg/:/_}.json 2>> $O/${TAG}_rehearsal.err
    line $O/${TAG}_rehearsal_${cfg/:/_}.json
  done
  tail -2 $O/${TAG}_rehearsal.err | cut -c1-200
fi
