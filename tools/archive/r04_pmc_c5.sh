# one PMC pass over a short config-5 run, summarised per supernode kernel:  bash tools/r04_pmc_c5.sh <tag> "<counters>"
TAG=${1:-r04_pmc}
CTR=${2:-SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${TAG}
rm -rf $OUT
timeout 900 rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d $OUT -o p -- python $GRAFT_REPO_ROOT/bench.py --workload c5 --cpu-steps 0 --no-extras --steps 2 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/${TAG}.log 2>&1
python - $OUT $GRAFT_REPO_ROOT/gpurun_out/${TAG}_summary.json <<'PY'
import csv, glob, json, sys, collections
out, dst = sys.argv[1], sys.argv[2]
files = glob.glob(out + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter(); seen = set()
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("chip::dev::(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        if not (k.startswith("k_snode") or k.startswith("k_dblk") or k.startswith("k_psd") or k.startswith("k_scatter")):
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key); cnt[k] += 1
res = {}
for k, c in acc.items():
    d = dict(dispatches=cnt[k], **c)
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    if gui and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        d["mfma_busy_over_gui_x_1024"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * 1024), 4)
    if c.get("SQ_BUSY_CU_CYCLES"):
        d["mfma_busy_over_busy_cu_cycles"] = round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / c["SQ_BUSY_CU_CYCLES"], 4)
    if c.get("SQ_WAVE_CYCLES"):
        d["wait_inst_frac"] = round(c.get("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"], 3)
    h, m = c.get("TCC_HIT_sum", 0.0), c.get("TCC_MISS_sum", 0.0)
    if h + m:
        d["l2_hit_rate"] = round(h / (h + m), 4)
    res[k] = d
json.dump(res, open(dst, "w"), indent=1)
for k, d in res.items():
    print(k, {x: d[x] for x in d if x in ("dispatches", "mfma_busy_over_gui_x_1024", "mfma_busy_over_busy_cu_cycles", "wait_inst_frac", "l2_hit_rate")})
PY
tail -3 $GRAFT_REPO_ROOT/gpurun_out/${TAG}.log | cut -c1-300
rm -rf $OUT
