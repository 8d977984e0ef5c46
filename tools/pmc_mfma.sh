# matrix-core utilisation of the supernode kernels (config-5 shape at a quarter of its size):
# one rocprofv3 counter pass (no trace domains besides --kernel-trace), summarised per kernel
# usage (on the GPU box): bash tools/pmc_mfma.sh
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_mfma
rm -rf $OUT
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT -o p -- python $GRAFT_REPO_ROOT/tools/scale_check.py c5m > $GRAFT_REPO_ROOT/gpurun_out/pmc_mfma.log 2>&1
python - <<'PY'
import csv, glob, json, os, collections
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_mfma"
files = glob.glob(out + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
seen = set()
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("chip::dev::(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        if not k.startswith("k_snode") and not k.startswith("k_factor_B"):
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
            cnt[k] += 1
res = {}
for k, c in acc.items():
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    res[k] = dict(dispatches=cnt[k], **{n: v for n, v in c.items()})
    if gui:
        res[k]["mfma_util_pct(gfx94x formula: MFMA_BUSY/(GUI_ACTIVE*256*4))"] = round(100.0 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui * 256 * 4), 2)
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    if wc:
        res[k]["wait_any_frac"] = round(c.get("SQ_WAIT_ANY", 0.0) / wc, 3)
        res[k]["wait_inst_frac"] = round(c.get("SQ_WAIT_INST_ANY", 0.0) / wc, 3)
        res[k]["active_inst_frac"] = round(c.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 3)
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:3000])
PY
find $OUT -name "*.csv" -size +1M -delete
rm -f $OUT/*.db $OUT/*/*.db 2>/dev/null
