# rocprofv3 kernel-trace summary of tools/ipm_scale.py <args> (the device-resident IPM loop);
# keeps the stats CSVs and a per-kernel list in launch order of the LAST iteration
# usage (on the GPU box): bash tools/prof_ipm.sh 1000 1000
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_ipm
rm -rf $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o r -- python $GRAFT_REPO_ROOT/tools/ipm_scale.py "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_ipm.log 2>&1
python - <<'PY'
import csv, os
out = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/prof_ipm"
rows = list(csv.DictReader(open(out + "/r_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last ~400 launches: a bit more than one iteration
with open(out + "/last_launches.txt", "w") as f:
    t0 = int(rows[-400]["Start_Timestamp"]) if len(rows) >= 400 else int(rows[0]["Start_Timestamp"])
    for r in rows[-400:]:
        f.write("%10.1f %8.1f %s\n" % ((int(r["Start_Timestamp"]) - t0) / 1e3,
                                      (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                      r["Kernel_Name"].replace("chip::dev::(anonymous namespace)::", "")[:70]))
PY
rm -f $OUT/r_kernel_trace.csv $OUT/*.db
tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_ipm.log | cut -c1-300
