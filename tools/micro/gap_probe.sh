R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/gap_probe $R/tools/micro/gap_probe.hip || exit 1
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $O/gap_probe -o r -- /tmp/gap_probe > $O/gap_probe.log 2>&1)
python - $O/gap_probe/r_kernel_trace.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Workgroup_Size_X", ""), r.get("Grid_Size_X", ""), r.get("LDS_Block_Size", ""), r.get("Scratch_Size", "")) for r in rows)
pe = None
for s, e, n, w, g, l, sc in ev[-16:]:
    print("%-24s grid %7s lds %6s scratch %5s  dur %7.1f us  gap %6.1f us" % (n[:24], g, l, sc, (e - s) / 1e3, 0 if pe is None else (s - pe) / 1e3))
    pe = e
PY
rm -rf $O/gap_probe
