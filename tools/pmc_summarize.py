#!/usr/bin/env python3
"""Summarise the two rocprofv3 --pmc passes of tools/pmc_traffic.sh into profiles/<tag>_pmc_traffic.json.

HBM traffic per launch = FETCH_SIZE * 2 + WRITE_SIZE (counter unit: KiB).  The x2 on the
read side is the gfx950 correction of MI355X_MICROARCH.md (HBM section): on this rocprofv3
FETCH_SIZE tallies 128-byte requests at 64 B, i.e. reports exactly 1/2 of a coalesced
stream.  Calibrated here on k_add_vec (reads 2 x 8N bytes, writes 8N): FETCH_SIZE = 8N/..."""
import collections
import csv
import re
import json
import sys


def per_kernel(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].replace("chip::dev::(anonymous namespace)::", "")
        k = re.sub(r"<.*>", "", k.split("(")[0].replace("void ", ""))  # k_bundle_ir<256> -> k_bundle_ir
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    return {k: (c, v / c * 1024.0) for k, (c, v) in agg.items()}


def main():
    base, out = sys.argv[1], sys.argv[2]
    workload = sys.argv[3] if len(sys.argv) > 3 else "auto"
    desc = {"auto": "bench.py default (config 3, n=10^6)", "c2": "bench.py --workload c2 (config 2, QP n=1e5)",
            "c5": "bench.py --workload c5 (config 5, 200 x PSD(50) + 200 x SOC(51))",
            "c4": "bench.py --workload c4"}.get(workload, workload)
    f = per_kernel(base + "/pmc_FETCH_SIZE/p_counter_collection.csv", "FETCH_SIZE")
    w = per_kernel(base + "/pmc_WRITE_SIZE/p_counter_collection.csv", "WRITE_SIZE")
    res = {"unit": "bytes per launch", "correction": "read = 2 x FETCH_SIZE (gfx950), write = WRITE_SIZE",
           "workload": desc, "kernels": {}}
    for k in sorted(set(f) | set(w)):
        if not k.startswith("k_"):
            continue
        fr = f.get(k, (0, 0.0))[1]
        wr = w.get(k, (0, 0.0))[1]
        res["kernels"][k] = {"launches_sampled": f.get(k, (0, 0))[0], "fetch_size_raw": round(fr),
                             "write_size_raw": round(wr), "hbm_bytes": round(2 * fr + wr)}
    json.dump(res, open(out, "w"), indent=1)
    for k, v in sorted(res["kernels"].items(), key=lambda kv: -kv[1]["hbm_bytes"])[:8]:
        print("%-28s %8.1f MB" % (k, v["hbm_bytes"] / 1e6))


if __name__ == "__main__":
    main()
