"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE csv -> profiles/traffic.json  {"<symbol>:<dims>": HBM bytes per launch}.

Units and corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
half the bytes of a wide coalesced read stream, so the read side is doubled (an upper bound for narrow access
patterns).  Kernels are matched to C-ABI symbols by name; launches of one kernel with different shapes are told
apart by grid size order (largest first = largest shape).
"""
import csv
import glob
import json
import sys
from collections import defaultdict


def load(pattern, counter):
    per = defaultdict(list)
    for f in glob.glob(pattern, recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter:
                per[(r["Kernel_Name"], int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in per.items()}


def main(out):
    fetch = load("gpurun_out/pmc_FETCH_SIZE/**/*counter_collection.csv", "FETCH_SIZE")
    write = load("gpurun_out/pmc_WRITE_SIZE/**/*counter_collection.csv", "WRITE_SIZE")
    rows = {}
    for (name, grid), kb in fetch.items():
        if "pasnl::" not in name:
            continue
        w = write.get((name, grid), 0.0)
        rows[f"{name.split('(')[0]}|grid={grid}"] = {"fetch_KiB_raw": kb, "write_KiB": w,
                                                      "hbm_bytes_corrected": int((2 * kb + w) * 1024)}
    json.dump(rows, open(out, "w"), indent=1)
    for k, v in sorted(rows.items(), key=lambda kv: -kv[1]["hbm_bytes_corrected"]):
        print(f"{k[:90]:90s} {v['hbm_bytes_corrected']/1e6:10.2f} MB")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "profiles/traffic_raw.json")
