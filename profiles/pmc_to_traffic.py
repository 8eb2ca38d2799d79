"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE csv -> profiles/traffic.json  {"<symbol>:<dims>": HBM bytes per launch}.

Units and corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
half the bytes of a wide coalesced read stream, so the read side is doubled (an upper bound for narrow access
patterns; Infinity-Cache hits are counted too).  The two counters are collected in SEPARATE passes.

Attribution: `bench.py --no-graph --launch-order` prints the per-forward sequence of C-ABI launches; every launch
of the forward path starts exactly one `pasnl::` kernel, and every forward of the run is the same sequence, so the
j-th pasnl dispatch of the trace (ordered by dispatch id) belongs to launch j mod len(sequence).  The script checks
that the kernel name at each position is the same in every period before trusting the alignment.

    bash profiles/collect_traffic.sh          (on the GPU box)
    python profiles/pmc_to_traffic.py gpurun_out profiles/traffic.json
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def dispatches(root, counter):
    rows = {}
    for f in glob.glob(os.path.join(root, f"pmc_{counter}", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter and "pasnl::" in r["Kernel_Name"]:
                rows[int(r["Dispatch_Id"])] = (r["Kernel_Name"].split("(")[0], float(r["Counter_Value"]))
    return [rows[k] for k in sorted(rows)]


def per_position(root, counter, order):
    d = dispatches(root, counter)
    n = len(order)
    if not d or len(d) % n:
        raise SystemExit(f"{counter}: {len(d)} pasnl dispatches is not a multiple of the {n} launches of one forward")
    names = [None] * n
    acc = defaultdict(list)
    for j, (name, val) in enumerate(d):
        p = j % n
        if names[p] is None:
            names[p] = name
        elif names[p] != name:
            raise SystemExit(f"{counter}: position {p} is {names[p]} in one forward and {name} in another")
        acc[p].append(val)
    return names, [sum(acc[p]) / len(acc[p]) for p in range(n)]


def main(root, out):
    order = None
    try:  # the bench line printed under rocprofv3 by profiles/collect_traffic.sh
        order = json.load(open(os.path.join(root, "pmc_FETCH_SIZE.json")))["launch_order"]
    except Exception:
        pass
    if not order:
        raise SystemExit("no bench line with launch_order found (run profiles/collect_traffic.sh)")
    names, fetch = per_position(root, "FETCH_SIZE", order)
    _, write = per_position(root, "WRITE_SIZE", order)
    agg = defaultdict(list)
    kern = {}
    for p, (sym, dims) in enumerate(order):
        key = sym + ":" + ",".join(map(str, dims))
        agg[key].append((fetch[p], write[p]))
        kern[key] = names[p]
    traffic, detail = {}, {}
    for key, v in agg.items():
        f = sum(a for a, _ in v) / len(v)
        w = sum(b for _, b in v) / len(v)
        traffic[key] = int((2 * f + w) * 1024)
        detail[key] = {"kernel": kern[key], "launches_per_forward": len(v), "fetch_KiB_raw": round(f, 1),
                       "write_KiB": round(w, 1), "hbm_bytes_corrected": traffic[key]}
    # provenance: the commit and the digests of the kernel sources the counters were collected on; bench.py refuses to
    # report a figure whose kernel source has changed since (bench.measured_traffic)
    import subprocess
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, here)
    import bench
    commit = subprocess.run(["git", "-C", here, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or "?"
    traffic["_source"] = {"commit": commit, "csrc_sha256": bench.csrc_digests()}
    json.dump(traffic, open(out, "w"), indent=1, sort_keys=True)
    del traffic["_source"]
    json.dump(detail, open(out.replace(".json", "_detail.json"), "w"), indent=1, sort_keys=True)
    for k, v in sorted(traffic.items(), key=lambda kv: -kv[1]):
        print(f"{k:60s} {v/1e6:10.2f} MB   {kern[k][:60]}")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out", sys.argv[2] if len(sys.argv) > 2 else "profiles/traffic.json")
