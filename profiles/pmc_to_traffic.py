"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE csv -> profiles/traffic.json  {"<symbol>:<dims>": HBM bytes per launch}.

Units and corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
half the bytes of a wide coalesced read stream, so the read side is doubled (an upper bound for narrow access
patterns; Infinity-Cache hits are counted too).  The two counters are collected in SEPARATE passes.

Attribution: `bench.py --traffic-pass` enqueues a marker kernel (torch.cuda._sleep -> "spin_kernel") in front of every
C-ABI launch and prints the sequence of launches; the `pasnl::` rows between marker i and marker i+1 of the trace
(ordered by dispatch id) are launch i -- whatever number of kernels the entry point starts (grid build + query, sort +
sample ...).  Vendor GEMMs and torch kernels between the markers are not pasnl:: rows and are ignored.  The script
checks that the number of markers equals the number of launches before trusting the alignment.

    bash profiles/collect_traffic.sh          (on the GPU box)
    python profiles/pmc_to_traffic.py gpurun_out profiles/traffic.json
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def dispatches(root, counter, passdir=None):
    rows = {}
    for f in glob.glob(os.path.join(root, f"pmc_{passdir or counter}", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter:
                rows[int(r["Dispatch_Id"])] = (r["Kernel_Name"], float(r["Counter_Value"]))
    return [rows[k] for k in sorted(rows)]


def per_launch(root, counter, nlaunch, passdir=None):
    """-> [(kernel names, summed counter value)] per marked launch"""
    out, cur = [], None
    for name, val in dispatches(root, counter, passdir):
        if "spin_kernel" in name:
            cur = [[], 0.0]
            out.append(cur)
        elif cur is not None and "pasnl::" in name:
            cur[0].append(name)
            cur[1] += val
    if len(out) != nlaunch:
        raise SystemExit(f"{counter}: {len(out)} markers in the trace but {nlaunch} launches in the sequence")
    return out


def main(root, out):
    try:  # the line printed under rocprofv3 by profiles/collect_traffic.sh
        seq = json.loads(open(os.path.join(root, "pmc_FETCH_SIZE.json")).read().strip().splitlines()[-1])["launch_sequence"]
    except Exception as e:
        raise SystemExit(f"no launch_sequence line found (run profiles/collect_traffic.sh): {e}")
    fetch = per_launch(root, "FETCH_SIZE", len(seq))
    write = per_launch(root, "WRITE_SIZE", len(seq))
    # VALU-issue fraction (round 6): SQ_ACTIVE_INST_VALU counts the cycles the vector issue ports are busy in units of FOUR cycles,
    # summed over the SIMDs; GRBM_GUI_ACTIVE the cycles the launch kept the GPU busy, summed over the 8 XCDs:
    #   valu_frac = 4 ACTIVE / (1024 SIMDs x GUI / 8) = ACTIVE / (32 GUI)      (1.0 = every SIMD issues a vector instruction every cycle)
    valu = {}
    try:
        act = per_launch(root, "SQ_ACTIVE_INST_VALU", len(seq), "VALU")
        gui = per_launch(root, "GRBM_GUI_ACTIVE", len(seq), "VALU")
        vagg = defaultdict(list)
        for (sym, dims), (_, a), (_, g) in zip(seq, act, gui):
            if not sym.startswith("_") and g > 0:
                vagg[sym + ":" + ",".join(map(str, dims))].append(a / (32.0 * g))
        valu = {k: round(sum(v) / len(v), 4) for k, v in vagg.items()}
    except SystemExit as e:
        print(f"(no VALU pass: {e})")
    agg, kern = defaultdict(list), {}
    for (sym, dims), (names, f), (_, w) in zip(seq, fetch, write):
        if sym.startswith("_"):  # "_unmeasured": first launches, which also create weights and workspaces
            continue
        key = sym + ":" + ",".join(map(str, dims))
        agg[key].append((f, w))
        kern[key] = " + ".join(n.replace("void pasnl::", "").split("(")[0].split("<")[0] for n in names)
    traffic, detail = {}, {}
    for key, v in agg.items():
        f = sum(a for a, _ in v) / len(v)
        w = sum(b for _, b in v) / len(v)
        traffic[key] = int((2 * f + w) * 1024)
        detail[key] = {"kernels": kern[key], "launches_measured": len(v), "fetch_KiB_raw": round(f, 1),
                       "write_KiB": round(w, 1), "hbm_bytes_corrected": traffic[key]}
    # provenance: the commit and the digests of the kernel sources the counters were collected on; bench.py refuses to
    # report a figure whose kernel source has changed since (bench.measured_traffic)
    import subprocess
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, here)
    import bench
    commit = subprocess.run(["git", "-C", here, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or "?"
    traffic["_source"] = {"commit": commit, "csrc_sha256": bench.csrc_digests()}
    traffic["_valu_frac"] = valu
    json.dump(traffic, open(out, "w"), indent=1, sort_keys=True)
    del traffic["_source"]
    del traffic["_valu_frac"]
    json.dump(detail, open(out.replace(".json", "_detail.json"), "w"), indent=1, sort_keys=True)
    for k, v in sorted(traffic.items(), key=lambda kv: -kv[1]):
        print(f"{k:64s} {v/1e6:10.2f} MB   {kern[k][:60]}")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out", sys.argv[2] if len(sys.argv) > 2 else "profiles/traffic.json")
