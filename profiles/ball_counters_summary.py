"""rocprofv3 --pmc csvs of profiles/collect_ball_counters.sh -> one json: per batch size (grid y of the launch) the average of
every counter over the launches of ball_query_grid_kernel, plus derived ratios."""
import csv, glob, json, os, sys
from collections import defaultdict
root, out = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "ballpmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "ball_grid_kernel" not in r["Kernel_Name"] and "ball_query_grid_kernel" not in r["Kernel_Name"]:
            continue
        # clouds of the launch: one workgroup of Workgroup_Size threads per cloud at the profiled shape (512 queries)
        b = int(r["Grid_Size"]) // max(1, int(r.get("Workgroup_Size") or 256))
        acc[b][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for b in sorted(acc):
    c = {k: sum(v) / len(v) for k, v in acc[b].items()}
    d = dict(c)
    if "SQ_WAVE_CYCLES" in c and "SQ_BUSY_CYCLES" in c and c["SQ_BUSY_CYCLES"]:
        d["waves_per_busy_cycle (occupancy, all SIMDs)"] = round(c["SQ_WAVE_CYCLES"] / c["SQ_BUSY_CYCLES"], 2)
    if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_fraction_of_lds_active"] = round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 3)
    if "SQ_ACTIVE_INST_LDS" in c and c.get("SQ_ACTIVE_INST_VALU"):
        d["lds_to_valu_active_ratio"] = round(c["SQ_ACTIVE_INST_LDS"] / c["SQ_ACTIVE_INST_VALU"], 3)
    res[f"B={b}"] = d
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
for b, d in res.items():
    print(b, {k: (round(v, 1) if isinstance(v, float) else v) for k, v in d.items()})
