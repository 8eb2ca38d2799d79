#!/bin/bash
# MFMA utilisation of the hand-written matrix kernels from PMC counters: SQ_VALU_MFMA_BUSY_CYCLES (summed over all SIMDs)
# against GRBM_GUI_ACTIVE (busy cycles, reported as the SUM over the 8 XCDs: a 426 us kernel reads 8.1 M) x 128 SIMDs per XCD.  Run on the GPU box through gpurun:
#   gpurun -- 'bash profiles/collect_mfma_util.sh'   -> gpurun_out/pmc_mfma/..., summarised into gpurun_out/mfma_util.json
export TMPDIR=/tmp
mkdir -p gpurun_out; rm -rf gpurun_out/pmc_mfma
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA -d gpurun_out/pmc_mfma -o cls -f csv -- \
  python bench.py --worker --steps 3 --warmup 1 --no-cpu-baseline --no-graph --no-others > gpurun_out/pmc_mfma.json 2> gpurun_out/pmc_mfma.err || true
python - <<'PY'
import csv, glob, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_mfma/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "pasnl::" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, v in acc.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}
    if m.get("SQ_INSTS_MFMA", 0) > 0 and m.get("GRBM_GUI_ACTIVE", 0) > 0:
        out[k] = {"mfma_instructions": round(m["SQ_INSTS_MFMA"]), "mfma_busy_cycles_all_simds": round(m["SQ_VALU_MFMA_BUSY_CYCLES"]),
                  "gpu_active_cycles_sum_over_8_xcds": round(m["GRBM_GUI_ACTIVE"]),
                  "mfma_utilisation": round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] * 128), 4)}
json.dump(out, open("gpurun_out/mfma_util.json", "w"), indent=1, sort_keys=True)
for k, v in out.items():
    print(f"{k:60s} {v}")
PY
