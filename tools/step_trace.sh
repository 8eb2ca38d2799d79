cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/trace_cls
export PASNL_X=1
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/trace_cls -o t -f csv -- python bench.py --worker --steps 6 --warmup 3 --no-cpu-baseline --no-others > /dev/null 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/trace_cls/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last occurrence of the tree kernel, and everything within 400 us before / 100 us after its end
idx = [i for i, r in enumerate(rows) if "knn_tree_small" in r["Kernel_Name"]]
i0 = idx[-2]
t_end = int(rows[i0]["End_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > t_end - 1500000 and s < t_end + 120000:
        print(f"{(s - t_end) / 1000:9.1f} {(e - t_end) / 1000:9.1f}  q{r.get('Queue_Id','?')} {r['Kernel_Name'][:60]} grid {r.get('Grid_Size_X', r.get('Grid_Size','?'))} wg {r.get('Workgroup_Size_X', r.get('Workgroup_Size','?'))}")
PY
