cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for s in cls scannet; do
  timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/tp_$s -o tp -f csv -- python tools/tie_path_time.py $s 2>&1 | grep listed | cut -c1-60
  python - <<PY
import csv,glob
f=glob.glob("gpurun_out/tp_$s/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    print(r["Name"][:90], r["Calls"], r["AverageNs"])
PY
done
