"""One replayed forward as a timeline from a rocprofv3 --kernel-trace csv: start offset, duration, queue, kernel.
    rocprofv3 --kernel-trace -d gpurun_out/tl -o tl -f csv -- python bench.py --worker --steps 6 --warmup 2 --no-cpu-baseline
    python tools/timeline.py gpurun_out/tl [step_from_end]"""
import csv, glob, os, sys
root = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
marker = sys.argv[3] if len(sys.argv) > 3 else None  # substring of the kernel a forward starts with
f = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(({"s": int(r["Start_Timestamp"]), "e": int(r["End_Timestamp"]), "q": r.get("Queue_Id", "?"), "n": r["Kernel_Name"]}
               for r in csv.DictReader(open(f))), key=lambda r: r["s"])
# a forward starts with the first FPS kernel of n=1024 (fps_kernel<4, 4...) -- split on it
starts = [i for i, r in enumerate(rows) if (marker in r["n"] if marker else ("fps_kernel<4, 4" in r["n"] or "fps_kernel<16" in r["n"]))]
if marker:  # several launches of the marker kernel per forward: keep the first of each burst (> 1 ms apart)
    starts = [i for n_, i in enumerate(starts) if n_ == 0 or rows[i]["s"] - rows[starts[n_ - 1]]["s"] > 1_000_000]
if len(starts) < back + 1:
    starts = [0, len(rows)]
a, b = starts[-back - 1], starts[-back]
t0 = rows[a]["s"]
busy = 0
for r in rows[a:b]:
    name = r["n"].split("(")[0].replace("void ", "")
    if name.startswith("Cijk"):
        name = "GEMM " + name.split("_MT")[1].split("_")[0]
    print(f"{(r['s'] - t0) / 1e3:9.1f} us  +{(r['e'] - r['s']) / 1e3:8.1f}  q{r['q']:>3}  {name[:90]}")
    busy += r["e"] - r["s"]
print(f"step span {(rows[b - 1]['e'] - t0) / 1e3:.1f} us, next step starts at {(rows[b]['s'] - t0) / 1e3 if b < len(rows) else -1:.1f} us, sum of kernel durations {busy / 1e3:.1f} us")
