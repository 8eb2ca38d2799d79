"""rocprofv3 (rocpd sqlite output) -> the kernel-stats summary committed under profiles/.

    rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o cls -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline
    python profiles/summarize_rocpd.py gpurun_out/prof/cls_results.db profiles/<name>_kernel_stats.csv
"""
import csv
import sqlite3
import sys


def main(db_path, out_path):
    cur = sqlite3.connect(db_path).cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    with open(out_path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDuration(us)", "AverageDuration(us)", "Percentage"])
        for name, calls, total, avg, pct in rows:
            w.writerow([name, calls, round(total, 3), round(avg, 3), round(pct, 3)])
    print(f"{len(rows)} kernels -> {out_path}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
