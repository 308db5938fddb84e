"""rocprofv3 (ROCm 7.2) writes a rocpd SQLite database; this aggregates its `kernels` view by
kernel name into the per-kernel statistics CSV kept under profiles/.
python tools/rocpd_stats.py gpurun_out/prof/b4_results.db profiles/rNN_kernel_stats.csv"""
import csv, sqlite3, statistics, sys

db = sqlite3.connect(sys.argv[1])
by = {}
for name, dur in db.execute("select name, duration from kernels"):
    by.setdefault(name, []).append(dur)
tot = sum(sum(v) for v in by.values())
rows = sorted(((n, len(v), sum(v), sum(v) / len(v), 100.0 * sum(v) / tot, min(v), max(v),
                statistics.pstdev(v)) for n, v in by.items()), key=lambda r: -r[2])
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for r in rows:
        w.writerow([r[0], r[1], r[2], round(r[3], 3), round(r[4], 4), r[5], r[6], round(r[7], 3)])
for r in rows[:8]:
    print(f"{r[0][:70]:70s} {r[1]:6d} {r[3] / 1e3:9.1f} us {r[4]:6.2f} %")
