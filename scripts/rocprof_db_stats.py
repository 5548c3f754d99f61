"""Per-kernel summary (calls, total / average / min / max duration) of a rocprofv3 results database, in the column layout of
rocprofv3's own kernel_stats.csv:  python scripts/rocprof_db_stats.py gpurun_out/<dir>/<name>_results.db profiles/rNN/<name>_kernel_stats.csv"""
import collections, csv, sqlite3, statistics, sys

db, out = sys.argv[1], sys.argv[2]
con = sqlite3.connect(db)
dur = collections.defaultdict(list)
for name, s, e in con.execute("select name, start, end from kernels"):
    dur[name].append(e - s)
total = sum(sum(v) for v in dur.values())
with open(out, "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for n, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([n, len(v), sum(v), round(sum(v) / len(v), 6), round(100 * sum(v) / total, 2), min(v), max(v), round(statistics.pstdev(v), 6)])
