"""Per-step kernel counts from a rocprofv3 kernel_stats.csv: python scripts/kernel_count_report.py <dir> <steps>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
steps = int(sys.argv[2])
tot = 0
for r in csv.DictReader(open(f)):
    n = r["Name"]; c = int(r["Calls"])
    short = n.split("dasp::")[1].split("(")[0] if "dasp::" in n else n[:80]
    if c >= steps:
        print("%5.1f  %7.1f us  %s" % (c / steps, float(r["AverageNs"]) / 1e3, short))
        tot += c / steps
print("kernels per step", tot)
