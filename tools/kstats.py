"""Print a rocprofv3 kernel_stats.csv compactly: python tools/kstats.py <dir> [top]"""
import csv, glob, sys
d = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 6
for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[:1]:
    rows = list(csv.DictReader(open(f)))
    for r in rows[:top]:
        print("%-60s calls %4s  avg %10.3f ms  total %10.3f ms" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6))
