"""Per-call durations of the kernels whose name contains a pattern: python tools/ktrace.py <dir> <pattern>"""
import csv, glob, sys
d, pat = sys.argv[1], sys.argv[2]
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[:1]:
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            print("%-50s grid %s wg %s lds %s: %.3f ms" % (r["Kernel_Name"][:50], r.get("Grid_Size_X", "?"), r.get("Workgroup_Size_X", "?"),
                                                          r.get("LDS_Block_Size", "?"), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
