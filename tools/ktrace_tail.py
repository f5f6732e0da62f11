"""Steady-state kernel time from a rocprofv3 kernel trace: the kernels of the LAST `steps` of `total` equal steps (by time:
the trace's last steps/total fraction after the first kernel), summed by name -- leaves out the library's solver search and
JIT of the first steps.  python tools/ktrace_tail.py <dir> <steps> <total> [top]"""
import collections, csv, glob, sys
d, steps, total = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
top = int(sys.argv[4]) if len(sys.argv) > 4 else 40
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
# steps are separated by the optimizer's kernel: find the starts of the last `steps` multi_tensor_apply groups
marks = [s for s, e, n in rows if "multi_tensor_apply" in n or "fused_adam" in n.lower()]
groups = []
for m in marks:
    if not groups or m - groups[-1][-1] > 2e6:
        groups.append([m])
    else:
        groups[-1].append(m)
ends = [g[-1] for g in groups]
assert len(ends) >= steps + 1, (len(ends), steps)
t0, t1 = ends[-steps - 1], ends[-1]
agg = collections.defaultdict(lambda: [0.0, 0])
for s, e, n in rows:
    if t0 < s <= t1:
        a = agg[n]; a[0] += (e - s) / 1e6; a[1] += 1
wall = (t1 - t0) / 1e6 / steps
tot = sum(v[0] for v in agg.values()) / steps
print("last %d steps: %.2f ms wall per step, %.2f ms of kernels per step, %d launches per step" % (steps, wall, tot, sum(v[1] for v in agg.values()) // steps))
for n, (ms, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:top]:
    print("%8.3f ms x%-4d %s" % (ms / steps, c // steps, n[:150]))
