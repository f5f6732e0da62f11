#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2 rocpd SQLite) kernel trace into the
per-kernel table `rocprofv3 --stats` would print: calls, total/avg/min/max
duration.  Usage: python tools/rocpd_summary.py <results.db> [out.md]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute(
        "select name, count(*), sum(end-start), avg(end-start), min(end-start), "
        "max(end-start) from kernels group by name order by sum(end-start) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |",
             "|---|---|---|---|---|---|---|"]
    for name, calls, tot, avg, mn, mx in rows:
        short = name if len(name) < 90 else name[:87] + "..."
        lines.append("| `%s` | %d | %.3f | %.1f | %.1f | %.1f | %.1f |" % (
            short, calls, tot / 1e6, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total))
    text = "\n".join(lines)
    print(text)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")


if __name__ == "__main__":
    main()
