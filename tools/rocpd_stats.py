#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (``*_results.db``) into the same per-kernel table
that ``rocprofv3 --kernel-trace --stats`` prints: name, calls, total / average / min / max
duration (ns) and share of GPU time.  Usage: tools/rocpd_stats.py results.db [out.csv]"""
import csv
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    q = """select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start)
           from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
           group by s.kernel_name order by 3 desc"""
    rows = list(cur.execute(q))
    total = sum(r[2] for r in rows) or 1
    out = open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout
    w = csv.writer(out)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
    for r in rows:
        w.writerow([r[0], r[1], int(r[2]), f"{r[3]:.1f}", int(r[4]), int(r[5]), f"{100 * r[2] / total:.2f}"])


if __name__ == "__main__":
    main()
