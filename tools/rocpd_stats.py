#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (``*_results.db``) into the same per-kernel table
that ``rocprofv3 --kernel-trace --stats`` prints: name, calls, total / average / min / max
duration (ns) and share of GPU time.  Usage: tools/rocpd_stats.py results.db [out.csv]
With BY_GRID=1 in the environment the rows are per (kernel, grid size): one row per GEMM shape."""
import csv
import sqlite3
import sys


def by_grid(cur, out):
    q = """select s.kernel_name, d.grid_size_x, d.grid_size_y, d.grid_size_z, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start)
           from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
           group by s.kernel_name, d.grid_size_x, d.grid_size_y, d.grid_size_z order by 6 desc"""
    w = csv.writer(out)
    w.writerow(["Name", "GridX", "GridY", "GridZ", "Calls", "TotalDurationNs", "AverageNs", "MinNs"])
    for r in cur.execute(q):
        w.writerow([r[0], r[1], r[2], r[3], r[4], int(r[5]), f"{r[6]:.1f}", int(r[7])])


def main():
    import os
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    if os.environ.get("BY_GRID"):
        return by_grid(cur, open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout)
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
