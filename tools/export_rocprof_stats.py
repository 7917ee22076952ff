"""Exports the kernel statistics of a rocprofv3 results database (rocpd sqlite, `--kernel-trace --stats`) to CSV."""
import csv
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
con = sqlite3.connect(db)
# one row per (kernel, grid, workgroup): a kernel name alone mixes launches of different batch sizes; the
# bench line's `extra` entries name their kernel and grid, so every quoted fraction can be recomputed from this file
rows = con.execute(
    "select name, count(*), sum(duration)/1000.0, avg(duration)/1000.0, min(duration)/1000.0, max(duration)/1000.0, "
    "grid_x, workgroup_x, max(lds_size), max(vgpr_count), max(sgpr_count), max(scratch_size) "
    "from kernels group by name, grid_x, workgroup_x order by sum(duration) desc").fetchall()
total = sum(r[2] for r in rows) or 1.0
with open(out, "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct", "grid_x", "workgroup_x", "lds_bytes", "vgprs",
                "sgprs", "scratch"])
    for r in rows:
        w.writerow([r[0].replace("(anonymous namespace)::", ""), r[1], round(r[2], 3), round(r[3], 3), round(r[4], 3), round(r[5], 3),
                    round(100 * r[2] / total, 2)] + list(r[6:]))
print("wrote", out, len(rows), "kernels")
