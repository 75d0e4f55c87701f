#!/usr/bin/env python3
"""Summarise a rocprofv3 results .db (rocpd sqlite) as a kernel-stats table:
name, calls, total/avg/min/max duration (ns), % of GPU kernel time."""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    rows = db.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), "
        "max(duration), max(vgpr_count), max(sgpr_count), max(lds_size), "
        "max(grid_x), max(workgroup_x) from kernels group by name "
        "order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print("| kernel | calls | total_ns | avg_ns | min_ns | max_ns | % | vgpr "
          "| sgpr | lds | grid_x | wg_x |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        name = r[0].split("(")[0][-60:]
        print(f"| {name} | {r[1]} | {r[2]} | {r[3]:.0f} | {r[4]} | {r[5]} | "
              f"{100.0 * r[2] / tot:.2f} | {r[6]} | {r[7]} | {r[8]} | {r[9]} "
              f"| {r[10]} |")


if __name__ == "__main__":
    main(sys.argv[1])
