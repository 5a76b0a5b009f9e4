#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into a per-kernel stats CSV + markdown table.
usage: python tools/prof_summary.py <results.db> <out_prefix> [steps]"""
import csv
import sqlite3
import sys


def main():
    db, out = sys.argv[1], sys.argv[2]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else None
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
                       "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(workgroup_x) "
                       "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    with open(out + "_kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage", "VGPR", "AGPR", "SGPR", "LDS", "WG"])
        for r in rows:
            w.writerow([r[0], r[1], int(r[2]), round(r[3], 1), r[4], r[5], round(100.0 * r[2] / tot, 2)] + list(r[6:]))
    with open(out + "_kernel_stats.md", "w") as f:
        f.write(f"rocprofv3 --kernel-trace summary of `{db}`; total kernel time {tot / 1e6:.2f} ms"
                + (f" over {steps} steps = {tot / 1e6 / steps:.2f} ms/step" if steps else "") + "\n\n")
        f.write("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
        for r in rows[:30]:
            f.write(f"| `{r[0][:90]}` | {r[1]} | {r[2] / 1e6:.2f} | {r[3] / 1e3:.1f} | {100.0 * r[2] / tot:.1f} |\n")
    print(open(out + "_kernel_stats.md").read())


if __name__ == "__main__":
    main()
