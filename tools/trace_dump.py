#!/usr/bin/env python
"""Kernel timeline of the last full step of a rocprofv3 --kernel-trace database: start (us from the step's first kernel), duration, queue, name.
usage: python tools/trace_dump.py <results.db> [out.txt]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    rows = sqlite3.connect(db).execute("select start, end, name, queue_id, grid_x, workgroup_x from kernels order by start").fetchall()
    marks = [i for i, r in enumerate(rows) if r[2].startswith("adamw_kernel")]
    seg = rows[marks[-2] + 1:marks[-1] + 1]
    t0 = seg[0][0]
    qs = sorted({r[3] for r in seg})
    for s, e, n, q, gx, wx in seg:
        out.write(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f}  {'  ' * qs.index(q)}q{q} {n[:70]}  [{gx // max(wx, 1)} wg]\n")


if __name__ == "__main__":
    main()
