#!/usr/bin/env python
"""GPU idle time between kernels of the steady-state steps, from a rocprofv3 --kernel-trace database.
usage: python tools/trace_gaps.py <results.db> <steps in the trace> [steps to analyse from the end]"""
import sqlite3
import sys


def main():
    db, steps = sys.argv[1], int(sys.argv[2])
    last = int(sys.argv[3]) if len(sys.argv) > 3 else max(1, steps // 2)
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select start, end, name from kernels order by start").fetchall()
    marks = [i for i, r in enumerate(rows) if r[2].startswith("adamw_kernel")]        # one per step, the last big kernel of a step
    assert len(marks) >= last + 1, (len(marks), last)
    lo, hi = marks[-last - 1] + 1, marks[-1] + 1
    seg = rows[lo:hi]
    busy = sum(e - s for s, e, _ in seg)
    wall = seg[-1][1] - seg[0][0]
    gaps = [max(0, seg[i + 1][0] - seg[i][1]) for i in range(len(seg) - 1)]
    overlap = sum(max(0, seg[i][1] - seg[i + 1][0]) for i in range(len(seg) - 1))
    big = sorted(gaps, reverse=True)[:10]
    print(f"{last} steps: {len(seg) / last:.0f} kernels/step, wall {wall / last / 1e6:.3f} ms/step, busy {busy / last / 1e6:.3f} ms/step, "
          f"idle {sum(gaps) / last / 1e6:.3f} ms/step (mean gap {sum(gaps) / len(gaps) / 1e3:.2f} us, median {sorted(gaps)[len(gaps) // 2] / 1e3:.2f} us), "
          f"overlap {overlap / last / 1e6:.3f} ms/step")
    print("largest gaps (us):", [round(g / 1e3, 1) for g in big])
    step0 = rows[marks[-2] + 1:marks[-1] + 1]
    g0 = sorted(((step0[i + 1][0] - step0[i][1], i) for i in range(len(step0) - 1)), reverse=True)[:14]
    print("largest gaps of the last step, in launch order (gap us: kernel index, previous -> next):")
    for g, i in sorted(g0, key=lambda t: t[1]):
        print(f"  {g / 1e3:7.1f}: #{i:3d} {step0[i][2][:48]:48s} -> {step0[i + 1][2][:60]}")
    # which kernels precede the large gaps
    by = {}
    for i, g in enumerate(gaps):
        k = seg[i + 1][2][:60]
        by.setdefault(k, [0, 0]); by[k][0] += g; by[k][1] += 1
    for k, (g, n) in sorted(by.items(), key=lambda kv: -kv[1][0])[:12]:
        print(f"  before {k:60s} {g / last / 1e3:8.1f} us/step over {n / last:.0f} launches ({g / n / 1e3:.2f} us each)")


if __name__ == "__main__":
    main()
