#!/usr/bin/env python
"""Two-stream timeline of the steady-state steps from a rocprofv3 --kernel-trace database: wall time, union of busy intervals, per-queue
busy time, and -- per kernel name -- how long the MAIN queue sat idle in front of it (i.e. what it was waiting for).
usage: python tools/trace_streams.py <results.db> <steps to analyse from the end>"""
import sqlite3
import sys


def main():
    db, last = sys.argv[1], int(sys.argv[2])
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info(kernels)").fetchall()]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    print("columns:", cols)
    rows = con.execute(f"select start, end, name, {qcol or '0'} from kernels order by start").fetchall()
    marks = [i for i, r in enumerate(rows) if r[2].startswith("adamw_kernel")]
    lo, hi = marks[-last - 1] + 1, marks[-1] + 1
    seg = rows[lo:hi]
    wall = seg[-1][1] - seg[0][0]
    # union of intervals
    union, cur_s, cur_e = 0, None, None
    for s, e, _, _ in seg:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                union += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    union += cur_e - cur_s
    total = sum(e - s for s, e, _, _ in seg)
    print(f"{last} steps: wall {wall / last / 1e6:.3f} ms/step, union busy {union / last / 1e6:.3f}, sum of kernels {total / last / 1e6:.3f}, "
          f"idle {(wall - union) / last / 1e6:.3f}, concurrent {(total - union) / last / 1e6:.3f}")
    queues = {}
    for s, e, n, q in seg:
        queues.setdefault(q, []).append((s, e, n))
    main_q = max(queues, key=lambda q: sum(e - s for s, e, _ in queues[q]))
    for q, ks in queues.items():
        print(f"  queue {q}: {len(ks) / last:.0f} kernels/step, busy {sum(e - s for s, e, _ in ks) / last / 1e6:.3f} ms/step" + (" (main)" if q == main_q else ""))
    ks = queues[main_q]
    by = {}
    for i in range(1, len(ks)):
        g = max(0, ks[i][0] - ks[i - 1][1])
        k = ks[i][2][:70]
        by.setdefault(k, [0, 0]); by[k][0] += g; by[k][1] += 1
    print("main-queue idle in front of:")
    for k, (g, n) in sorted(by.items(), key=lambda kv: -kv[1][0])[:14]:
        print(f"  {k:70s} {g / last / 1e3:8.1f} us/step over {n / last:.0f} launches ({g / n / 1e3:.2f} us each)")
    # duration of the main-queue kernels while a side kernel is running vs alone
    side = sorted((s, e) for q, v in queues.items() if q != main_q for s, e, _ in v)
    if side:
        import bisect
        starts = [s for s, _ in side]
        ov = {}
        for s, e, n in ks:
            j = bisect.bisect_left(starts, e)
            o = 0
            for a, b in side[max(0, j - 40):j]:
                o += max(0, min(e, b) - max(s, a))
            d = ov.setdefault(n[:70], [0, 0, 0]); d[0] += e - s; d[1] += o; d[2] += 1
        print("main-queue kernels: avg duration, share of it with a side-queue kernel running")
        for k, (t, o, n) in sorted(ov.items(), key=lambda kv: -kv[1][0])[:14]:
            print(f"  {k:70s} {t / n / 1e3:8.1f} us  {100.0 * o / t:5.1f} %")


if __name__ == "__main__":
    main()
