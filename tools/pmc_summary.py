#!/usr/bin/env python
"""Per-kernel averages of rocprofv3 --pmc counters from a rocpd sqlite database (rocprofv3 writes *_results.db).
usage: python tools/pmc_summary.py <results.db> [kernel-name substring]"""
import sqlite3
import sys
from collections import defaultdict


def main():
    db = sqlite3.connect(sys.argv[1])
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    views = [r[0] for r in db.execute("select name from sqlite_master where type in ('view','table')")]
    view = [v for v in views if v.startswith("counters_collection")]
    if not view:
        print("no counters_collection view; have:", views); return
    cols = [r[1] for r in db.execute(f"pragma table_info({view[0]})")]
    kcol = "kernel_name" if "kernel_name" in cols else [c for c in cols if "kernel" in c and "name" in c][0]
    ccol = "counter_name" if "counter_name" in cols else [c for c in cols if "counter" in c and "name" in c][0]
    vcol = "value" if "value" in cols else [c for c in cols if "value" in c][0]
    dcol = "dispatch_id" if "dispatch_id" in cols else None
    acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(set)
    q = f"select {kcol}, {ccol}, {vcol}" + (f", {dcol}" if dcol else "") + f" from {view[0]}"
    for row in db.execute(q):
        k, c, v = row[0], row[1], row[2]
        if flt and flt not in k:
            continue
        acc[k][c] += float(v)
        cnt[k].add(row[3] if dcol else len(cnt[k]))
    print("| kernel | dispatches | counter | sum | per dispatch |\n|---|---|---|---|---|")
    for k in sorted(acc, key=lambda k: -sum(acc[k].values())):
        for c, v in acc[k].items():
            n = max(len(cnt[k]), 1)
            print(f"| `{k[:90]}` | {n} | {c} | {v:.4g} | {v / n:.6g} |")


if __name__ == "__main__":
    main()
