#!/usr/bin/env python
"""Effective clock and MFMA busy fraction per kernel from ONE rocprofv3 pass with
--pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES (+ --kernel-trace): joins the counters with the dispatch durations.
usage: python tools/pmc_clock.py <results.db> [kernel-name substring ...]"""
import sqlite3
import sys
from collections import defaultdict


def main():
    db = sqlite3.connect(sys.argv[1])
    flts = sys.argv[2:]
    views = [r[0] for r in db.execute("select name from sqlite_master where type in ('view','table')")]
    view = [v for v in views if v.startswith("counters_collection")][0]
    acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(set); dur = defaultdict(float)
    cols = [r[1] for r in db.execute(f"pragma table_info({view})")]
    has_t = "start" in cols and "end" in cols
    q = f"select kernel_name, counter_name, value, dispatch_id" + (", start, end" if has_t else "") + f" from {view}"
    seen = set()
    for row in db.execute(q):
        k = row[0]
        if flts and not any(f in k for f in flts):
            continue
        acc[k][row[1]] += float(row[2]); n[k].add(row[3])
        if has_t and (k, row[3]) not in seen:
            seen.add((k, row[3])); dur[k] += row[5] - row[4]
    if not has_t:
        for name, s in db.execute("select name, sum(end-start) from kernels group by name"):
            dur[name] = s
    print("| kernel | dispatches | avg us | eff. clock GHz (GRBM_GUI_ACTIVE / time) | MFMA busy (MFMA_BUSY_CYCLES / (GUI_ACTIVE x 1024 SIMDs)) | CU busy |")
    print("|---|---|---|---|---|---|")
    for k in sorted(acc, key=lambda k: -dur[k]):
        c = acc[k]; d = dur[k]
        gui = c.get("GRBM_GUI_ACTIVE", 0.0)
        print(f"| `{k[:70]}` | {len(n[k])} | {d / len(n[k]) / 1e3:.1f} | {gui / d if d else 0:.3f} | "
              f"{c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (gui * 1024) if gui else 0:.3f} | {c.get('SQ_BUSY_CU_CYCLES', 0) / (gui * 256) if gui else 0:.3f} |")
        print("    raw per dispatch:", {x: round(v / len(n[k]), 1) for x, v in c.items()})


if __name__ == "__main__":
    main()
