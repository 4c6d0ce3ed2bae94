#!/usr/bin/env python3
"""Summarise rocprofv3 (ROCm 7.2 rocpd sqlite) outputs: per-kernel counter sums / means and
kernel durations.  usage: rocpd_summary.py <dir-or-db> [kernel-substring]"""
import glob
import os
import sqlite3
import sys
from collections import defaultdict

root = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
dbs = [root] if root.endswith(".db") else sorted(glob.glob(os.path.join(root, "**", "*.db"), recursive=True))
for p in dbs:
    c = sqlite3.connect(p)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    print("==", os.path.relpath(p, root) if os.path.isdir(root) else p)
    try:
        rows = c.execute("select name, count(*), avg(end-start), min(end-start), max(end-start) from kernels group by name").fetchall()
        for name, n, avg, mn, mx in rows:
            if flt in name:
                print("  kernel %-70s calls %4d  avg %9.2f us  min %9.2f  max %9.2f" % (name[:70], n, avg / 1e3, mn / 1e3, mx / 1e3))
    except Exception as e:  # noqa
        print("  (no kernels view)", e, cols)
    try:
        ccols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
        namecol = "kernel_name" if "kernel_name" in ccols else "name"
        agg = defaultdict(lambda: defaultdict(list))
        for kn, cn, val in c.execute(f"select {namecol}, counter_name, value from counters_collection"):
            agg[kn][cn].append(val)
        for kn, d in agg.items():
            if flt in kn:
                print("  counters for", kn[:80])
                for cn, vals in sorted(d.items()):
                    print("      %-26s mean/dispatch %14.4g   (n=%d)" % (cn, sum(vals) / len(vals), len(vals)))
    except Exception as e:  # noqa
        print("  (no counters)", e)
