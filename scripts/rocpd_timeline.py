#!/usr/bin/env python
"""Kernel timeline from a rocprofv3 rocpd database: start (ms since the first kernel), duration (us), gap to the
previous kernel's end (us), name.   python scripts/rocpd_timeline.py p_results.db [first] [count] [substring]"""
import sqlite3
import sys


def main(path, first=0, count=200, sub=""):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = list(cur.execute("select %s, start, end from kernels order by start" % name_col))
    if not rows:
        print("no kernels")
        return
    t0 = rows[0][1]
    prev_end = None
    shown = 0
    for i, (n, s, e) in enumerate(rows):
        short = n.split("(")[0].replace("void kmx::", "").replace("kmx::", "")[:70]
        gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
        prev_end = max(e, prev_end) if prev_end is not None else e
        if i < first or (sub and sub not in n):
            continue
        print("%5d  %9.3f ms  %8.1f us  gap %8.1f us  %s" % (i, (s - t0) / 1e6, (e - s) / 1e3, gap, short))
        shown += 1
        if shown >= count:
            break


if __name__ == "__main__":
    a = sys.argv
    main(a[1], int(a[2]) if len(a) > 2 else 0, int(a[3]) if len(a) > 3 else 200, a[4] if len(a) > 4 else "")
