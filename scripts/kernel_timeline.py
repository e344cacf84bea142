#!/usr/bin/env python
"""The last N kernel launches of a rocprofv3 rocpd database as a timeline: start (us, relative), duration (us), name."""
import sqlite3
import sys


def main(path, n=80):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = list(cur.execute("select start, end, %s from kernels order by start" % name_col))
    rows = rows[-int(n):]
    t0 = rows[0][0]
    for s, e, name in rows:
        short = name.split("(")[0].replace("void kmx::", "").replace("_ZN3kmx", "")[:70]
        print("%10.1f %8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, short))




def summary(path):
    """Busy / idle time of the device between the first and the last launch, and the kernels by total time."""
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = list(cur.execute("select start, end, %s from kernels order by start" % name_col))
    span = rows[-1][1] - rows[0][0]
    busy, cur_end = 0, rows[0][0]
    gaps = []
    for s, e, name in rows:
        if s > cur_end:
            gaps.append((s - cur_end, name))
            busy += e - s
            cur_end = e
        elif e > cur_end:
            busy += e - cur_end
            cur_end = e
    print("launches %d, span %.3f ms, busy %.3f ms, idle %.3f ms" % (len(rows), span / 1e6, busy / 1e6, (span - busy) / 1e6))
    gaps.sort(reverse=True)
    print("largest idle gaps (us, before kernel):")
    for g, name in gaps[:12]:
        print("  %9.1f  %s" % (g / 1e3, name.split("(")[0][:80]))
    tot = {}
    for s, e, name in rows:
        k = name.split("(")[0][:90]
        t = tot.setdefault(k, [0, 0])
        t[0] += e - s
        t[1] += 1
    print("kernels by total time (ms, calls):")
    for k, (t, c) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:14]:
        print("  %9.3f %6d  %s" % (t / 1e6, c, k))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "summary":
        summary(sys.argv[1])
    else:
        main(*sys.argv[1:])
