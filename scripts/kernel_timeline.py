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


if __name__ == "__main__":
    main(*sys.argv[1:])
