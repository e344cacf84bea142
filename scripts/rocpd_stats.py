#!/usr/bin/env python
"""Per-kernel summary (calls, total/avg/min/max duration, share) from a rocprofv3 rocpd SQLite
database -- the same table `rocprofv3 --stats` prints, for runs whose output format was rocpd."""
import sqlite3
import sys


def main(path, out=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = list(cur.execute("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                            "from kernels group by %s order by 3 desc" % (name_col, name_col)))
    total = sum(r[2] for r in rows) or 1
    lines = ["Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage"]
    for n, c, t, a, mn, mx in rows:
        lines.append('"%s",%d,%d,%.1f,%d,%d,%.2f' % (n, c, t, a, mn, mx, 100.0 * t / total))
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
