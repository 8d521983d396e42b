#!/usr/bin/env python
"""Kernel statistics (the `--stats` table) from a rocprofv3 rocpd SQLite database: calls, avg/min/max duration, share."""
import sqlite3
import sys


def main(db, out=None):
    con = sqlite3.connect(db)
    rows = con.execute("select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) from kernels "
                       "group by name order by sum(end-start) desc").fetchall()
    tot = sum(r[5] for r in rows) or 1
    lines = ["%-72s %8s %10s %10s %10s %12s %7s" % ("KERNEL", "CALLS", "AVG_us", "MIN_us", "MAX_us", "TOTAL_ms", "PCT")]
    for r in rows:
        lines.append("%-72s %8d %10.3f %10.3f %10.3f %12.3f %6.2f%%" % (r[0][:72], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3,
                                                                      r[5] / 1e6, 100.0 * r[5] / tot))
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    sys.stdout.write(txt)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
