#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (--kernel-trace) into a per-kernel table:
calls, total / average / min / max duration.  Usage: summarize_rocpd.py results.db [out.csv]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in cur.execute("pragma table_info(%s)" % sym)]
    namecol = "kernel_name" if "kernel_name" in cols else "display_name"
    rows = cur.execute(
        "select s.%s, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), "
        "max(d.end-d.start) from %s d join %s s on d.kernel_id = s.id group by s.%s "
        "order by 3 desc" % (namecol, disp, sym, namecol)).fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["kernel,calls,total_ms,avg_us,min_us,max_us,pct"]
    for name, n, tot, avg, mn, mx in rows:
        short = name.split("(")[0]
        lines.append("%s,%d,%.3f,%.1f,%.1f,%.1f,%.1f" % (short.replace(",", ";"), n, tot / 1e6,
                                                         avg / 1e3, mn / 1e3, mx / 1e3,
                                                         100.0 * tot / total))
    text = "\n".join(lines)
    print(text)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")


if __name__ == "__main__":
    main()
