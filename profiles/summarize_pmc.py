#!/usr/bin/env python3
"""Per-kernel PMC sums from a rocprofv3 rocpd database collected with --pmc.
Usage: summarize_pmc.py results.db [out.csv]"""
import sqlite3
import sys
from collections import defaultdict


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    t = lambda p: [x for x in tabs if x.startswith(p)][0]
    pmc, info, disp, sym = t("rocpd_pmc_event"), t("rocpd_info_pmc"), t("rocpd_kernel_dispatch"), \
        t("rocpd_info_kernel_symbol")
    cols = [r[1] for r in cur.execute("pragma table_info(%s)" % sym)]
    namecol = "kernel_name" if "kernel_name" in cols else "display_name"
    rows = cur.execute(
        "select s.%s, i.name, sum(p.value), count(distinct d.id), sum(distinct d.end-d.start) "
        "from %s p join %s i on p.pmc_id=i.id join %s d on p.event_id=d.event_id "
        "join %s s on d.kernel_id=s.id group by s.%s, i.name" % (namecol, pmc, info, disp, sym, namecol)
    ).fetchall()
    data = defaultdict(dict)
    calls = {}
    for k, c, v, n, dur in rows:
        k = k.split("(")[0]
        data[k][c] = v
        calls[k] = n
    names = sorted({c for d in data.values() for c in d})
    lines = ["kernel,calls," + ",".join(names)]
    for k in sorted(data, key=lambda k: -data[k].get("SQ_WAVE_CYCLES", 0)):
        lines.append(k.replace(",", ";") + ",%d," % calls[k] + ",".join("%.4g" % data[k].get(c, 0) for c in names))
    text = "\n".join(lines)
    print(text)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")


if __name__ == "__main__":
    main()
