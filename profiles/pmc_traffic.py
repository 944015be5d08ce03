#!/usr/bin/env python3
"""HBM traffic per kernel launch from two rocprofv3 PMC passes (separate --pmc FETCH_SIZE and --pmc
WRITE_SIZE runs, as MI355X_MICROARCH.md prescribes) over one bench.py command.
Usage: pmc_traffic.py fetch.db write.db bench_line.json out.json

Units and gfx950 corrections (guide, HBM section): the counters are KiB; FETCH_SIZE reports exactly half
of the bytes of a wide coalesced streaming read, so reads = 2 x FETCH_SIZE x 1024; writes = WRITE_SIZE x
1024.  Launches of one kernel symbol are grouped by grid size (the same kernel runs on buckets of
different record counts); per group: launches, mean read / write / total bytes per launch.  For the
graded scatter kernel the per-launch traffic is also put next to the algorithmic 2*n*R of the bench
line (records_per_launch = the bucket average)."""
import hashlib
import json
import os
import sqlite3
import sys


def load(path, counter):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    t = lambda p: [x for x in tabs if x.startswith(p)][0]
    pmc, info, disp, sym = t("rocpd_pmc_event"), t("rocpd_info_pmc"), t("rocpd_kernel_dispatch"), \
        t("rocpd_info_kernel_symbol")
    cols = [r[1] for r in cur.execute("pragma table_info(%s)" % sym)]
    namecol = "kernel_name" if "kernel_name" in cols else "display_name"
    rows = cur.execute(
        "select s.%s, d.id, sum(p.value) from %s p join %s i on p.pmc_id=i.id "
        "join %s d on p.event_id=d.event_id join %s s on d.kernel_id=s.id where i.name=? "
        "group by d.id" % (namecol, pmc, info, disp, sym), (counter,)).fetchall()
    out = {}
    for k, did, v in rows:
        out.setdefault(k.split("(")[0], []).append(v)
    return out


def main():
    f = load(sys.argv[1], "FETCH_SIZE")
    w = load(sys.argv[2], "WRITE_SIZE")
    line = json.loads([x for x in open(sys.argv[3]).read().splitlines() if x.startswith("{")][-1])
    kern = {}
    for k in sorted(set(f) | set(w)):
        fv, wv = f.get(k, []), w.get(k, [])
        if not fv or not wv:
            continue
        rb = 2 * 1024.0 * sum(fv)
        wb = 1024.0 * sum(wv)
        kern[k] = dict(launches=len(fv), read_bytes_total=rb, write_bytes_total=wb,
                       traffic_bytes_per_launch=(rb + wb) / len(fv))
    roof = line["roofline"]
    cfg = line["config"]
    scat = [v for k, v in kern.items() if "k_rx_scatter" in k and "Li3ELi12ELb1" in k.replace(" ", "").replace("<3,12,true>", "Li3ELi12ELb1")]
    if not scat:
        scat = [v for k, v in kern.items() if "k_rx_scatter<3, 12, true>" in k or "k_rx_scatterILi3ELi12ELb1" in k]
    doc = dict(
        what="HBM traffic from rocprofv3 PMC counters on MI355X for the bench.py workload below; separate passes "
             "`rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python3 bench.py --steps 1 --warmup 0 ...` and the same "
             "with WRITE_SIZE (tools/collect_profiles.sh)",
        units_and_corrections="counters are KiB; gfx950: reads = 2 x FETCH_SIZE x 1024 (FETCH_SIZE reports half of a "
                              "coalesced streaming read, MI355X_MICROARCH.md HBM section), writes = WRITE_SIZE x 1024",
        workload=cfg["workload"], weighted_kmers=cfg["weighted_kmers"], supermers=cfg["supermers"],
        records_per_launch=roof["records_per_launch"], algorithmic_bytes_per_launch=roof["algorithmic_bytes"],
        # bench.py reports this file's traffic only while the scatter kernel's source is the one it was measured on
        kernel_source="fastk_amd/csrc/fk_radix.hip",
        kernel_source_sha256=hashlib.sha256(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                             "fastk_amd", "csrc", "fk_radix.hip"), "rb").read()).hexdigest(),
        kernels=kern)
    if scat:
        tot = sum(v["read_bytes_total"] + v["write_bytes_total"] for v in scat)
        n = sum(v["launches"] for v in scat)
        doc["scatter_traffic_bytes_per_launch"] = tot / n
        doc["scatter_traffic_over_algorithmic"] = round(tot / n / roof["algorithmic_bytes"], 4)
    json.dump(doc, open(sys.argv[4], "w"), indent=1)
    for k, v in sorted(kern.items(), key=lambda kv: -kv[1]["read_bytes_total"] - kv[1]["write_bytes_total"])[:25]:
        print("%-60s n=%4d read %9.3f GB write %9.3f GB" % (k[:60], v["launches"], v["read_bytes_total"] / 1e9,
                                                           v["write_bytes_total"] / 1e9))
    print("scatter traffic / algorithmic:", doc.get("scatter_traffic_over_algorithmic"))


if __name__ == "__main__":
    main()
