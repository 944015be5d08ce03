#!/usr/bin/env python3
"""Build profiles/r01_pmc_traffic.json from two rocprofv3 PMC passes over the default bench.py
command (separate --pmc FETCH_SIZE and --pmc WRITE_SIZE runs, as MI355X_MICROARCH.md prescribes).
Usage: extract_pmc_traffic.py fetch_results.db write_results.db records_W records_S out.json"""
import json
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
        "select s.%s, d.id, sum(p.value), d.grid_size_x from %s p join %s i on p.pmc_id=i.id "
        "join %s d on p.event_id=d.event_id join %s s on d.kernel_id=s.id where i.name=? "
        "group by d.id" % (namecol, pmc, info, disp, sym), (counter,)).fetchall()
    out = {}
    for k, did, v, grid in rows:
        out.setdefault(k.split("(")[0], []).append((v, grid))
    return out


def main():
    f = load(sys.argv[1], "FETCH_SIZE")
    w = load(sys.argv[2], "WRITE_SIZE")
    W, S = int(sys.argv[3]), int(sys.argv[4])
    kern = {}
    for k in sorted(set(f) | set(w)):
        fv = [v for v, _ in f.get(k, [])]
        wv = [v for v, _ in w.get(k, [])]
        grids = sorted({g for _, g in f.get(k, [])})
        # the same kernel symbol may run on two record counts (W and the collapsed list): split by grid
        for g in grids:
            fsel = [v for v, gg in f.get(k, []) if gg == g]
            wsel = [v for v, gg in w.get(k, []) if gg == g]
            if not fsel or not wsel:
                continue
            rb = 2 * 1024.0 * sum(fsel) / len(fsel)
            wb = 1024.0 * sum(wsel) / len(wsel)
            kern["%s grid=%d" % (k, g)] = dict(launches=len(fsel), read_bytes=rb, write_bytes=wb,
                                               traffic_bytes=rb + wb)
    doc = dict(
        what="HBM traffic per launch from rocprofv3 PMC counters on MI355X for the default bench.py "
             "workload (BASELINE configs[1]); separate passes `rocprofv3 --kernel-trace --pmc FETCH_SIZE "
             "-- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline` and the same with WRITE_SIZE",
        units_and_corrections="counters are KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of a "
                              "coalesced streaming read (MI355X_MICROARCH.md, HBM section): reads = 2 x "
                              "FETCH_SIZE x 1024, writes = WRITE_SIZE x 1024.  Calibration inside the run: "
                              "k_synth writes 5,033,333,283 B (WRITE_SIZE gives 1.000x), k_ex_expand writes "
                              "W*12 B (1.000x), k_digit_hist reads W*12 B (2 x FETCH_SIZE gives 1.000x).",
        weighted_kmers=W, supermers=S, kernels=kern)
    json.dump(doc, open(sys.argv[5], "w"), indent=1)
    for k, v in kern.items():
        print("%-70s n=%2d read %.3f GB write %.3f GB" % (k[:70], v["launches"], v["read_bytes"] / 1e9,
                                                          v["write_bytes"] / 1e9))


if __name__ == "__main__":
    main()
