#!/usr/bin/env python3
"""HBM traffic per kernel of one end-to-end FastK_amd run from two rocprofv3 PMC passes (--pmc FETCH_SIZE, --pmc
WRITE_SIZE, the binary itself behind `--`; tools/e2e_profile.sh with E2E_PMC=1).
Usage: e2e_traffic.py fetch.db write.db out.json
Units and gfx950 corrections as in pmc_traffic.py: KiB counters, reads = 2 x FETCH_SIZE x 1024, writes = WRITE_SIZE x 1024."""
import json
import sys

from pmc_traffic import load


def main():
    f = load(sys.argv[1], "FETCH_SIZE")
    w = load(sys.argv[2], "WRITE_SIZE")
    kern = {}
    for k in sorted(set(f) | set(w)):
        fv, wv = f.get(k, []), w.get(k, [])
        if not fv or not wv:
            continue
        rb, wb = 2 * 1024.0 * sum(fv), 1024.0 * sum(wv)
        kern[k] = dict(launches=len(fv), read_bytes_total=rb, write_bytes_total=wb)
    doc = dict(what="HBM traffic per kernel of one FastK_amd -k40 -t4 -T32 -M256 run on the configs[2] FASTA file (rocprofv3 "
                    "--kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate runs, the binary itself after `--`)",
               units_and_corrections="counters are KiB; gfx950: reads = 2 x FETCH_SIZE x 1024, writes = WRITE_SIZE x 1024",
               kernels=kern,
               total_read_bytes=sum(v["read_bytes_total"] for v in kern.values()),
               total_write_bytes=sum(v["write_bytes_total"] for v in kern.values()))
    json.dump(doc, open(sys.argv[3], "w"), indent=1)
    for k, v in sorted(kern.items(), key=lambda kv: -kv[1]["read_bytes_total"] - kv[1]["write_bytes_total"])[:20]:
        print("%-60s n=%5d read %9.3f GB write %9.3f GB" % (k[:60], v["launches"], v["read_bytes_total"] / 1e9, v["write_bytes_total"] / 1e9))


if __name__ == "__main__":
    sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
    main()
