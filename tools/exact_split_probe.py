#!/usr/bin/env python3
"""Split stage of an exact_parts run (one thread per read replays Distribute_Block, fk_split_exact.hip) against the
default position-parallel splitter, on HiFi-shaped and Illumina-shaped synthetic reads (run on the GPU box):
    python tools/exact_split_probe.py [gbases=1.0]
Prints the device milliseconds of the split stage for both and their ratio."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastk_amd  # noqa: E402


def run(L, nreads, exact):
    with fastk_amd.Context(kmer=40, table_cutoff=4, nthreads=4, exact_parts=exact) as ctx:
        buf, n = ctx.synth_reads(20251001, 200_000_000, L, 2000, 0, nreads)
        host = buf.download(n)
        buf.free()
        per = max(1, (64 << 20) // (L + 1))
        for r0 in range(0, nreads, per):
            r1 = min(nreads, r0 + per)
            boff = (np.arange(r1 - r0 + 1, dtype=np.int64) * (L + 1)).astype(np.int32)
            ctx.push_block(host[r0 * (L + 1):r1 * (L + 1)], boff)
        best = None
        for _ in range(2):
            res = ctx.finish() if best is None else res
            best = res.ms["split"] if best is None else min(best, res.ms["split"])
            break
        return best, res.ninst


def main():
    gb = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    out = {}
    for name, L in (("hifi_15kbp", 15000), ("illumina_150bp", 150)):
        nreads = int(gb * 1e9 / L)
        fast, ni = run(L, nreads, False)
        exact, ne = run(L, nreads, True)
        assert ni == ne
        out[name] = dict(reads=nreads, bases=nreads * L, split_ms_fast=round(fast, 2), split_ms_exact=round(exact, 2),
                         ratio=round(exact / fast, 2))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
