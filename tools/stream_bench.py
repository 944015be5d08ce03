#!/usr/bin/env python3
"""HBM-budgeted run on a data set that is generated on the device piece by piece and pushed with
fk_push_device: chunked ingest + bucket streaming (BASELINE configs[2]-shaped: long reads, -t4).
Checks the conservation law  sum_c c*hist[c] (c < 0x7fff) + max_inst == k-mer instances  and, with
--compare, equality with the all-resident single-bucket run.

  python tools/stream_bench.py --genome-mbp 200 --coverage 50 --read-len 15000 --err-ppm 2000 \
         --budget-gb 32 --buckets 4 --compare
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastk_amd                                                     # noqa: E402


def run(args, budget_gb, buckets):
    glen = int(args.genome_mbp * 1e6)
    L = args.read_len
    nreads = int(args.coverage * glen / L)
    piece = max(1, int(args.piece_mb * 1e6) // (L + 1))              # reads generated per push
    with fastk_amd.Context(kmer=args.kmer, table_cutoff=args.cutoff, nthreads=4, nbuckets=buckets,
                           hbm_budget=int(budget_gb * 1e9)) as ctx:
        if args.verbose:
            ctx.debug_set("verbose", 1)
        buf = ctx.alloc(piece * (L + 1) + 64)
        t0 = time.perf_counter()
        t_gen = 0.0
        for first in range(0, nreads, piece):
            n = min(piece, nreads - first)
            g0 = time.perf_counter()
            ctx.synth_reads(args.seed, glen, L, args.err_ppm, first, n, buf=buf)
            t_gen += time.perf_counter() - g0
            ctx.push_device(buf.ptr, n * (L + 1) - 1)                # push adds the last terminator
        t_push = time.perf_counter() - t0
        res = ctx.finish()
        dt = time.perf_counter() - t0
        buf.free()
    inst = nreads * (L - args.kmer + 1)
    h = res.hist.astype(np.int64)
    conserved = int((h[1:0x7fff] * np.arange(1, 0x7fff)).sum()) + int(res.max_inst)
    return dict(budget_gb=budget_gb, buckets=buckets, seconds=dt, synth_seconds=t_gen, push_seconds=t_push,
                kmers_per_s=inst / (dt - t_gen), ninst=int(res.ninst), expected_inst=inst,
                conserved=conserved, nsuper=int(res.nsuper), nweighted=int(res.nweighted),
                ndistinct=int(res.ndistinct), ntable=int(res.ntable), device_ms=res.ms), res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome-mbp", type=float, default=200.0)
    ap.add_argument("--coverage", type=float, default=50.0)
    ap.add_argument("--read-len", type=int, default=15000)
    ap.add_argument("--err-ppm", type=int, default=2000)
    ap.add_argument("--kmer", type=int, default=40)
    ap.add_argument("--cutoff", type=int, default=4)
    ap.add_argument("--seed", type=int, default=20251001)
    ap.add_argument("--budget-gb", type=float, default=32.0)
    ap.add_argument("--buckets", type=int, default=4)
    ap.add_argument("--piece-mb", type=float, default=512.0)
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--compare", action="store_true", help="also run all-resident and compare")
    args = ap.parse_args()
    out, res = run(args, args.budget_gb, args.buckets)
    assert out["ninst"] == out["expected_inst"], out
    assert out["conserved"] == out["ninst"], out
    if args.compare:
        ref, rres = run(args, 0, 1)
        assert np.array_equal(res.hist, rres.hist) and res.max_inst == rres.max_inst
        assert res.ntable == rres.ntable and np.array_equal(res.table, rres.table)
        out["resident_seconds"] = ref["seconds"] - ref["synth_seconds"]
        out["resident_device_ms"] = ref["device_ms"]
        out["equal_to_resident_run"] = True
    print(json.dumps(out))


if __name__ == "__main__":
    main()
