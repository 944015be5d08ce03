#!/usr/bin/env python3
"""Emit pass of the splitter at configs[1] shape (single bucket) under the FK_HACK_* ablations: wall time of
fk_split_supermers_emit (one kernel + one sync)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastk_amd

G, L, COV = 100_000_000, 150, 50
nreads = G * COV // L
with fastk_amd.Context(kmer=40, table_cutoff=1) as ctx:
    rd, n = ctx.synth_reads(1234, G, L, 1000, 0, nreads)
    ns, ni, bc = ctx.split(rd.ptr, n)
    print("bases %d super-mers %d instances %d" % (n, ns, ni))
    out = ctx.alloc(ns * ctx.w.smer_stride + 4096)
    for c, i in ((0, 0), (0, 1), (1, 1), (4, 1), (16, 1), (64, 1), (16, 0), (0, 0)):
        os.environ["FK_HACK_C"] = str(c); os.environ["FK_HACK_I"] = str(i)
        ts = []
        for r in range(4):
            t0 = time.perf_counter()
            ctx.split_emit(rd.ptr, n, out.ptr, ns, bc + [0] * (256 - len(bc)))
            ts.append((time.perf_counter() - t0) * 1e3)
        print("cursors %2d spread-instance-counters %d: %s ms" % (c, i, " ".join("%.2f" % t for t in ts)))
