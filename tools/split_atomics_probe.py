#!/usr/bin/env python3
"""Emit pass of the splitter at configs[1] shape (single bucket) under the FK_HACK_* ablations: wall time of
fk_split_supermers_emit (one kernel + one sync).

The FK_HACK_C / FK_HACK_I switches (reserve from C cursors 4 KB apart / spread the instance counters; records
overlap, results wrong) lived in fk_split.hip only for this measurement and were replaced by the real thing
(SplitArgs.lstreams + k_split_compact); without them every line below measures the shipped kernel.  Output of the
run that found the ceiling (MI355X, 5.03 G bases, 244.5 M super-mers, 1.23 M tiles):

    cursors  0 spread-instance-counters 0: 15.06 14.93 14.91 15.07 ms
    cursors  0 spread-instance-counters 1: 15.09 15.02 15.02 14.96 ms
    cursors  1 spread-instance-counters 1: 14.89 15.00 14.89 15.04 ms
    cursors  4 spread-instance-counters 1: 8.54 8.44 8.44 8.44 ms
    cursors 16 spread-instance-counters 1: 8.39 8.39 8.39 8.39 ms
    cursors 64 spread-instance-counters 1: 8.37 8.36 8.37 8.35 ms
    cursors 16 spread-instance-counters 0: 8.39 8.39 8.39 8.39 ms
"""
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
