#!/usr/bin/env python3
"""Do two contexts that count alternate minimizer buckets on two host threads (two streams) finish sooner
than one context that counts all of them?  1/10 of configs[2], 48 buckets, all super-mers resident."""
import sys, time, threading
import numpy as np
sys.path.insert(0, '.')
import fastk_amd

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
L, glen, k, nb = 15000, int(3e9 * scale), 40, 48
nreads = int(50 * glen / L)
nbytes = nreads * (L + 1)
main = fastk_amd.Context(kmer=k, table_cutoff=4, nbuckets=nb)
buf, _ = main.synth_reads(20251001, glen, L, 2000, 0, nreads)
cap, offs = main.split_plan(buf.ptr, nbytes)
out = main.alloc(cap * main.w.smer_stride)
stride = main.w.smer_stride

def split():
    counts, ni = main.split_planned(buf.ptr, nbytes, out.ptr, cap, offs)
    return counts

def run(ctxs):
    counts = split()
    t0 = time.perf_counter()
    def work(i):
        c = ctxs[i]
        c.rounds_begin()
        for b in range(i, nb, len(ctxs)):
            c.rounds_add(out.ptr + offs[b] * stride, counts[b])
    th = [threading.Thread(target=work, args=(i,)) for i in range(len(ctxs))]
    for t in th: t.start()
    for t in th: t.join()
    res = [c.rounds_finish(fetch_table=False) for c in ctxs]
    dt = time.perf_counter() - t0
    return dt, sum(r.ndistinct for r in res), sum(r.ntable for r in res)

others = [fastk_amd.Context(kmer=k, table_cutoff=4, nbuckets=nb) for _ in range(3)]
for n in (1, 2, 3, 1, 2, 3):
    ctxs = [main] + others[:n - 1]
    dt, nd, nt = run(ctxs)
    print("%d context(s): count phase %.1f ms, distinct %d, table %d" % (n, dt * 1e3, nd, nt), flush=True)
