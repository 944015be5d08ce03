#!/usr/bin/env python3
"""Records per bucket (= per rank of a sharded run) for the default serpentine deal of the minimizer
ranks and after fk_set_bucket_weights on a 2 MB sample.  python tools/bucket_balance.py [nbuckets]"""
import sys
sys.path.insert(0, '.')
import numpy as np
import fastk_amd

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 8
with fastk_amd.Context(kmer=40, nbuckets=nb) as ctx:
    buf, n = ctx.synth_reads(20251001, 20_000_000, 150, 1000, 0, 6_666_666)
    ns, ni, counts = ctx.split(buf.ptr, n)
    c = np.array(counts, dtype=float)
    print("serpentine deal, super-mers per bucket / mean:", (c / c.mean()).round(3))
    sample = buf.download(2 << 20)
    ctx.set_bucket_weights(ctx.bucket_census(sample))
    ns2, ni2, counts = ctx.split(buf.ptr, n)
    assert ns2 == ns and ni2 == ni
    c = np.array(counts, dtype=float)
    print("trained deal,    super-mers per bucket / mean:", (c / c.mean()).round(3))
