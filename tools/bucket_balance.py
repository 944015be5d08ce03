import sys; sys.path.insert(0,'.')
import fastk_amd, numpy as np
with fastk_amd.Context(kmer=40, nbuckets=8) as ctx:
    buf, n = ctx.synth_reads(20251001, 20_000_000, 150, 1000, 0, 6_666_666)
    ns, ni, counts = ctx.split(buf.ptr, n)
    c = np.array(counts, dtype=float); print("bucket balance nb=8:", (c / c.mean()).round(3))
