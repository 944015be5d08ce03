"""Time the profile stage (fk_make_profiles) on synthetic reads resident in HBM.
   python tools/profile_bench.py [coverage=50] [genome=100000000] [k=40] [err_ppm=1000]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastk_amd  # noqa: E402

cov = int(sys.argv[1]) if len(sys.argv) > 1 else 50
G = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
k = int(sys.argv[3]) if len(sys.argv) > 3 else 40
err = int(sys.argv[4]) if len(sys.argv) > 4 else 1000
L = 150
nreads = cov * G // L
with fastk_amd.Context(kmer=k, table_cutoff=1) as ctx:
    buf, nbytes = ctx.synth_reads(20240607, G, L, err, 0, nreads)
    res = ctx.count_device_reads(buf.ptr, nbytes, fetch_table=False)
    print("count: %.1f ms, %d table entries" % (res.ms["total"], res.ntable))
    for _ in range(2):
        t0 = time.time()
        data, offs = ctx.make_profiles(buf.ptr, nbytes)
        t1 = time.time()
        print("profiles: %d reads, %.2f GB encoded, %.3f s wall (incl. D2H)" %
              (len(offs) - 1, len(data) / 1e9, t1 - t0))
