#!/usr/bin/env python3
"""Reference FastK (oracle/_ref/FastK) on the GPU box's host cores: thread-count sweep on the bounded samples
bench.py may use.  python tools/cpu_baseline_sweep.py [genome_mbp ...]   (default 20 200)"""
import os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import orc

L, cov, err, k, cutoff, seed = 15000, 50, 2000, 40, 4, 20251001
for mbp in [float(x) for x in (sys.argv[1:] or ["20", "200"])]:
    glen = int(mbp * 1e6)
    nreads = int(cov * glen / L)
    d = tempfile.mkdtemp(prefix="fksweep", dir=os.environ.get("FK_SWEEP_TMP"))
    path = os.path.join(d, "s.fasta")
    t0 = time.perf_counter()
    with open(path, "wb") as f:
        for r0 in range(0, nreads, 20000):
            n = min(20000, nreads - r0)
            b, _ = orc.synth_block(seed, glen, L, err, r0, n)
            mat = np.empty((n, 3 + L + 1), dtype=np.uint8)
            mat[:, 0:3] = np.frombuffer(b">r\n", dtype=np.uint8)
            mat[:, 3:3 + L] = b.reshape(n, L + 1)[:, :L]
            mat[:, 3 + L] = ord("\n")
            mat.tofile(f)
    print("%g Mbp: %d reads, file written in %.1f s" % (mbp, nreads, time.perf_counter() - t0), flush=True)
    inst = nreads * (L - k + 1)
    for T in (16, 32, 64, 128, 256):
        if T > (os.cpu_count() or 1):
            continue
        for rep in range(2):
            t0 = time.perf_counter()
            subprocess.run([os.path.join(orc.REF_DIR, "FastK"), "-k%d" % k, "-t%d" % cutoff, "-T%d" % T, "-P" + d, path],
                           check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=d)
            dt = time.perf_counter() - t0
            print("  -T%-3d rep %d: %.2f s = %.3f G k-mers/s" % (T, rep, dt, inst / dt / 1e9), flush=True)
    subprocess.run(["rm", "-rf", d])
