#!/usr/bin/env python3
"""End-to-end wall time (FASTQ file on local disk -> .hist + .ktab files) of fastk_amd/bin/FastK_amd
with the FASTQ text parsed on the GPU, with the host parser (-H), and of the reference FastK built
in oracle/_ref -- the "t = end-to-end incl. FASTQ parse and file writes" figure of SURVEY.md 8(d).

  python tools/e2e_bench.py --genome-mbp 20
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome-mbp", type=float, default=20.0)
    ap.add_argument("--coverage", type=float, default=50.0)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--err-ppm", type=int, default=1000)
    ap.add_argument("--kmer", type=int, default=40)
    ap.add_argument("--seed", type=int, default=20251001)
    ap.add_argument("--profiles", action="store_true", help="also time -p (profiles) against the reference")
    args = ap.parse_args()
    from oracle import orc
    glen = int(args.genome_mbp * 1e6)
    L = args.read_len
    nreads = int(args.coverage * glen / L)
    bases, boff = orc.synth_block(args.seed, glen, L, args.err_ppm, 0, nreads)
    inst = nreads * (L - args.kmer + 1)
    d = tempfile.mkdtemp(prefix="fke2e")
    out = dict(reads=nreads, read_len=L, kmer=args.kmer, kmer_instances=inst)
    try:
        mat = np.empty((nreads, 3 + L + 3 + L + 1), dtype=np.uint8)
        mat[:, 0:3] = np.frombuffer(b"@r\n", dtype=np.uint8)
        mat[:, 3:3 + L] = bases.reshape(nreads, L + 1)[:, :L]
        mat[:, 3 + L:6 + L] = np.frombuffer(b"\n+\n", dtype=np.uint8)
        mat[:, 6 + L:6 + 2 * L] = ord("I")
        mat[:, 6 + 2 * L] = ord("\n")
        path = os.path.join(d, "s.fastq")
        mat.tofile(path)
        out["fastq_bytes"] = int(mat.nbytes)
        del mat
        exe = os.path.join(ROOT, "fastk_amd", "bin", "FastK_amd")
        digests = {}
        for tag, extra in (("gpu_parse", []), ("host_parse", ["-H"])):
            best = None
            for _ in range(2):                     # second run: file in the page cache, GPU warm
                t0 = time.perf_counter()
                subprocess.run([exe, "-k%d" % args.kmer, "-t1", "-T4", "-N" + os.path.join(d, tag)] + extra + [path],
                               check=True)
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            out[tag + "_seconds"] = best
            out[tag + "_kmers_per_s"] = inst / best
            digests[tag] = open(os.path.join(d, tag + ".hist"), "rb").read()
        assert digests["gpu_parse"] == digests["host_parse"]
        ref = os.path.join(orc.REF_DIR, "FastK")
        if os.path.exists(ref):
            cores = os.cpu_count() or 1
            t0 = time.perf_counter()
            subprocess.run([ref, "-k%d" % args.kmer, "-t1", "-T%d" % cores, "-P" + d, "-N" + os.path.join(d, "ref"), path],
                           check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=d)
            dt = time.perf_counter() - t0
            out["reference_seconds"] = dt
            out["reference_cores"] = cores
            out["reference_kmers_per_s"] = inst / dt
            assert open(os.path.join(d, "ref.hist"), "rb").read() == digests["gpu_parse"]
            out["hist_equal_to_reference"] = True
        if args.profiles:                          # the same with -p: .prof + .pidx files as well
            best = None
            for _ in range(2):
                t0 = time.perf_counter()
                subprocess.run([exe, "-k%d" % args.kmer, "-t1", "-T4", "-p", "-N" + os.path.join(d, "gp"), path],
                               check=True)
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            out["profiles_seconds"] = best
            if os.path.exists(ref):
                t0 = time.perf_counter()
                subprocess.run([ref, "-k%d" % args.kmer, "-t1", "-p", "-T%d" % cores, "-P" + d,
                                "-N" + os.path.join(d, "rp"), path],
                               check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=d)
                out["reference_profiles_seconds"] = time.perf_counter() - t0
                # spot check: the first and last 2000 reads decode identically
                _, mine = orc.read_profiles(d, "gp")
                _, theirs = orc.read_profiles(d, "rp")
                assert len(mine) == len(theirs) == nreads
                for i in list(range(2000)) + list(range(nreads - 2000, nreads)):
                    assert orc.profile_decode(mine[i]) == orc.profile_decode(theirs[i]), i
                out["profiles_decode_equal_to_reference"] = True
    finally:
        subprocess.run(["rm", "-rf", d])
    print(json.dumps(out))


if __name__ == "__main__":
    main()
