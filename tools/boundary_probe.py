#!/usr/bin/env python3
"""Boundary cases of the library against the oracle (GPU box): extreme k, cutoffs, thread counts, degenerate reads.
    python tools/boundary_probe.py
Prints one line per case; exit code 1 when any differs."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastk_amd  # noqa: E402
from oracle import orc  # noqa: E402

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
bad = 0


def rnd(rng, n):
    return bytes(ACGT[rng.integers(0, 4, size=n)])


def check(name, k, reads, cutoff=1, T=2, modes=(False, True)):
    global bad
    bases, boff = orc.block_from_reads(reads)
    o = orc.fastk(k, bases, boff, cutoff=cutoff, nthreads=T)
    for exact in modes:
        try:
            with fastk_amd.Context(kmer=k, table_cutoff=cutoff, nthreads=T, exact_parts=exact) as ctx:
                if len(reads) > 0:
                    ctx.push_block(bases, boff.astype(np.int32))
                res = ctx.finish()
            ok = np.array_equal(res.hist, o.hist) and res.max_inst == o.max_inst and np.array_equal(res.table, o.table)
            print("%-44s k=%d t=%d T=%d %-7s %s  (%d entries)" % (name, k, cutoff, T, "exact" if exact else "default",
                                                                  "ok" if ok else "MISMATCH", res.ntable))
            bad += 0 if ok else 1
        except Exception as e:
            print("%-44s k=%d t=%d T=%d %-7s ERROR %s" % (name, k, cutoff, T, "exact" if exact else "default", str(e)[:120]))
            bad += 1


def main():
    rng = np.random.default_rng(77)
    genome = rnd(rng, 60000)
    cov = [genome[i:i + 700] for i in rng.integers(0, len(genome) - 700, size=900)]
    for k in (63, 64, 48, 47, 33, 32, 31, 17, 16, 15):
        check("random coverage", k, cov, cutoff=1, T=3)
    for t in (2, 3, 5, 9, 40, 1000):
        check("cutoff sweep", 25, cov, cutoff=t, T=4)
    for T in (1, 5, 16, 64, 128):
        check("thread sweep", 21, cov, cutoff=2, T=T)
    check("no reads", 40, [], T=4)
    check("one read shorter than k", 40, [rnd(rng, 39)], T=4)
    check("one read of exactly k", 40, [rnd(rng, 40)], T=4)
    check("reads of k-1, k, k+1", 40, [rnd(rng, 39), rnd(rng, 40), rnd(rng, 41)], T=1)
    check("empty reads among others", 40, [b"", rnd(rng, 100), b"", b"", rnd(rng, 41), b""], T=2)
    check("all N", 40, [b"N" * 500, b"N" * 39, b"N"], T=2)
    check("N at every k-th position", 21, [bytes(ord("N") if i % 21 == 20 else b"ACGT"[i % 4] for i in range(2000))], T=2)
    check("one valid k-mer between Ns", 21, [b"N" * 30 + rnd(rng, 21) + b"N" * 30], T=2)
    check("lower case and mixed", 33, [rnd(rng, 300).lower(), rnd(rng, 300), rnd(rng, 300).swapcase()], T=2)
    check("IUPAC codes and other bytes", 33, [rnd(rng, 100) + b"RYKMSWBDHVN-*." + rnd(rng, 100)], T=2)
    check("homopolymer 100 kbp", 40, [b"A" * 100000], T=4)
    check("palindromes", 40, [b"ACGT" * 5000, b"AATT" * 5000, b"GATC" * 300], T=4)
    check("one read of 3 Mbp", 40, [rnd(rng, 3000000)], T=8)
    check("same read 40000 times (count > 32767)", 40, [rnd(rng, 60)] * 40000, cutoff=1, T=4)
    check("200000 reads of 41", 40, [rnd(rng, 41) for _ in range(200000)], T=4)
    print("differences:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
