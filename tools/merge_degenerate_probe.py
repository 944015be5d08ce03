#!/usr/bin/env python3
"""Fastmerge_amd on degenerate sets of sources (GPU box): one source, the same source twice, an empty table among the
sources, only empty tables.  Expected tables by numpy from the oracle's counts.   python tools/merge_degenerate_probe.py"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402

EXE = os.path.join(ROOT, "fastk_amd", "bin", "FastK_amd")
MRG = os.path.join(ROOT, "fastk_amd", "bin", "Fastmerge_amd")
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def main():
    rng = np.random.default_rng(3)
    k = 40
    kb = orc.params(k).kmer_bytes
    d = tempfile.mkdtemp(prefix="fkmrg")
    genome = bytes(ACGT[rng.integers(0, 4, size=20000)])
    sets = {"a": [genome[i:i + 300] for i in rng.integers(0, 19000, size=400)],
            "b": [genome[i:i + 300] for i in rng.integers(0, 19000, size=300)],
            "e": [genome[:30], genome[100:120]]}
    tabs = {}
    for n, reads in sets.items():
        with open(os.path.join(d, n + ".fasta"), "wb") as f:
            for i, r in enumerate(reads):
                f.write(b">r%d\n" % i + r + b"\n")
        subprocess.run([EXE, "-k%d" % k, "-t1", "-T3", os.path.join(d, n + ".fasta")], check=True, cwd=d,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        bases, boff = orc.block_from_reads(reads)
        tabs[n] = orc.fastk(k, bases, boff, cutoff=1).table
    bad = 0
    for name, srcs in (("one source", ["a"]), ("the same source twice", ["a", "a"]), ("an empty table among the sources", ["a", "e", "b"]),
                       ("an empty table first", ["e", "a"]), ("only empty tables", ["e", "e"]), ("two sources", ["a", "b"])):
        out = os.path.join(d, "out_" + name.replace(" ", "_"))
        p = subprocess.run([MRG, "-ht", "-T2", out] + [os.path.join(d, s) for s in srcs], cwd=d, capture_output=True, text=True)
        if p.returncode != 0:
            print("%-36s rc %d: %s" % (name, p.returncode, (p.stdout + p.stderr)[-300:]))
            bad += 1
            continue
        allrec = np.concatenate([tabs[s] for s in srcs]) if sum(len(tabs[s]) for s in srcs) else np.zeros((0, kb + 2), dtype=np.uint8)
        if len(allrec):
            order = np.lexsort(allrec[:, :kb].T[::-1])
            allrec = allrec[order]
            cnt = allrec[:, kb:kb + 2].copy().view("<u2").ravel().astype(np.int64)
            head = np.ones(len(allrec), dtype=bool)
            head[1:] = np.any(allrec[1:, :kb] != allrec[:-1, :kb], axis=1)
            tot = np.bincount(np.cumsum(head) - 1, weights=cnt).astype(np.int64)
            exp = allrec[head].copy()
            exp[:, kb:kb + 2] = np.minimum(tot, 0x7fff).astype("<u2").view(np.uint8).reshape(-1, 2)
            exp_hist = np.bincount(np.minimum(tot, 0x7fff), minlength=0x8000)[1:]
        else:
            exp = allrec
            exp_hist = np.zeros(0x7fff, dtype=np.int64)
        got = orc.read_ktab(out)
        h = np.frombuffer(open(out + ".hist", "rb").read()[28:], dtype=np.int64)
        ok = got["nels"] == len(exp) and got["stream_sha256"] == orc.table_stream_sha256(k, exp, got["ibytes"]) and np.array_equal(h, exp_hist)
        print("%-36s %s (%d entries)" % (name, "ok" if ok else "DIFFERENT", got["nels"]))
        bad += 0 if ok else 1
    subprocess.run(["rm", "-rf", d])
    print("differences:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
