"""Pin the oracle against the REFERENCE itself, run here: oracle/_ref/FastK (built by
oracle/Makefile from /root/reference).  Skipped where the reference build is absent (e.g. when
neither /root/reference nor a prebuilt oracle/_ref exists)."""
import os
import random

import numpy as np
import pytest

from oracle import orc

pytestmark = pytest.mark.skipif(not orc.have_ref() and not os.path.isdir(orc.REFERENCE_SRC),
                                reason="reference build not available")


@pytest.fixture(scope="module", autouse=True)
def _ref():
    if not orc.have_ref():
        orc.build(ref=True)


def _random_reads(seed, k, n):
    rnd = random.Random(seed)
    g = "".join(rnd.choice("acgt") for _ in range(4000))
    comp = {"a": "t", "c": "g", "g": "c", "t": "a"}
    out = []
    for _ in range(n):
        L = rnd.choice([k - 2, k, k + 3, 90, 150, 700])
        s = rnd.randrange(0, len(g) - L)
        r = g[s:s + L]
        if rnd.random() < 0.5:
            r = "".join(comp[c] for c in reversed(r))
        r = list(r)
        if rnd.random() < 0.15:
            r[rnd.randrange(len(r))] = "N"
        for j in range(len(r)):
            if rnd.random() < 0.004:
                r[j] = rnd.choice("ACGT")
        out.append("".join(r))
    return out


@pytest.mark.parametrize("k,T,cutoff,fmt", [(40, 4, 1, "fasta"), (51, 2, 2, "fastq"),
                                            (25, 5, 1, "fasta"), (64, 3, 1, "fasta")])
def test_files_byte_identical_to_reference(k, T, cutoff, fmt, tmp_path):
    reads = _random_reads(1000 + k, k, 4000)
    bases, boff = orc.block_from_reads(reads)
    d = str(tmp_path)
    path = os.path.join(d, "x." + fmt)
    (orc.write_fasta if fmt == "fasta" else orc.write_fastq)(path, bases, boff)
    orc.run_ref_fastk(path, k, cutoff, T, d)
    res = orc.fastk(k, bases, boff, cutoff=cutoff, nthreads=T)
    od = os.path.join(d, "o")
    os.mkdir(od)
    orc.write_outputs(res, cutoff, T, od, "x")
    for f in ["x.hist", "x.ktab"] + [".x.ktab.%d" % (t + 1) for t in range(T)]:
        assert open(os.path.join(d, f), "rb").read() == open(os.path.join(od, f), "rb").read(), f
    # the reference's own checker accepts the oracle's table
    import subprocess
    out = subprocess.run([os.path.join(orc.REF_DIR, "Tabex"), "-C", os.path.join(od, "x")],
                         capture_output=True, text=True)
    assert "Table is OK" in out.stdout + out.stderr


@pytest.mark.parametrize("rsize,n,nbytes,T", [(12, 50000, 10, 4), (20, 30011, 19, 3), (16, 1000, 5, 1), (12, 7, 10, 4)])
def test_oracle_lsd_engine_equals_reference_lsd_sort(rsize, n, nbytes, T):
    """Sort-engine unit parity against the reference's own LSD_Sort (libfkref.so = LSDsort.c compiled
    where it lies): same records in, same bytes out, ties (partial keys) in input order."""
    if not orc.have_fkref():
        pytest.skip("oracle/_ref/libfkref.so not built (needs the reference sources at build time)")
    rng = np.random.default_rng(rsize * 1000 + n)
    recs = rng.integers(0, 256, size=(n, rsize), dtype=np.uint8)
    recs[:, 1] = rng.integers(0, 3, size=n)
    order = list(range(nbytes - 1, -1, -1))
    assert np.array_equal(orc.lsd_sort(recs, order), orc.ref_lsd_sort(recs, order, T))
