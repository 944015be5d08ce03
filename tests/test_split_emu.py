"""The DEFAULT splitter's kernel (fastk_amd/csrc/fk_split.hip: k_split -- position-parallel super-mers on 7-mer minimizers,
the product path's replacement of Distribute_Block + Stuff_Seq, split.c:1016-1393,864-989) run on the CPU from its .hip
source (tests/csrc/hip_emu.h, tests/csrc/split_emu.cpp), counting pass and emit pass as fkx_split drives them.

Its super-mers are not the reference's (only the k-mer multiset is invariant), so the check is the one
tests/test_gpu_parity.py::test_split_covers_every_kmer_exactly_once makes on the GPU: the records of all buckets together
hold every canonical k-mer of the reads exactly as often as it occurs -- no k-mer across a read end, none over a base that
is not acgt -- every record has 1 ... k - 6 k-mers, the instance count is right, equal k-mers share a bucket, and the
records of a locus read from either strand are byte-identical."""
import collections
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import orc

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SO = os.path.join(HERE, "csrc", "split_emu.so")
NRANKS = 16384


@pytest.fixture(scope="module")
def emu():
    src = os.path.join(HERE, "csrc", "split_emu.cpp")
    deps = [src, os.path.join(HERE, "csrc", "hip_emu.h")] + [os.path.join(ROOT, "fastk_amd", "csrc", f)
                                                             for f in ("fk_split.hip", "fk_common.h")]
    if not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in deps):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-pthread", "-DFK_HOST_EMU", "-shared", "-fPIC",
                               "-I", os.path.join(ROOT, "fastk_amd", "csrc"), "-I", os.path.join(HERE, "csrc"), "-o", SO, src])
    L = C.CDLL(SO)
    L.emu_split.restype = C.c_int64
    L.emu_split.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64,
                            C.c_void_p, C.POINTER(C.c_int64)]
    return L


COMP = {"a": "t", "c": "g", "g": "c", "t": "a"}


def _canon(s):
    r = "".join(COMP[c] for c in reversed(s))
    return min(s, r)


def _reads(k, seed):
    rng = np.random.default_rng(seed)
    genome = rng.integers(0, 4, size=9000)
    reads = []
    for _ in range(120):
        L = int(rng.choice([k - 1, k, k + 1, 90, 150, 400, 1500, 4100, 9000]))
        L = min(L, len(genome) - 1)
        s0 = int(rng.integers(0, len(genome) - L))
        r = genome[s0:s0 + L].copy()
        for j in range(L):
            if rng.random() < 0.004:
                r[j] = rng.integers(0, 4)
        if rng.random() < 0.5:
            r = (3 - r)[::-1].copy()
        s = "".join("acgt"[x] for x in r)
        if rng.random() < 0.3:
            s = s.upper()
        if rng.random() < 0.25 and L > 5:
            a = int(rng.integers(0, L))
            s = s[:a] + "N" * int(rng.integers(1, 50)) + s[a:]
        reads.append(s)
    reads += ["a" * 700, "ac" * 400, "aacgt" * 200, "t" * (k - 1), "", "n" * 70, "acgtn" * 100]
    return reads


@pytest.mark.parametrize("k,nb", [(21, 1), (40, 1), (40, 5), (51, 3), (64, 2)])
def test_default_splitter_kernel_covers_every_kmer_exactly_once(emu, k, nb):
    P = orc.params(k)
    reads = _reads(k, 70 + k + nb)
    bases, boff = orc.block_from_reads(reads)
    want = collections.Counter()
    for r in reads:
        r = r.lower()
        for j in range(len(r) - k + 1):
            w = r[j:j + k]
            if all(c in "acgt" for c in w):
                want[_canon(w)] += 1
    stride = (P.smer_word + 3) & ~3
    sww = stride // 4
    raw = np.zeros(len(bases) + 256, dtype=np.uint8)
    o = (-raw.ctypes.data) % 16 + 64
    raw[o:o + len(bases)] = bases
    mb = (np.arange(NRANKS) % nb).astype(np.uint8)
    cap = len(bases)
    out = np.zeros((cap, stride), dtype=np.uint8)
    counts = np.zeros(256, dtype=np.int64)
    ninst = C.c_int64(0)
    ns = emu.emu_split(raw.ctypes.data + o, len(bases), k, P.smer_bytes, sww, nb, mb.ctypes.data, out.ctypes.data, cap,
                       counts.ctypes.data, C.byref(ninst))
    assert ns > 0 and ns == int(counts[:nb].sum())
    assert ninst.value == sum(want.values())
    got = collections.Counter()
    bucket_of = {}
    recs = set()
    for i in range(ns):
        rec = out[i]
        n = int(rec[P.smer_bytes]) + 1
        assert 1 <= n <= k - 6
        bits = np.unpackbits(rec[:P.smer_bytes])
        L = n + k - 1
        seq = "".join("acgt"[2 * int(bits[2 * j]) + int(bits[2 * j + 1])] for j in range(L))
        assert not bits[2 * L:].any(), "padding behind the bases is not zero"
        b = int(np.searchsorted(np.cumsum(counts[:nb]), i, side="right"))
        for j in range(n):
            c = _canon(seq[j:j + k])
            got[c] += 1
            assert bucket_of.setdefault(c, b) == b, "a k-mer in two buckets"
        recs.add(bytes(rec[:P.smer_word]))
    assert got == want
    # a locus read from the other strand gives byte-identical records (the record is turned when its minimizer lies on
    # the - strand): the reverse complement of the first long read, alone, yields records that the forward read yields too
    long = max(reads, key=len).lower().replace("n", "")
    rc = "".join(COMP[c] for c in reversed(long))
    for s in (long, rc):
        b2, _ = orc.block_from_reads([s])
        raw2 = np.zeros(len(b2) + 256, dtype=np.uint8)
        o2 = (-raw2.ctypes.data) % 16 + 64
        raw2[o2:o2 + len(b2)] = b2
        out2 = np.zeros((len(b2), stride), dtype=np.uint8)
        n2 = emu.emu_split(raw2.ctypes.data + o2, len(b2), k, P.smer_bytes, sww, 1, np.zeros(NRANKS, dtype=np.uint8).ctypes.data,
                           out2.ctypes.data, len(b2), counts.ctypes.data, C.byref(ninst))
        s_recs = collections.Counter(bytes(x[:P.smer_word]) for x in out2[:n2])
        if s is long:
            fwd = s_recs
    # (tile edges cut the two strands' super-mers at mirrored places, so only records away from the cuts must agree: most do)
    common = sum((fwd & s_recs).values())
    assert common >= 0.8 * sum(fwd.values())
