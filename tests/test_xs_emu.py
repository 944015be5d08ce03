"""The exact splitter's KERNELS (fastk_amd/csrc/fk_split_exact.hip: the replay of the reference's Distribute_Block,
split.c:1016-1393) run on the CPU from their .hip source (tests/csrc/hip_emu.h, tests/csrc/xs_emu.cpp) against the
oracle's restatement of Distribute_Block, record for record: one bucket, with and without the segments that cut long
reads where the reference's state is known, with the register chain of minimizers at its full length and cut to one
entry (the ring walk behind it), and the device's own scan kernels -- the 32-bit tile sums' overflow word among them
(round 6).  A CPU test of index arithmetic and state machines; on the MI355X the same kernels are compared
with the reference's files by tests/test_gpu_parity.py (test_exact_*)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import orc

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SO = os.path.join(HERE, "csrc", "xs_emu.so")


@pytest.fixture(scope="module")
def emu():
    src = os.path.join(HERE, "csrc", "xs_emu.cpp")
    deps = [src, os.path.join(HERE, "csrc", "hip_emu.h")] + [os.path.join(ROOT, "fastk_amd", "csrc", f)
                                                             for f in ("fk_split_exact.hip", "fk_common.h")]
    if not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in deps):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-pthread", "-DFK_HOST_EMU", "-shared", "-fPIC",
                               "-I", os.path.join(ROOT, "fastk_amd", "csrc"), "-I", os.path.join(HERE, "csrc"), "-o", SO, src])
    L = C.CDLL(SO)
    L.emu_xs_scan.restype = C.c_int64
    L.emu_xs_scan.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    L.emu_split_exact.restype = C.c_int64
    L.emu_split_exact.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_int, C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    return L


def test_the_devices_scan_and_its_overflow_word(emu):
    rng = np.random.default_rng(1)
    for n in (1, 255, 4096, 32768, 32769, 100001):
        v = rng.integers(0, 50, size=n).astype(np.uint32)
        out = np.zeros(n + 16, dtype=np.uint64)
        tot = np.zeros(1, dtype=np.uint64)
        assert emu.emu_xs_scan(v.ctypes.data, n, out.ctypes.data, tot.ctypes.data) == 0
        ex = np.concatenate([[0], np.cumsum(v.astype(np.uint64))[:-1]])
        assert np.array_equal(out[:n], ex) and int(tot[0]) == int(v.astype(np.uint64).sum())
    # a tile of 4096 counts whose sum does not fit 32 bits raises the word (the offsets are then not to be used)
    n = 10 * 4096 + 9                                # (above 32,768 counts: the tiled route)
    v = np.full(n, 5, dtype=np.uint32)
    v[4096:8192] = 0x00200000                        # 4096 x 2^21 = 2^33
    out = np.zeros(n + 16, dtype=np.uint64)
    tot = np.zeros(1, dtype=np.uint64)
    assert emu.emu_xs_scan(v.ctypes.data, n, out.ctypes.data, tot.ctypes.data) == 1
    v[4096:8192] = 0x000fffff                        # just below: 4096 x (2^20 - 1) < 2^32
    assert emu.emu_xs_scan(v.ctypes.data, n, out.ctypes.data, tot.ctypes.data) == 0
    assert int(tot[0]) == int(v.astype(np.uint64).sum())


def _reads(k, seed):
    rng = np.random.default_rng(seed)
    genome = rng.integers(0, 4, size=30000)
    reads = []
    for i in range(260):
        L = int(rng.choice([k - 1, k, k + 1, 90, 150, 400, 1500, 5000, 12000]))
        s0 = int(rng.integers(0, len(genome) - L))
        r = genome[s0:s0 + L].copy()
        for j in range(L):
            if rng.random() < 0.003:
                r[j] = rng.integers(0, 4)
        if rng.random() < 0.5:
            r = (3 - r)[::-1].copy()
        s = "".join("acgt"[x] for x in r)
        if rng.random() < 0.3:
            s = s.upper()
        if rng.random() < 0.2 and L > 5:
            a = int(rng.integers(0, L))
            s = s[:a] + "N" * int(rng.integers(1, 60)) + s[a:]
        reads.append(s)
    reads += ["a" * 3000, "ac" * 2500, "acg" * 900, "aacgt" * 700, "t" * (k - 1), "", "n" * 100,
              "acgtn" * 300, ("acgt" * 300 + "n") * 5]
    return orc.block_from_reads(reads)


@pytest.mark.parametrize("k", [21, 40, 51, 64])
def test_exact_splitter_kernels_against_the_oracle(emu, k):
    P = orc.params(k)
    bases, boff = _reads(k, 40 + k)
    exp, einst = orc.distribute(P, bases, boff)
    stride = (P.smer_word + 3) & ~3
    sww = stride // 4
    raw = np.zeros(len(bases) + 128, dtype=np.uint8)
    o = (-raw.ctypes.data) % 16 + 32
    raw[o:o + len(bases)] = bases
    roff = np.ascontiguousarray(boff, dtype=np.int64)
    tran = (C.c_int * 4)(*list(P.tran))
    cap = len(exp) + 1000
    for segments, dq in ((0, 6), (1, 6), (1, 1), (0, 2)):
        out = np.zeros((cap, stride), dtype=np.uint8)
        ninst, nseg = C.c_int64(0), C.c_int64(0)
        ns = emu.emu_split_exact(raw.ctypes.data + o, roff.ctypes.data, len(roff) - 1, k, tran, P.smer_bytes, sww, segments, dq, 0,
                                 out.ctypes.data, cap, C.byref(ninst), C.byref(nseg))
        what = (k, segments, dq)
        assert ns == len(exp), what
        assert ninst.value == einst, what
        if segments:
            assert nseg.value > len(roff) - 1, what              # the long reads were cut
        else:
            assert nseg.value == len(roff) - 1, what
        got = out[:ns, :P.smer_word]
        assert np.array_equal(got, exp), what
