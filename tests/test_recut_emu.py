"""The KERNELS of round 6's k-mer stage, run on the CPU from their .hip sources, against the oracle.

tests/csrc/hip_emu.h is just enough of the HIP kernel language to execute a kernel's source with host threads (one
per work-item, a workgroup at a time); tests/csrc/recut_emu.cpp compiles fastk_amd/csrc/fk_recut.hip and
fk_expand.hip with it.  What is run: k_recut, k_ref_count, k_exscan_tiles, k_ex_expand<.., REF> and k_ref_bounds --
the device code itself, not a restatement -- on the distinct super-mers of read sets with both strands, errors,
homopolymers, tandem repeats, reads of k ... k + 2 and one read 40,000 times:

  * the references k_recut writes are the pieces tests/test_recut_cpu.py's restatement cuts (as a set);
  * the W records k_ex_expand<REF> writes for the sorted references are, as a multiset, the records the ORACLE's
    kmer_list_thread restatement (count.c:339-542) makes from the same super-mers, clipped multiplicities included;
  * the records of one canonical k-mer lie inside one key group, and k_ref_bounds' fills begin at group boundaries,
    are monotone, cover [0, W] and keep to target + the largest group.

This is a CPU test of index arithmetic and layouts; the same stage was checked on the MI355X against the oracle
(profiles/r06_b_*) before the GPU pool closed for the round, and tests/test_gpu_parity.py holds that test."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import orc
from tests import test_recut_cpu as R

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SO = os.path.join(HERE, "csrc", "recut_emu.so")


@pytest.fixture(scope="module")
def emu():
    src = os.path.join(HERE, "csrc", "recut_emu.cpp")
    deps = [src, os.path.join(HERE, "csrc", "hip_emu.h")] + [os.path.join(ROOT, "fastk_amd", "csrc", f)
                                                             for f in ("fk_recut.hip", "fk_expand.hip", "fk_common.h")]
    if not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in deps):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-pthread", "-DFK_HOST_EMU", "-shared", "-fPIC",
                               "-I", os.path.join(ROOT, "fastk_amd", "csrc"), "-I", os.path.join(HERE, "csrc"), "-o", SO, src])
    L = C.CDLL(SO)
    L.emu_recut.restype = C.c_int64
    L.emu_recut.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    L.emu_ref_offsets.restype = C.c_int64
    L.emu_ref_offsets.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
    L.emu_expand_refs.restype = C.c_int64
    L.emu_expand_refs.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p,
                                  C.c_void_p, C.c_int]
    L.emu_ref_bounds.restype = None
    L.emu_ref_bounds.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64]
    return L


def _reads(k, seed):
    rng = np.random.default_rng(seed)
    genome = rng.integers(0, 4, size=8000)
    reads = []
    for _ in range(260):
        L = int(rng.choice([k, k + 1, k + 2, 90, 150, 400]))
        s0 = int(rng.integers(0, len(genome) - L))
        r = genome[s0:s0 + L].copy()
        for j in range(L):
            if rng.random() < 0.005:
                r[j] = rng.integers(0, 4)
        if rng.random() < 0.5:
            r = (3 - r)[::-1]
        reads.append("".join("acgt"[x] for x in r))
    unit = "".join("acgt"[x] for x in genome[300:300 + 2 * k + 11])
    reads += ["a" * 200] * 3 + ["ac" * 90] * 3 + ["aacgt" * 40] * 2
    reads += [unit] * 40000                                       # multiplicities beyond 0x7fff (count.c:455-458)
    return orc.block_from_reads(reads)


def _bases_of(rec, nbases):
    """the first nbases 2-bit codes of a record's byte string (MSB first)"""
    bits = np.unpackbits(rec)
    return (bits[0:2 * nbases:2] * 2 + bits[1:2 * nbases:2]).astype(np.int64)


@pytest.mark.parametrize("k", [32, 40, 51, 56])
def test_the_kernels_of_the_reference_stage_on_the_cpu(emu, k):
    P = orc.params(k)
    bases, boff = _reads(k, 900 + k)
    smers, _ = orc.distribute(P, bases, boff)
    ss = orc.msd_sort(smers, P.smer_word)
    w_ref, ovf_ref, nd_ref = orc.kmer_list(P, ss)

    # the de-duplicated records as fkx_dedup_supermers leaves them: the record at the device stride + a dword with its
    # multiplicity, in no particular order
    stride = (P.smer_word + 3) & ~3
    rw = stride // 4
    uniq, counts = np.unique(smers, axis=0, return_counts=True)
    assert len(uniq) == nd_ref and counts.max() > 0x7fff
    perm = np.random.default_rng(k).permutation(len(uniq))
    uniq, counts = uniq[perm], counts[perm]
    n = len(uniq)
    dd = np.zeros((n, stride + 4), dtype=np.uint8)
    dd[:, :P.smer_word] = uniq
    dd[:, stride:] = counts.astype("<u4").view(np.uint8).reshape(n, 4)
    dd = np.ascontiguousarray(dd)

    # ---- k_recut
    cap = 3 * n + 65536
    refs = np.zeros(cap, dtype=np.uint64)
    flags = C.c_int64(0)
    nref = emu.emu_recut(rw, dd.ctypes.data, n, k, P.smer_bytes, refs.ctypes.data, cap, C.byref(flags))
    assert nref > 0 and flags.value == 0
    refs = refs[:nref]
    want = []
    for i in range(n):
        nk = int(uniq[i, P.smer_bytes]) + 1
        seq = _bases_of(uniq[i, :P.smer_bytes], nk + k - 1)
        M = R.kmer_M(seq, k)
        starts = [0] + [j for j in range(1, nk) if M[j] != M[j - 1]]
        for a, b in zip(starts, starts[1:] + [nk]):
            want.append((int(R.key_of(M[a:a + 1])[0]) << 42) | (i << 14) | (a << 7) | (b - a))
    assert np.array_equal(np.sort(refs), np.sort(np.array(want, dtype=np.uint64)))

    # ---- the sort the radix engine runs on bytes 5, 6, 7 (stable)
    refs = refs[np.argsort(refs >> np.uint64(40), kind="stable")]
    refs = np.ascontiguousarray(refs)

    # ---- k_ref_count + k_exscan_tiles, k_ex_expand<REF>
    ntiles = (nref + 511) // 512
    koff = np.zeros(ntiles + 1, dtype=np.uint64)
    W = emu.emu_ref_offsets(refs.ctypes.data, nref, koff.ctypes.data)
    assert W == len(w_ref) == int((refs & np.uint64(127)).sum())
    kstride = (P.kmer_word + 3) & ~3
    kn, ow = (2 * k + 31) // 32, kstride // 4
    out = np.zeros((W, kstride), dtype=np.uint8)
    ovf = emu.emu_expand_refs(rw, kn, ow, dd.ctypes.data, refs.ctypes.data, nref, k, P.smer_bytes, koff.ctypes.data,
                              out.ctypes.data, P.kmer_bytes)
    assert ovf >= 0, "widths (%d, %d, %d) are not in tests/csrc/recut_emu.cpp" % (rw, kn, ow)
    assert ovf == ovf_ref
    got = np.zeros((W, P.kmer_word), dtype=np.uint8)
    got[:, :P.kmer_bytes] = out[:, :P.kmer_bytes]
    got[:, P.kmer_bytes:] = out[:, kstride - 2:]
    order = lambda a: a[np.lexsort(a.T[::-1])]
    assert np.array_equal(order(got), order(w_ref)), "the W records are not the oracle's"

    # ---- grouped: the records of one k-mer lie inside one key group
    keys_of_ref = (refs >> np.uint64(42)).astype(np.int64)
    nk_of_ref = (refs & np.uint64(127)).astype(np.int64)
    key_of_rec = np.repeat(keys_of_ref, nk_of_ref)
    assert np.all(np.diff(key_of_rec) >= 0)
    seen = {}
    for x, kk in zip(map(bytes, got[:, :P.kmer_bytes]), key_of_rec):
        assert seen.setdefault(x, kk) == kk

    # ---- k_ref_bounds
    gstart = np.flatnonzero(np.concatenate([[True], np.diff(key_of_rec) != 0]))
    gsize = np.diff(np.concatenate([gstart, [W]]))
    for target in (7680, 96, 7):
        nf = (W + target - 1) // target
        bounds = np.zeros(nf + 1, dtype=np.uint64)
        emu.emu_ref_bounds(refs.ctypes.data, nref, koff.ctypes.data, W, target, bounds.ctypes.data, nf)
        b = bounds.astype(np.int64)
        assert b[0] == 0 and b[-1] == W and np.all(np.diff(b) >= 0)
        assert np.all(np.isin(b[:-1], gstart) | (b[:-1] == W))
        for f in range(nf):
            nxt = gstart[gstart >= f * target]
            assert b[f] == (nxt[0] if len(nxt) else W)
        assert np.diff(b).max() <= target + gsize.max()
