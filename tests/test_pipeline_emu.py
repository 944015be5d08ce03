"""The device code of the counting path's own stages, composed on the CPU, against the REFERENCE's golden digests.

Every stage that this library's kernels compute without wave intrinsics beyond a shuffle runs from its .hip source
through tests/csrc/hip_emu.h: the default splitter (k_split: reads -> super-mers), the cut into minimizer domains
(k_recut), the expansion by references (k_ref_count, k_exscan_tiles, k_ex_expand<.., REF>) and the fills
(k_ref_bounds).  The three stages in between that lean on ballots, DPP and inline LDS instructions -- the radix
scatter, the LDS de-duplication, the LDS aggregation -- are stood in for by numpy with the same contracts (identical
records brought together and counted; references ordered by their key bytes, stably; the weights of equal k-mers
summed fill by fill, saturating as MSDsort.c:491-509 does).  The histogram and the sorted table that come out must have
the digests the golden fixtures hold -- the files reference FastK itself wrote (tests/golden/make_golden.py).

Round 6: written after the GPU pool had closed for the round; the same composition ran on the MI355X against the
oracle before that (profiles/r06_b_*) and at full BASELINE configs[2] size against the reference's .hist digest
(profiles/r06_a_*)."""
import ctypes as C

import numpy as np
import pytest

from oracle import orc
from tests import util
from tests.test_recut_emu import emu as recut_emu          # noqa: F401  (fixtures)
from tests.test_split_emu import emu as split_emu, NRANKS   # noqa: F401

split_lib = split_emu
recut_lib = recut_emu


@pytest.mark.parametrize("name,nb,target", [("synth_tiny_k40_t1_T2", 1, 7680), ("edge_k40_t1_T4", 3, 7680),
                                            ("edge_k40_t4_T1", 2, 96), ("edge_k51_t1_T4", 1, 7680)])
def test_split_recut_expand_compose_to_the_references_digests(split_lib, recut_lib, name, nb, target):
    case, bases, boff = util.load_case(name)
    k, cutoff = case["k"], case["cutoff"]
    P = orc.params(k)
    stride = (P.smer_word + 3) & ~3
    rw = stride // 4

    # ---- k_split: reads -> super-mers, grouped by bucket
    raw = np.zeros(len(bases) + 256, dtype=np.uint8)
    o = (-raw.ctypes.data) % 16 + 64
    raw[o:o + len(bases)] = bases
    mb = (np.arange(NRANKS) % nb).astype(np.uint8)
    cap = len(bases)
    smers = np.zeros((cap, stride), dtype=np.uint8)
    counts = np.zeros(256, dtype=np.int64)
    ninst = C.c_int64(0)
    ns = split_lib.emu_split(raw.ctypes.data + o, len(bases), k, P.smer_bytes, rw, nb, mb.ctypes.data, smers.ctypes.data,
                             cap, counts.ctypes.data, C.byref(ninst))
    assert ns > 0
    hist = np.zeros(0x8000, dtype=np.int64)
    max_inst = 0
    table = {}
    first = 0
    for b in range(nb):                                         # the buckets one after the other, as count_bucket takes them
        recs = smers[first:first + counts[b]]
        first += counts[b]
        if len(recs) == 0:
            continue
        # (stand-in for fkx_group + fkx_dedup_supermers: every distinct record once, with its multiplicity)
        uniq, mult = np.unique(recs, axis=0, return_counts=True)
        n = len(uniq)
        dd = np.zeros((n, stride + 4), dtype=np.uint8)
        dd[:, :stride] = uniq
        dd[:, stride:] = mult.astype("<u4").view(np.uint8).reshape(n, 4)
        dd = np.ascontiguousarray(dd)
        # ---- k_recut
        rcap = 3 * n + 65536
        refs = np.zeros(rcap, dtype=np.uint64)
        flags = C.c_int64(0)
        nref = recut_lib.emu_recut(rw, dd.ctypes.data, n, k, P.smer_bytes, refs.ctypes.data, rcap, C.byref(flags))
        assert nref >= n and flags.value == 0
        # (stand-in for fkx_lsd_sort on bytes 5, 6, 7)
        refs = np.ascontiguousarray(refs[:nref][np.argsort(refs[:nref] >> np.uint64(40), kind="stable")])
        # ---- k_ref_count, k_exscan_tiles, k_ex_expand<REF>, k_ref_bounds
        koff = np.zeros((nref + 511) // 512 + 1, dtype=np.uint64)
        W = recut_lib.emu_ref_offsets(refs.ctypes.data, nref, koff.ctypes.data)
        kstride = (P.kmer_word + 3) & ~3
        out = np.zeros((W, kstride), dtype=np.uint8)
        ovf = recut_lib.emu_expand_refs(rw, (2 * k + 31) // 32, kstride // 4, dd.ctypes.data, refs.ctypes.data, nref, k,
                                        P.smer_bytes, koff.ctypes.data, out.ctypes.data, P.kmer_bytes)
        assert ovf >= 0
        max_inst += ovf
        nf = (W + target - 1) // target
        bounds = np.zeros(nf + 1, dtype=np.uint64)
        recut_lib.emu_ref_bounds(refs.ctypes.data, nref, koff.ctypes.data, W, target, bounds.ctypes.data, nf)
        # (stand-in for k_ag_count2: fill by fill -- a k-mer whose records lay in two fills would come out twice)
        keys = out[:, :P.kmer_bytes]
        wgt = np.ascontiguousarray(out[:, kstride - 2:]).view("<u2").ravel().astype(np.int64)
        for f in range(nf):
            lo, hi = int(bounds[f]), int(bounds[f + 1])
            if lo == hi:
                continue
            u, inv = np.unique(keys[lo:hi], axis=0, return_inverse=True)
            cnt = np.bincount(inv.ravel(), weights=wgt[lo:hi], minlength=len(u)).astype(np.int64)
            for kb, c in zip(map(bytes, u), cnt):
                assert kb not in table, "a k-mer met in two fills (or two buckets)"
                table[kb] = int(c)
    for c in table.values():                                    # MSDsort.c:498-506
        if c >= 0x7fff:
            hist[0x7fff] += 1
            max_inst += c
        else:
            hist[c] += 1
    rows = sorted((kb, min(c, 0x7fff)) for kb, c in table.items() if c >= cutoff)
    tab = np.zeros((len(rows), P.kmer_word), dtype=np.uint8)
    for i, (kb, c) in enumerate(rows):
        tab[i, :P.kmer_bytes] = np.frombuffer(kb, dtype=np.uint8)
        tab[i, P.kmer_bytes:] = np.frombuffer(np.uint16(c).tobytes(), dtype=np.uint8)
    util.check_against_golden(case, hist, max_inst, tab)
