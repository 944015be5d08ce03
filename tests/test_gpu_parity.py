"""Parity of the HIP path (through the C-ABI, libfastk_amd.so) against the CPU oracle and the
golden vectors captured from the reference.  Needs a real MI355X: run with -m gpu."""
import os
import numpy as np
import pytest

import fastk_amd
from oracle import orc
from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx40():
    c = fastk_amd.Context(kmer=40, table_cutoff=1, nthreads=4)
    yield c
    c.close()


def _pad(recs, stride):
    """reference-width records -> device stride (k-mer count moves to the last two bytes)."""
    n, w = recs.shape
    if w == stride:
        return np.ascontiguousarray(recs)
    out = np.zeros((n, stride), dtype=np.uint8)
    out[:, :w] = recs
    return out


def _pad_kmers(recs, kb, stride):
    n, w = recs.shape
    if w == stride:
        return np.ascontiguousarray(recs)
    out = np.zeros((n, stride), dtype=np.uint8)
    out[:, :kb] = recs[:, :kb]
    out[:, stride - 2:] = recs[:, kb:kb + 2]
    return out


def _unpad_kmers(recs, kb):
    n, s = recs.shape
    if s == kb + 2:
        return recs
    out = np.zeros((n, kb + 2), dtype=np.uint8)
    out[:, :kb] = recs[:, :kb]
    out[:, kb:] = recs[:, s - 2:]
    return out


# ------------------------------------------------------------------------------ radix engine

@pytest.mark.parametrize("rsize,n,seed", [(12, 1000, 1), (12, 300001, 2), (20, 123457, 3),
                                          (16, 50000, 4), (8, 77777, 5), (28, 40000, 6),
                                          (4, 100000, 7), (12, 0, 8), (12, 1, 9)])
def test_lsd_sort_bit_exact(ctx40, rsize, n, seed):
    rng = np.random.default_rng(seed)
    recs = rng.integers(0, 256, size=(n, rsize), dtype=np.uint8)
    if n > 10:
        recs[:, 1] = rng.integers(0, 3, size=n)      # few-valued digit
        recs[:, 2] = 7                                # constant digit (pass is skipped)
        recs[: n // 2] = recs[n // 2: 2 * (n // 2)]   # many duplicates
    byte_list = list(range(min(rsize, 10) - 1, -1, -1))
    a = ctx40.alloc(max(recs.nbytes, 16)).upload(recs)
    b = ctx40.alloc(max(recs.nbytes, 16))
    res = ctx40.lsd_sort(a.ptr, b.ptr, n, rsize, byte_list)
    got = a.download(recs.nbytes, ptr=res).reshape(n, rsize)
    exp = orc.lsd_sort(recs, byte_list)
    assert np.array_equal(got, exp)
    a.free(); b.free()


def test_lsd_sort_is_stable_on_partial_keys(ctx40):
    """Sorting on two bytes only must keep the input order of equal keys (LSDsort.c contract)."""
    rng = np.random.default_rng(11)
    n, rsize = 200000, 12
    recs = rng.integers(0, 256, size=(n, rsize), dtype=np.uint8)
    recs[:, 5] = rng.integers(0, 4, size=n)
    a = ctx40.alloc(recs.nbytes).upload(recs)
    b = ctx40.alloc(recs.nbytes)
    res = ctx40.lsd_sort(a.ptr, b.ptr, n, rsize, [5, 0])
    got = a.download(recs.nbytes, ptr=res).reshape(n, rsize)
    assert np.array_equal(got, orc.lsd_sort(recs, [5, 0]))
    a.free(); b.free()


def test_msd_sort_matches_oracle_msd(ctx40):
    rng = np.random.default_rng(12)
    n, rsize, ksize = 150000, 20, 20
    recs = rng.integers(0, 4, size=(n, rsize), dtype=np.uint8)     # heavy ties
    a = ctx40.alloc(recs.nbytes).upload(recs)
    b = ctx40.alloc(recs.nbytes)
    res = ctx40.msd_sort(a.ptr, b.ptr, n, rsize, ksize)
    got = a.download(recs.nbytes, ptr=res).reshape(n, rsize)
    assert np.array_equal(got, orc.msd_sort(recs, ksize))
    a.free(); b.free()


def test_msd_sort_key_order_equals_reference_engine(ctx40):
    """fk_msd_sort_records against the reference's own Weighted_Kmer_Sort (MSDsort.c:536-544, libfkref.so) on 1.5 M
    weighted 40-mers with heavy duplication: the same key order (the engine leaves a flag in byte 0 of every run head
    and the run's sum in its count field, so the comparison is on the key bytes, first byte restored)."""
    if not orc.have_fkref():
        util.no_reference("oracle/_ref/libfkref.so not built")
    rng = np.random.default_rng(99)
    n, kb = 1500000, 10
    keys = rng.integers(0, 256, size=(n // 5, kb), dtype=np.uint8)
    recs = np.zeros((n, kb + 2), dtype=np.uint8)
    recs[:, :kb] = keys[rng.integers(0, n // 5, size=n)]
    recs[:, kb] = rng.integers(1, 4, size=n)
    a = ctx40.alloc(recs.nbytes).upload(recs)
    b = ctx40.alloc(recs.nbytes)
    res = ctx40.msd_sort(a.ptr, b.ptr, n, kb + 2, kb)
    got = a.download(recs.nbytes, ptr=res).reshape(n, kb + 2)
    ref, hist, _ = orc.ref_weighted_kmer_sort(recs, 40, 8)
    ref_keys = ref[:, :kb].copy()
    ref_keys[:, 0] = np.sort(recs[:, 0], kind="stable")
    assert np.array_equal(got[:, :kb], ref_keys)
    assert int(hist.sum()) == len(np.unique(recs[:, :kb], axis=0))
    a.free(); b.free()


@pytest.mark.parametrize("kmer,what", [(40, "kmers"), (51, "kmers"), (40, "supermers"), (51, "supermers")])
def test_msd_engine_against_both_reference_sorts(kmer, what):
    """fk_msd_sort_records -- ceil(log256 n) radix levels + the LDS finish of the small parts (fk_tsort.hip) -- against
    the reference's own engines (libfkref.so): Weighted_Kmer_Sort on R = 12 / 15-byte weighted k-mers (MSDsort.c:536-544)
    and Supermer_Sort on R = 20 / 26-byte super-mer records (:458-489), 3 M records each: heavy duplication, a cluster
    of keys that agree on their first nine bytes (a part too long for a tile's halo: the compaction route) and a
    constant key byte.  Same key order; and the whole sort takes at most six passes over the array."""
    if not orc.have_fkref():
        pytest.fail("oracle/_ref/libfkref.so is not built (make -C oracle ref needs /root/reference)")
    rng = np.random.default_rng(kmer * 7 + len(what))
    n = 3_000_000
    with fastk_amd.Context(kmer=kmer) as ctx:
        w = ctx.w
        if what == "kmers":
            kb, rs, stride = w.kmer_bytes, w.kmer_word, w.kmer_stride
            keys = rng.integers(0, 256, size=(n // 4, kb), dtype=np.uint8)
            keys[:5000, :9] = keys[0, :9]                        # one long part
            keys[:, 4] = 0x5a                                    # a constant digit
            keys[:, kb - 1] &= (0xff << (2 * (4 * kb - kmer))) & 0xff
            ref_in = np.zeros((n, rs), dtype=np.uint8)
            ref_in[:, :kb] = keys[rng.integers(0, n // 4, size=n)]
            ref_in[:, kb] = rng.integers(1, 4, size=n)
            ksize = kb
            ref, _, _ = orc.ref_weighted_kmer_sort(ref_in, kmer, 8)
        else:
            rs, stride = w.smer_word, w.smer_stride
            sb = w.smer_bytes
            keys = rng.integers(0, 256, size=(n // 4, rs), dtype=np.uint8)
            keys[:5000, :9] = keys[0, :9]
            keys[:, 6] = 0x33
            keys[:, sb:] = 0
            keys[:, rs - 1] = rng.integers(0, kmer - 8, size=n // 4)     # n - 1 of a super-mer
            ref_in = keys[rng.integers(0, n // 4, size=n)]
            ksize = rs
            ref = orc.ref_supermer_sort(ref_in, kmer, 8)
        dev_in = np.zeros((n, stride), dtype=np.uint8)
        dev_in[:, :ksize] = ref_in[:, :ksize]
        if what == "kmers":
            dev_in[:, stride - 2:] = ref_in[:, rs - 2:]
        a = ctx.alloc(dev_in.nbytes).upload(dev_in)
        b = ctx.alloc(dev_in.nbytes)
        res = ctx.msd_sort(a.ptr, b.ptr, n, stride, ksize)
        got = a.download(dev_in.nbytes, ptr=res).reshape(n, stride)
        st = ctx.sort_stats()
        ref_keys = ref[:, :ksize].copy()
        ref_keys[:, 0] = np.sort(ref_in[:, 0], kind="stable")    # (the engine keeps a flag in byte 0 of a run's head)
        assert np.array_equal(got[:, :ksize], ref_keys)
        assert st["passes"] <= 4, st                             # 1 histogram read + <= 4 levels + the LDS finish <= 6
        a.free(); b.free()


# ------------------------------------------------------------------------------ stages

@pytest.mark.parametrize("name", ["edge_k40_t1_T4", "synth_illumina_k40_t1_T4",
                                  "synth_hifi_k40_t4_T8", "edge_k51_t1_T4", "edge_k21_t2_T3"])
def test_split_covers_every_kmer_exactly_once(name):
    """GPU super-mers, finished by the ORACLE's sort/expand/count, must give the reference's
    histogram and table: every valid k-mer instance is in exactly one well-formed record."""
    case, bases, boff = util.load_case(name)
    k = case["k"]
    with fastk_amd.Context(kmer=k, table_cutoff=case["cutoff"]) as ctx:
        w = ctx.w
        rd = ctx.alloc(len(bases) + 64).upload(bases)
        ns, ni, _ = ctx.split(rd.ptr, len(bases))
        out = ctx.alloc(max(ns, 1) * w.smer_stride)
        ns2, ni2, _ = ctx.split(rd.ptr, len(bases), out.ptr, ns)
        assert (ns2, ni2) == (ns, ni)
        recs = out.download(ns * w.smer_stride).reshape(ns, w.smer_stride)[:, :w.smer_word]
    exp = orc.fastk(k, bases, boff, cutoff=case["cutoff"])
    assert ni == exp.ninst
    P = orc.params(k)
    assert (P.smer_word, P.kmer_word) == (w.smer_word, w.kmer_word)
    lens = recs[:, w.smer_bytes].astype(int) + 1
    assert lens.sum() == ni and lens.max() <= w.max_super
    ss = orc.msd_sort(np.ascontiguousarray(recs), P.smer_word)
    kl, ovf, nd = orc.kmer_list(P, ss)
    ks = orc.msd_sort(kl, P.kmer_bytes)
    res = orc.count_sorted(P, ks, case["cutoff"])
    util.check_against_golden(case, res.hist, res.max_inst + ovf, res.table)


@pytest.mark.parametrize("name", ["edge_k40_t1_T4", "synth_tiny_k40_t1_T2", "edge_k51_t1_T4",
                                  "edge_k21_t2_T3"])
def test_expand_bit_exact(name):
    case, bases, boff = util.load_case(name)
    k = case["k"]
    P = orc.params(k)
    smers, _ = orc.distribute(P, bases, boff)
    ss = orc.msd_sort(smers, P.smer_word)
    exp, eovf, end = orc.kmer_list(P, ss)
    with fastk_amd.Context(kmer=k) as ctx:
        w = ctx.w
        dev = _pad(ss, w.smer_stride)
        a = ctx.alloc(max(dev.nbytes, 16)).upload(dev)
        nw, nd, _ = ctx.expand(a.ptr, len(ss))
        assert (nw, nd) == (len(exp), end)
        o = ctx.alloc(max(nw, 1) * w.kmer_stride)
        nw2, nd2, ovf = ctx.expand(a.ptr, len(ss), o.ptr, nw)
        got = o.download(nw * w.kmer_stride).reshape(nw, w.kmer_stride)
    assert ovf == eovf
    assert np.array_equal(_unpad_kmers(got, w.kmer_bytes), exp)


@pytest.mark.parametrize("name", ["edge_k40_t1_T4", "edge_k40_t4_T1", "edge_k51_t1_T4",
                                  "synth_illumina_k40_t1_T4"])
def test_count_bit_exact(name):
    case, bases, boff = util.load_case(name)
    k, cutoff = case["k"], case["cutoff"]
    P = orc.params(k)
    smers, _ = orc.distribute(P, bases, boff)
    kl, ovf, _ = orc.kmer_list(P, orc.msd_sort(smers, P.smer_word))
    ks = orc.msd_sort(kl, P.kmer_bytes)
    with fastk_amd.Context(kmer=k, table_cutoff=cutoff) as ctx:
        w = ctx.w
        dev = _pad_kmers(ks, w.kmer_bytes, w.kmer_stride)
        a = ctx.alloc(dev.nbytes).upload(dev)
        t = ctx.alloc(dev.nbytes)
        hist, mi, nd, nt = ctx.count(a.ptr, len(ks), cutoff, t.ptr, len(ks))
        tab = _unpad_kmers(t.download(nt * w.kmer_stride).reshape(nt, w.kmer_stride), w.kmer_bytes)
    util.check_against_golden(case, hist, mi + ovf, tab)


@pytest.mark.parametrize("small_table", [False, True])
@pytest.mark.parametrize("name", util.golden_names())
def test_count_unsorted_bit_exact(name, small_table):
    """fk_count_unsorted_kmers (hash grouping + LDS aggregation + table sort) on the weighted k-mer
    list in RANDOM order reproduces the reference's histogram and table.  small_table shrinks the
    LDS table to eight k-mers so that bins must be split into rounds."""
    _count_unsorted_case(name, small_table, 0)


@pytest.mark.parametrize("engine", [1, 2])
@pytest.mark.parametrize("small_table", [False, True])
@pytest.mark.parametrize("name", ["edge_k40_t4_T1", "edge_k51_t1_T4", "synth_illumina_k40_t1_T4"])
def test_count_unsorted_other_aggregation_routes(name, small_table, engine):
    """The same through k_ag_count (engine 1: the counting sort in LDS of round 3, kept for comparison) and
    through k_ag_count2 with wave lists of two entries (engine 2), so that the elections of the whole
    workgroup -- the route taken when a wave's list is full -- decide nearly every fill."""
    _count_unsorted_case(name, small_table, engine)


def _count_unsorted_case(name, small_table, engine):
    case, bases, boff = util.load_case(name)
    k, cutoff = case["k"], case["cutoff"]
    P = orc.params(k)
    smers, _ = orc.distribute(P, bases, boff)
    kl, ovf, _ = orc.kmer_list(P, orc.msd_sort(smers, P.smer_word))
    kl = kl[np.random.default_rng(11).permutation(len(kl))]
    with fastk_amd.Context(kmer=k, table_cutoff=cutoff) as ctx:
        w = ctx.w
        dev = _pad_kmers(kl, w.kmer_bytes, w.kmer_stride)
        a = ctx.alloc(max(dev.nbytes, 16)).upload(dev)
        t = ctx.alloc(max(dev.nbytes, 16))
        distinct = len(np.unique(kl[:, :P.kmer_bytes], axis=0)) if len(kl) else 0
        limit = 8 if small_table else 0
        ctx.debug_set("aggr_limit", limit)
        ctx.debug_set("aggr_engine", engine)
        hist, mi, nd, nt, tp = ctx.count_unsorted(a.ptr, t.ptr, len(kl), cutoff)
        if small_table and distinct > 65536 * 6:
            assert ctx.debug_get("aggr_extra_rounds") > 0
        assert nd == distinct
        tab = _unpad_kmers(a.download(nt * w.kmer_stride, ptr=tp).reshape(nt, w.kmer_stride),
                           w.kmer_bytes)
    assert nd == int(hist.sum())
    util.check_against_golden(case, hist, mi + ovf, tab)


@pytest.mark.parametrize("cutoff", [3, 5, 9, 20])
def test_pipeline_with_cutoffs_around_and_above_four(cutoff):
    """k_ag_count2 counts the k-mers seen once, twice and three times by ballots, sweeps the table candidates and
    has a third route for counts between four and the cutoff: every combination, the whole path against the oracle
    on a read set with coverage (counts from 1 to 26)."""
    case, bases, boff = util.load_case("synth_illumina_k40_t1_T4")
    k = case["k"]
    exp = orc.fastk(k, bases, boff, cutoff=cutoff)
    with fastk_amd.Context(kmer=k, table_cutoff=cutoff, nbuckets=3) as ctx:
        ctx.push_block(bases, np.asarray(boff - boff[0], dtype=np.int32))
        res = ctx.finish()
        assert res.ninst == exp.ninst
        assert np.array_equal(res.hist, exp.hist)
        assert res.max_inst == exp.max_inst
        assert res.ntable == exp.ntable and exp.ntable > 0
        assert np.array_equal(res.table, exp.table)


def test_count_unsorted_exact_max_inst_for_huge_counts(ctx40):
    """One k-mer with 100,000 records of weight 0x7fff (3.3e9 instances, beyond 32 bits) next to
    ordinary ones: the count saturates at 0x7fff and max_inst is exact (MSDsort.c:498-506)."""
    rng = np.random.default_rng(4)
    n_hot, n_cold = 100000, 5000
    recs = np.zeros((n_hot + n_cold, 12), dtype=np.uint8)
    recs[:n_hot, :10] = 0x5a
    recs[:n_hot, 10:] = np.frombuffer(np.uint16(0x7fff).tobytes(), dtype=np.uint8)
    recs[n_hot:, :10] = rng.integers(0, 256, size=(n_cold, 10))
    recs[n_hot:, 10] = 2
    recs = recs[rng.permutation(len(recs))]
    a = ctx40.alloc(recs.nbytes).upload(recs)
    t = ctx40.alloc(recs.nbytes)
    hist, mi, nd, nt, tp = ctx40.count_unsorted(a.ptr, t.ptr, len(recs), 1)
    assert nd == n_cold + 1 and nt == nd
    assert hist[0x7fff] == 1 and hist[2] == n_cold
    assert mi == n_hot * 0x7fff
    a.free(); t.free()


# ------------------------------------------------------------------------------ whole path

@pytest.mark.parametrize("name", util.golden_names())
def test_pipeline_matches_reference_golden(name, tmp_path):
    case, bases, boff = util.load_case(name)
    k = case["k"]
    with fastk_amd.Context(kmer=k, table_cutoff=case["cutoff"], nthreads=case["T"]) as ctx:
        # feed in several DATA_BLOCK-sized pieces, like io.c does
        nreads = len(boff) - 1
        step = max(1, nreads // 7)
        for s in range(0, nreads, step):
            e = min(nreads, s + step)
            ctx.push_block(bases[boff[s]:boff[e]], (boff[s:e + 1] - boff[s]).astype(np.int32))
        res = ctx.finish()
        util.check_against_golden(case, res.hist, res.max_inst, res.table)
        ctx.write_hist(res, str(tmp_path / "x.hist"))
        ctx.write_ktab(res, str(tmp_path), "x")
    import hashlib
    exp = case["expected"]
    assert hashlib.sha256(open(tmp_path / "x.hist", "rb").read()).hexdigest() == exp["hist_sha256"]
    t = orc.read_ktab(str(tmp_path / "x"))
    assert t["stream_sha256"] == exp["ktab"]["stream_sha256"]
    assert (t["kmer"], t["nparts"], t["minval"], t["ibytes"], t["nels"]) == \
        (k, case["T"], case["cutoff"], exp["ktab"]["ibytes"], exp["ktab"]["nels"])


@pytest.mark.parametrize("name", ["synth_illumina_k40_t1_T4", "edge_k51_t1_T4", "synth_hifi_k40_t4_T8", "edge_k21_t2_T3"])
def test_ktab_written_from_the_device_table(name, tmp_path):
    """fk_finish_device + fk_write_ktab_device: the table never comes to host memory as a whole, every part writer
    fetches its first-byte range piece by piece.  The files are fk_write_ktab's, byte for byte, and the stream is the
    reference's; the device buffers but the table may be released meanwhile (fk_release_device(keep_table))."""
    case, bases, boff = util.load_case(name)
    k, T = case["k"], case["T"]
    a, b = tmp_path / "a", tmp_path / "b"
    a.mkdir(); b.mkdir()
    with fastk_amd.Context(kmer=k, table_cutoff=case["cutoff"], nthreads=T) as ctx:
        ctx.push_block(bases, boff.astype(np.int32))
        res = ctx.finish()
        ctx.write_ktab(res, str(a), "x")
    with fastk_amd.Context(kmer=k, table_cutoff=case["cutoff"], nthreads=T) as ctx:
        ctx.push_block(bases, boff.astype(np.int32))
        res = ctx.finish_device()
        assert res.ntable == case["expected"]["ktab"]["nels"] and len(res.table) == 0
        ctx._ck(ctx.L.fk_release_device(ctx.h, 1))
        ctx.write_ktab_device(res, str(b), "x")
        # more parts than the context was made for (its pinned staging serves T writers): the writers take turns at
        # the staging that exists -- nothing is allocated beside the release -- and the files are the same
        c, d = tmp_path / "c", tmp_path / "d"
        c.mkdir(); d.mkdir()
        ctx.write_ktab_device(res, str(d), "x", nthreads=3 * T + 1)
    with fastk_amd.Context(kmer=k, table_cutoff=case["cutoff"], nthreads=T) as ctx:
        ctx.push_block(bases, boff.astype(np.int32))
        res = ctx.finish()
        ctx.write_ktab(res, str(c), "x", nthreads=3 * T + 1)
    import os
    files = sorted(os.listdir(a))
    assert files == sorted(os.listdir(b)) and len(files) == 1 + T
    for f in files:
        assert util.sha_file(a / f) == util.sha_file(b / f), f
    files = sorted(os.listdir(c))
    assert files == sorted(os.listdir(d)) and len(files) == 1 + 3 * T + 1
    for f in files:
        assert util.sha_file(c / f) == util.sha_file(d / f), f
    assert orc.read_ktab(str(b / "x"))["stream_sha256"] == case["expected"]["ktab"]["stream_sha256"]


def _pack_reads(bases, boff):
    """the 2-bit form fk_push_packed takes, from a DATA_BLOCK: (codes, nbases, rlen, inv)"""
    code = np.full(256, 255, dtype=np.uint8)
    for i, c in enumerate(b"acgt"):
        code[c] = i
        code[c - 32] = i
    n = len(boff) - 1
    rlen = (np.diff(boff) - 1).astype(np.int32)
    keep = np.ones(len(bases), dtype=bool)
    keep[boff[1:] - 1] = False                                   # the terminators
    flat = code[np.asarray(bases)[keep]]
    bad = np.nonzero(flat == 255)[0]
    inv = np.zeros((0, 2), dtype=np.int64)
    if len(bad):
        cut = np.nonzero(np.diff(bad) != 1)[0]
        first = bad[np.concatenate([[0], cut + 1])]
        last = bad[np.concatenate([cut, [len(bad) - 1]])]
        inv = np.stack([first, last - first + 1], axis=1).astype(np.int64)
    flat = np.where(flat == 255, 3, flat).astype(np.uint8)       # (what stands under an invalid base does not matter)
    nb = len(flat)
    pad = np.zeros((nb + 3) // 4 * 4, dtype=np.uint8)
    pad[:nb] = flat
    q = pad.reshape(-1, 4)
    codes = (q[:, 0] << 6) | (q[:, 1] << 4) | (q[:, 2] << 2) | q[:, 3]
    return codes.astype(np.uint8), nb, rlen, inv


@pytest.mark.parametrize("name", util.golden_names())
def test_packed_push_matches_reference_golden(name):
    """fk_push_packed: the reads in two bits per base (+ read lengths + the stretches without acgt), in several
    pieces; resident and chunked with a budget.  Same histogram and table as the reference."""
    case, bases, boff = util.load_case(name)
    k = case["k"]
    nreads = len(boff) - 1
    for kw in (dict(), dict(nbuckets=3, hbm_budget=64 << 20)):
        with fastk_amd.Context(kmer=k, table_cutoff=case["cutoff"], nthreads=case["T"], **kw) as ctx:
            step = max(1, nreads // 5)
            for s0 in range(0, nreads, step):
                e = min(nreads, s0 + step)
                codes, nb, rlen, inv = _pack_reads(bases[boff[s0]:boff[e]], boff[s0:e + 1] - boff[s0])
                ctx.push_packed(codes, nb, rlen, inv)
            res = ctx.finish()
            util.check_against_golden(case, res.hist, res.max_inst, res.table)


def _packed_on_device(ctx, bases, boff):
    """the whole DATA_BLOCK as the caller-owned resident form of fk_count_device_packed: (buffers, args)"""
    codes, nb, rlen, inv = _pack_reads(bases, boff)
    roff = np.concatenate([[0], np.cumsum(rlen.astype(np.int64))]).astype(np.int64)
    cpad = np.zeros((len(codes) + 3) // 4 * 4 + 4, dtype=np.uint8)
    cpad[:len(codes)] = codes
    d_codes = ctx.alloc(len(cpad)).upload(cpad)
    d_roff = ctx.alloc(roff.nbytes).upload(roff)
    d_inv = ctx.alloc(max(inv.nbytes, 16)).upload(inv) if len(inv) else None
    return (d_codes, d_roff, d_inv), (d_codes.ptr, nb, d_roff.ptr, len(rlen), d_inv.ptr if d_inv else None, len(inv))


@pytest.mark.parametrize("name", util.golden_names())
def test_count_device_packed_matches_reference_golden(name):
    """fk_count_device_packed: caller-owned reads resident in two bits per base, split by the PACKED tile loader
    (no ASCII anywhere) -- one bucket; 5 buckets in one pass; 7 buckets in 3 split passes with entry replay and
    without.  Same histogram and table as the reference."""
    case, bases, boff = util.load_case(name)
    # ("planes": the splitter also writes hash digit 1 of every record and the first grouping pass of the super-mers
    #  carries it along instead of hashing the record again -- round 5's experiment, off by default (it did not pay))
    import os
    for kw, dbg in ((dict(), {}), (dict(nbuckets=5), {}), (dict(nbuckets=7, split_passes=3), {}),
                    (dict(nbuckets=7, split_passes=3), {"split_replay": 0}), (dict(nbuckets=5), {"planes": 1})):
        if dbg.pop("planes", 0):
            os.environ["FASTK_AMD_TWO_DIGIT_PLANES"] = "1"
        else:
            os.environ.pop("FASTK_AMD_TWO_DIGIT_PLANES", None)
        with fastk_amd.Context(kmer=case["k"], table_cutoff=case["cutoff"], nthreads=case["T"], **kw) as ctx:
            for key, val in dbg.items():
                ctx.debug_set(key, val)
            bufs, args = _packed_on_device(ctx, bases, boff)
            res = ctx.count_device_packed(*args, fetch_table=True)
            util.check_against_golden(case, res.hist, res.max_inst, res.table)
            for b in bufs:
                if b is not None:
                    b.free()
    os.environ.pop("FASTK_AMD_TWO_DIGIT_PLANES", None)


def test_packed_and_ascii_splitters_agree_on_ragged_reads():
    """Reads of every length around k and around the 16-position words of the tile loader, N runs at read starts, read
    ends and across tile edges, empty reads: the PACKED splitter counts exactly the k-mer instances the ASCII splitter
    counts and the two pipelines give the same histogram and table (which the oracle confirms)."""
    rng = np.random.default_rng(4242)
    k = 40
    reads = []
    for L in list(range(0, 130)) + [4095, 4096, 4097, 8191, 8233, 20000]:
        r = rng.integers(0, 4, size=L)
        s = np.frombuffer(b"acgt", dtype=np.uint8)[r].copy()
        if L > 60 and L % 3 == 0:
            s[0:L % 7 + 1] = ord("N")
        if L > 60 and L % 5 == 0:
            s[L - (L % 11) - 1:] = ord("n")
        if L > 5000:
            s[4090:4101] = ord("N")
            s[4500] = ord("R")
        reads.append(s.tobytes())
    for lo, hi in ((3000, 15000), (100, 4096 * 3 + 7), (4096, 8192)):          # N runs longer than a tile
        s = np.frombuffer(b"acgt", dtype=np.uint8)[rng.integers(0, 4, size=30000)].copy()
        s[lo:hi] = ord("N")
        reads.append(s.tobytes())
    reads += reads[50:90]                                     # repeats: counts above 1
    bases, boff = orc.block_from_reads(reads)
    exp = orc.fastk(k, bases, boff, cutoff=1)
    with fastk_amd.Context(kmer=k, table_cutoff=1, nthreads=4) as ctx:
        ctx.push_block(bases, boff.astype(np.int32))
        a = ctx.finish()
    with fastk_amd.Context(kmer=k, table_cutoff=1, nthreads=4) as ctx:
        bufs, args = _packed_on_device(ctx, bases, boff)
        b = ctx.count_device_packed(*args, fetch_table=True)
    assert a.ninst == b.ninst == exp.ninst
    assert np.array_equal(a.hist, b.hist) and np.array_equal(b.hist[1:], exp.hist[1:])
    assert np.array_equal(a.table, b.table) and np.array_equal(b.table, exp.table)


def test_packed_reads_beyond_2_31_positions():
    """15 M reads of 150 bp = 2.25 G positions, packed on the device, with ONE invalid stretch: the last two positions.
    Every tile's bracket of the stretch list then holds a stretch that lies up to 2.25 G positions behind it (round 4's
    first tile loader narrowed that distance to 32 bits: at BASELINE configs[1] -- 5 G positions, a two-position pad
    behind the last pushed block -- it marked bases of 43 % of the tiles invalid).  Instances, conservation, and the same
    histogram as the ASCII splitter on the same reads (whose last read ends in two N's)."""
    import ctypes as C
    L, k, nreads, glen = 150, 40, 15_000_000, 50_000_000
    with fastk_amd.Context(kmer=k, table_cutoff=2) as ctx:
        buf, nbytes = ctx.synth_reads(77, glen, L, 1000, 0, nreads)
        codes = ctx.alloc(nreads * L // 4 + 64)
        ctx._ck(ctx.L.fk_pack_fixed_reads(ctx.h, buf.ptr, nreads, L, codes.ptr))
        roff = ctx.alloc((nreads + 1) * 8).upload(np.arange(nreads + 1, dtype=np.int64) * L)
        inv = ctx.alloc(16).upload(np.array([nreads * L - 2, 2], dtype=np.int64))
        b = ctx.count_device_packed(codes.ptr, nreads * L, roff.ptr, nreads, inv.ptr, 1, fetch_table=False)
        # the same reads as ASCII, the last two bases of the last read made N
        ctx._ck(ctx.L.fk_copy_to_device(ctx.h, buf.ptr + nbytes - 3, np.frombuffer(b"NN", dtype=np.uint8).ctypes.data, 2))
        a = ctx.count_device_reads(buf.ptr, nbytes, fetch_table=False)
    expect = nreads * (L - k + 1) - 2
    assert a.ninst == b.ninst == expect, (a.ninst, b.ninst, expect)
    hb = b.hist.astype(np.int64)
    assert int((hb[1:0x7fff] * np.arange(1, 0x7fff)).sum()) + int(b.max_inst) == expect
    assert np.array_equal(a.hist, b.hist) and a.ntable == b.ntable


def test_packed_pushes_of_changing_shape():
    """a block of one long read followed by a block of many short reads (more read offsets than the first
    block's staging held), then a block with many N stretches -- and the forms of a run do not mix."""
    rng = np.random.default_rng(99)
    k = 40
    acgt = np.frombuffer(b"acgt", dtype=np.uint8)
    long_read = acgt[rng.integers(0, 4, size=3_000_000)].tobytes()
    short = [acgt[rng.integers(0, 4, size=45)].tobytes() for _ in range(60_000)]
    holes = []
    for _ in range(3000):
        s = acgt[rng.integers(0, 4, size=100)].copy()
        s[rng.integers(0, 100, size=8)] = ord("N")
        holes.append(s.tobytes())
    blocks = [orc.block_from_reads(x) for x in ([long_read], short, holes)]
    allb, allo = orc.block_from_reads([long_read] + short + holes)
    exp = orc.fastk(k, allb, allo, cutoff=2)
    for kw in (dict(), dict(nbuckets=3, hbm_budget=64 << 20)):
        with fastk_amd.Context(kmer=k, table_cutoff=2, nthreads=4, **kw) as ctx:
            for bs, bo in blocks:
                ctx.push_packed(*_pack_reads(bs, bo))
            res = ctx.finish()
            assert res.ninst == exp.ninst and np.array_equal(res.hist[1:], exp.hist[1:])
            assert np.array_equal(res.table, exp.table)
            with pytest.raises(fastk_amd.FastKError):
                ctx.push_block(*[blocks[1][0], blocks[1][1].astype(np.int32)])
            ctx.reset()
            ctx.push_block(blocks[1][0], blocks[1][1].astype(np.int32))     # a new run may take the other form
            with pytest.raises(fastk_amd.FastKError):
                ctx.push_packed(*_pack_reads(*blocks[2]))


@pytest.mark.parametrize("name", ["edge_k40_t1_T4", "synth_illumina_k40_t1_T4"])
def test_packed_push_then_profiles_and_exact_parts(name):
    """The two consumers that walk reads byte by byte restore the ASCII reads from the packed store on the device:
    fk_make_profiles after packed pushes gives the profiles of the ASCII run, and an exact_parts run fed in two bits
    per base gives the first-byte census (hence the part boundaries) of the one fed in ASCII."""
    case, bases, boff = util.load_case(name)
    k, nreads = case["k"], len(boff) - 1
    step = max(1, nreads // 3)
    prof, wf = [], []
    for form in ("ascii", "packed"):
        for exact in (False, True):
            with fastk_amd.Context(kmer=k, table_cutoff=1, nthreads=case["T"], exact_parts=exact) as ctx:
                for s0 in range(0, nreads, step):
                    e = min(nreads, s0 + step)
                    bs, bo = bases[boff[s0]:boff[e]], boff[s0:e + 1] - boff[s0]
                    if form == "ascii":
                        ctx.push_block(bs, bo.astype(np.int32))
                    else:
                        ctx.push_packed(*_pack_reads(bs, bo))
                res = ctx.finish()
                if exact:
                    wf.append(np.array(res.wfirst))
                else:
                    prof.append(ctx.make_profiles())
    assert np.array_equal(wf[0], wf[1])
    assert np.array_equal(prof[0][0], prof[1][0]) and np.array_equal(prof[0][1], prof[1][1])


@pytest.mark.parametrize("name", ["synth_illumina_k40_t1_T4", "edge_k51_t1_T4"])
def test_pipeline_sort_collapse_path_matches_golden(name):
    """fk_debug_set("kmer_stage", 1): the sort / collapse / sort k-mer stage (the fallback of the
    hash-aggregation stage) gives the same result."""
    case, bases, boff = util.load_case(name)
    with fastk_amd.Context(kmer=case["k"], table_cutoff=case["cutoff"], nthreads=case["T"]) as ctx:
        ctx.debug_set("kmer_stage", 1)
        ctx.push_block(bases, boff.astype(np.int32))
        res = ctx.finish()
        util.check_against_golden(case, res.hist, res.max_inst, res.table)


@pytest.mark.parametrize("nb", [2, 5, 16])
@pytest.mark.parametrize("name", ["synth_illumina_k40_t1_T4", "edge_k40_t4_T1", "synth_illumina_k51_t1_T4"])
def test_bucket_streaming_matches_golden(name, nb, tmp_path):
    """nbuckets > 1 through fk_push_block / fk_finish: the minimizer buckets are counted one after
    the other (only one bucket's weighted k-mers are ever in HBM) and the result is the reference's."""
    case, bases, boff = util.load_case(name)
    with fastk_amd.Context(kmer=case["k"], table_cutoff=case["cutoff"], nthreads=case["T"],
                           nbuckets=nb) as ctx:
        ctx.push_block(bases, boff.astype(np.int32))
        res = ctx.finish()
        util.check_against_golden(case, res.hist, res.max_inst, res.table)
        ctx.write_hist(res, str(tmp_path / "x.hist"))
        ctx.write_ktab(res, str(tmp_path), "x")
    import hashlib
    exp = case["expected"]
    assert hashlib.sha256(open(tmp_path / "x.hist", "rb").read()).hexdigest() == exp["hist_sha256"]
    assert orc.read_ktab(str(tmp_path / "x"))["stream_sha256"] == exp["ktab"]["stream_sha256"]


@pytest.mark.parametrize("nb", [1, 4])
@pytest.mark.parametrize("name", ["synth_illumina_k40_t1_T4", "edge_k40_t1_T4", "synth_hifi_k40_t4_T8"])
def test_chunked_ingest_matches_golden(name, nb):
    """HBM-budgeted ingest: the pushed reads are split into super-mers every chunk_bytes and
    forgotten; at fk_finish every bucket is gathered from the chunks and counted."""
    case, bases, boff = util.load_case(name)
    with fastk_amd.Context(kmer=case["k"], table_cutoff=case["cutoff"], nthreads=case["T"],
                           nbuckets=nb) as ctx:
        ctx.debug_set("chunk_bytes", max(4096, len(bases) // 5))
        nreads = len(boff) - 1
        step = max(1, nreads // 23)
        for s0 in range(0, nreads, step):
            e = min(nreads, s0 + step)
            ctx.push_block(bases[boff[s0]:boff[e]], (boff[s0:e + 1] - boff[s0]).astype(np.int32))
        res = ctx.finish()
        exp = orc.fastk(case["k"], bases, boff, cutoff=case["cutoff"])
        assert res.ninst == exp.ninst
        util.check_against_golden(case, res.hist, res.max_inst, res.table)


@pytest.mark.parametrize("nb", [1, 3])
@pytest.mark.parametrize("name", ["synth_illumina_k40_t1_T4", "synth_hifi_k40_t4_T8"])
def test_chunks_spilled_to_host_memory(name, nb):
    """Chunks beyond the HBM allowance live in pinned host memory and come back bucket by bucket:
    same results, and the run really spilled."""
    case, bases, boff = util.load_case(name)
    with fastk_amd.Context(kmer=case["k"], table_cutoff=case["cutoff"], nthreads=case["T"],
                           nbuckets=nb) as ctx:
        ctx.debug_set("chunk_bytes", max(4096, len(bases) // 6))
        ctx.debug_set("spill_limit", max(4096, len(bases) // 30))    # first chunk or two stay, the rest spill
        nreads = len(boff) - 1
        step = max(1, nreads // 17)
        for s0 in range(0, nreads, step):
            e = min(nreads, s0 + step)
            ctx.push_block(bases[boff[s0]:boff[e]], (boff[s0:e + 1] - boff[s0]).astype(np.int32))
        res = ctx.finish()
        assert ctx.debug_get("spilled_bytes") > 0
        util.check_against_golden(case, res.hist, res.max_inst, res.table)
        # the context is reusable afterwards, resident this time
        ctx.debug_set("chunk_bytes", 0)
        ctx.push_block(bases, boff.astype(np.int32))
        res = ctx.finish()
        util.check_against_golden(case, res.hist, res.max_inst, res.table)


@pytest.mark.parametrize("name", ["synth_illumina_k40_t1_T4", "edge_k40_t1_T4", "synth_hifi_k40_t4_T8",
                                  "edge_k51_t1_T4"])
def test_fastq_parsed_on_device_matches_golden(name, tmp_path):
    """fk_push_fastq: raw FASTQ text cut into pieces at arbitrary bytes (inside headers, sequences,
    quality lines that contain the letters ACGT) gives the reads the host parser gives."""
    case, bases, boff = util.load_case(name)
    parts = []
    qpat = b"@ACGT+acgt>#I" * 4000
    for i in range(len(boff) - 1):
        seq = bases[boff[i]:boff[i + 1] - 1].tobytes()
        parts.append(b"@ACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT r%d\n" % i + seq + b"\n+ACGT\n"
                     + qpat[i % 7:i % 7 + len(seq)] + b"\n")
    text = b"".join(parts)
    rng = np.random.default_rng(9)
    with fastk_amd.Context(kmer=case["k"], table_cutoff=case["cutoff"], nthreads=case["T"]) as ctx:
        ph, nr, nb, pos = 0, 0, 0, 0
        while pos < len(text):
            n = int(rng.integers(1, 200000))
            ph, r, b = ctx.push_fastq(text[pos:pos + n], ph)
            nr += r; nb += b
            pos += n
        assert nr == len(boff) - 1
        assert nb == len(bases) - (len(boff) - 1)
        res = ctx.finish()
        util.check_against_golden(case, res.hist, res.max_inst, res.table)


@pytest.mark.parametrize("width", [0, 60, 7])
@pytest.mark.parametrize("name", ["synth_hifi_k40_t4_T8", "edge_k40_t1_T4", "synth_illumina_k40_t1_T4"])
def test_fasta_parsed_on_device_matches_golden(name, width):
    """fk_push_fasta: raw FASTA text (single-line and multi-line records, headers that contain ACGT and
    '>' characters) cut into pieces at arbitrary bytes gives the reference result."""
    case, bases, boff = util.load_case(name)
    parts = []
    for i in range(len(boff) - 1):
        seq = bases[boff[i]:boff[i + 1] - 1].tobytes()
        parts.append(b">r%d ACGT>ACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT\n" % i)
        if width:
            parts.extend(seq[j:j + width] + b"\n" for j in range(0, len(seq), width))
        else:
            parts.append(seq + b"\n")
    text = b"".join(parts)
    rng = np.random.default_rng(4)
    with fastk_amd.Context(kmer=case["k"], table_cutoff=case["cutoff"], nthreads=case["T"]) as ctx:
        st, nr, nb, pos = 2, 0, 0, 0
        while pos < len(text):
            n = int(rng.integers(1, 150000))
            st, r, b = ctx.push_fasta(text[pos:pos + n], st, last=(pos + n >= len(text)))
            nr += r; nb += b
            pos += n
        assert nr == len(boff) - 1
        assert nb == len(bases) - (len(boff) - 1)
        res = ctx.finish()
        util.check_against_golden(case, res.hist, res.max_inst, res.table)


def _hoco(bases, boff):
    """homopolymer-compress every read (io.c:284-294): drop a base equal to the one before it."""
    out, off = [], [0]
    for i in range(len(boff) - 1):
        r = bases[boff[i]:boff[i + 1] - 1]
        keep = np.ones(len(r), dtype=bool)
        keep[1:] = r[1:] != r[:-1]
        out.append(r[keep]); out.append(np.zeros(1, dtype=np.uint8))
        off.append(off[-1] + int(keep.sum()) + 1)
    return np.concatenate(out), np.array(off, dtype=np.int64)


@pytest.mark.parametrize("name", ["synth_illumina_k40_t1_T4", "edge_k40_t1_T4"])
def test_homopolymer_compression_device_host_reference(name, tmp_path):
    """-c: the device FASTQ parser with FK_FASTQ_HOCO, the host parser of FastK_amd -c -H and the
    reference FastK -c agree (histogram bytes)."""
    import hashlib, os, subprocess
    case, bases, boff = util.load_case(name)
    k = case["k"]
    cb, co = _hoco(bases, boff)
    exp = orc.fastk(k, cb, co, cutoff=case["cutoff"])
    path = str(tmp_path / "r.fastq")
    orc.write_fastq(path, bases, boff)
    text = open(path, "rb").read()
    rng = np.random.default_rng(2)
    with fastk_amd.Context(kmer=k, table_cutoff=case["cutoff"], nthreads=case["T"]) as ctx:
        ph, pos, nb = 0, 0, 0
        while pos < len(text):
            n = int(rng.integers(1, 100000))
            ph, r, b = ctx.push_fastq(text[pos:pos + n], ph, hoco=True)
            nb += b
            pos += n
        assert nb == len(cb) - (len(co) - 1)
        res = ctx.finish()
        assert res.ninst == exp.ninst
        assert np.array_equal(res.hist, exp.hist) and res.max_inst == exp.max_inst
        assert np.array_equal(res.table, exp.table)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = util.driver_exe("FastK_amd")
    digests = []
    for extra in ([], ["-H"]):
        subprocess.run([exe, "-k%d" % k, "-t%d" % case["cutoff"], "-c", "-N" + str(tmp_path / "o")] + extra + [path],
                       check=True, cwd=str(tmp_path))
        digests.append(hashlib.sha256(open(tmp_path / "o.hist", "rb").read()).hexdigest())
    assert digests[0] == digests[1]
    ref = os.path.join(orc.REF_DIR, "FastK")
    if os.path.exists(ref):
        subprocess.run([ref, "-k%d" % k, "-t%d" % case["cutoff"], "-c", "-T2", "-P" + str(tmp_path),
                        "-N" + str(tmp_path / "ref"), path], check=True, cwd=str(tmp_path),
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        assert hashlib.sha256(open(tmp_path / "ref.hist", "rb").read()).hexdigest() == digests[0]


@pytest.mark.parametrize("name", ["synth_illumina_k40_t1_T4", "edge_k40_t1_T4", "synth_illumina_k51_t1_T4"])
def test_table_merge_matches_reference_fastmerge(name, tmp_path):
    """Fastmerge_amd -ht (fk_merge_tables: the aggregation kernel sums equal k-mers of several tables,
    saturation bookkeeping of Fastmerge.c:313-329): three tables counted from thirds of the reads merge
    into exactly the table and histogram of the whole data set."""
    import hashlib, os, subprocess
    case, bases, boff = util.load_case(name)
    k = case["k"]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = util.driver_exe("FastK_amd")
    mrg = util.driver_exe("Fastmerge_amd")
    nreads = len(boff) - 1
    cuts = [0, nreads // 3, 2 * nreads // 3, nreads]
    srcs = []
    for i in range(3):
        lo, hi = cuts[i], cuts[i + 1]
        path = str(tmp_path / ("p%d.fastq" % i))
        orc.write_fastq(path, bases[boff[lo]:boff[hi]], boff[lo:hi + 1] - boff[lo])
        subprocess.run([exe, "-k%d" % k, "-t1", "-T3", path], check=True, cwd=str(tmp_path))
        srcs.append(str(tmp_path / ("p%d" % i)))
    subprocess.run([mrg, "-ht", "-T2", str(tmp_path / "ours")] + srcs, check=True, cwd=str(tmp_path))
    ours = orc.read_ktab(str(tmp_path / "ours"))
    ours_hist = open(tmp_path / "ours.hist", "rb").read()
    # -#3: three hidden parts per thread, same table
    subprocess.run([mrg, "-t", "-T2", "-#3", "-P/tmp", str(tmp_path / "six")] + srcs, check=True, cwd=str(tmp_path))
    six = orc.read_ktab(str(tmp_path / "six"))
    assert six["nparts"] == 6 and six["stream_sha256"] == ours["stream_sha256"]
    # expected: Fastmerge's rules (Fastmerge.c:313-329, 985-1030) applied to the three piece tables
    kb = orc.params(k).kmer_bytes
    pieces = [orc.fastk(k, bases[boff[cuts[i]]:boff[cuts[i + 1]]],
                        boff[cuts[i]:cuts[i + 1] + 1] - boff[cuts[i]], cutoff=1) for i in range(3)]
    allrec = np.concatenate([p_.table for p_ in pieces])
    order = np.lexsort(allrec[:, :kb].T[::-1])
    allrec = allrec[order]
    cnt = allrec[:, kb:kb + 2].copy().view("<u2").ravel().astype(np.int64)
    head = np.ones(len(allrec), dtype=bool)
    head[1:] = np.any(allrec[1:, :kb] != allrec[:-1, :kb], axis=1)
    gid = np.cumsum(head) - 1
    tot = np.bincount(gid, weights=cnt).astype(np.int64)
    low = np.bincount(gid, weights=np.where(cnt < 0x7fff, cnt, 0)).astype(np.int64)
    exp_hist = np.bincount(np.minimum(tot, 0x7fff), minlength=0x8000)[1:]
    exp_high = int(low[tot > 0x7fff].sum()) + sum(int(p_.max_inst) for p_ in pieces)
    exp_tab = allrec[head].copy()
    exp_tab[:, kb:kb + 2] = np.minimum(tot, 0x7fff).astype("<u2").view(np.uint8).reshape(-1, 2)
    h = np.frombuffer(ours_hist[28:], dtype=np.int64)
    assert np.array_equal(h, exp_hist)
    assert int(np.frombuffer(ours_hist[20:28], dtype=np.int64)[0]) == exp_high
    assert ours["nels"] == len(exp_tab)
    assert ours["stream_sha256"] == orc.table_stream_sha256(k, exp_tab, ours["ibytes"])
    # where nothing saturates, that is the table and histogram of the whole data set
    whole = orc.fastk(k, bases, boff, cutoff=1)
    if whole.hist[0x7fff] == 0:
        assert np.array_equal(h, whole.hist[1:]) and np.array_equal(exp_tab, whole.table)
    # the reference tool: same header fields; its merged counts depend on -T (k-mers next to its
    # thread partition points are not merged: for these sources -T1..-T4 give four different
    # histograms, none equal to the whole-data one), so only closeness can be asked of it
    ref = os.path.join(orc.REF_DIR, "Fastmerge")
    if os.path.exists(ref):
        subprocess.run([ref, "-ht", "-T2", str(tmp_path / "ref")] + srcs, check=True, cwd=str(tmp_path),
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        theirs = orc.read_ktab(str(tmp_path / "ref"))
        for f in ("kmer", "minval", "ibytes"):
            assert ours[f] == theirs[f], f
        rh = np.frombuffer(open(tmp_path / "ref.hist", "rb").read()[28:], dtype=np.int64)
        assert int(np.abs(rh - h).sum()) <= 64 and abs(theirs["nels"] - ours["nels"]) <= 32


def test_cli_memory_option(tmp_path):
    """FastK_amd -M<GB> (HBM budget: bucket streaming + chunked ingest) gives the same files."""
    import hashlib, os, subprocess
    case, bases, boff = util.load_case("synth_illumina_k40_t1_T4")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = util.driver_exe("FastK_amd")
    path = str(tmp_path / "reads.fastq")
    orc.write_fastq(path, bases, boff)
    p = subprocess.run([exe, "-k40", "-t1", "-T4", "-M1", "-v", path], cwd=str(tmp_path),
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    assert "minimizer bucket" in p.stderr
    exp = case["expected"]
    assert hashlib.sha256(open(tmp_path / "reads.hist", "rb").read()).hexdigest() == exp["hist_sha256"]
    assert orc.read_ktab(str(tmp_path / "reads"))["stream_sha256"] == exp["ktab"]["stream_sha256"]


@pytest.mark.parametrize("prefix", [1, 2, 3, 5])
@pytest.mark.parametrize("name", ["synth_illumina_k40_t1_T4", "edge_k40_t1_T4", "synth_illumina_k51_t1_T4"])
def test_table_sort_by_prefix_and_tie_repair(name, prefix):
    """fk_debug_set("table_sort", p): the table is sorted on its first p key bytes only and the
    records that tie with a neighbour are repaired (p = 2, 3: nearly everything ties on these small
    tables; p = 1 is the plain full-key sort); the table must come out in the reference's order."""
    case, bases, boff = util.load_case(name)
    with fastk_amd.Context(kmer=case["k"], table_cutoff=case["cutoff"], nthreads=case["T"]) as ctx:
        ctx.debug_set("table_sort", prefix)
        ctx.push_block(bases, boff.astype(np.int32))
        res = ctx.finish()
        if prefix in (2, 3):
            assert ctx.debug_get("table_sort_ties") > 0
        util.check_against_golden(case, res.hist, res.max_inst, res.table)


@pytest.mark.parametrize("k", [12, 16, 19, 20, 21, 28, 31, 32, 33, 44, 47, 48, 52, 56, 60, 63, 64])
def test_whole_path_across_k(k):
    """Record widths change with k (KMER_BYTES 3..16, one to five dwords per k-mer record, window
    W = k-4 below and above the 16-position block of the splitter): the whole path must agree with
    the CPU restatement for every width class, resident and bucket-streamed."""
    rng = np.random.default_rng(k)
    genome = rng.integers(0, 4, size=30000)
    reads = []
    for _ in range(1500):
        s0 = int(rng.integers(0, len(genome) - 200))
        r = genome[s0:s0 + int(rng.integers(k - 2, 200))].copy()
        if rng.random() < 0.5:
            r = (3 - r)[::-1]
        if rng.random() < 0.2 and len(r) > 5:
            r[int(rng.integers(0, len(r)))] = 4                     # an N
        reads.append("".join("acgtn"[x] for x in r))
    reads += ["a" * 150] * 300 + ["acgt" * 40] * 50                  # repeats, low complexity
    bases, boff = orc.block_from_reads(reads)
    exp = orc.fastk(k, bases, boff, cutoff=2)
    for nb in (1, 3):
        with fastk_amd.Context(kmer=k, table_cutoff=2, nbuckets=nb) as ctx:
            ctx.push_block(bases, boff.astype(np.int32))
            res = ctx.finish()
            assert res.ninst == exp.ninst
            assert np.array_equal(res.hist, exp.hist) and res.max_inst == exp.max_inst
            assert res.ntable == exp.ntable and np.array_equal(res.table, exp.table)


@pytest.mark.parametrize("name", ["synth_illumina_k40_t1_T4", "edge_k40_t1_T4", "synth_illumina_k51_t1_T4"])
def test_pipeline_four_pass_supermer_path_matches_golden(name):
    """fk_debug_set("smer_stage", 1): super-mers grouped by four hashed passes and run detection in
    the expansion (the path for very wide records and the fallback of the LDS de-duplication)."""
    case, bases, boff = util.load_case(name)
    with fastk_amd.Context(kmer=case["k"], table_cutoff=case["cutoff"], nthreads=case["T"]) as ctx:
        ctx.debug_set("smer_stage", 1)
        ctx.push_block(bases, boff.astype(np.int32))
        res = ctx.finish()
        util.check_against_golden(case, res.hist, res.max_inst, res.table)


@pytest.mark.parametrize("nb", [1, 3])
def test_histogram_only_run(nb):
    """No -t: no table is requested, the k-mer stage ends with the aggregation (no table sort)."""
    case, bases, boff = util.load_case("synth_illumina_k40_t1_T4")
    exp = orc.fastk(40, bases, boff, cutoff=1)
    with fastk_amd.Context(kmer=40, table_cutoff=0, nbuckets=nb) as ctx:
        ctx.push_block(bases, boff.astype(np.int32))
        res = ctx.finish()
        assert res.ntable == 0 and len(res.table) == 0
        assert np.array_equal(res.hist, exp.hist) and res.max_inst == exp.max_inst
        assert res.ndistinct == exp.ndistinct and res.ninst == exp.ninst


def test_empty_and_degenerate_inputs():
    with fastk_amd.Context(kmer=40, table_cutoff=1) as ctx:
        res = ctx.finish()
        assert res.ninst == 0 and res.ntable == 0 and res.hist.sum() == 0
    for reads in (["acgt"], ["a" * 39], ["n" * 100], ["acgtacgtacgtacgtacgtacgtacgtacgtacgtacgt"]):
        bases, boff = orc.block_from_reads(reads)
        exp = orc.fastk(40, bases, boff, cutoff=1)
        with fastk_amd.Context(kmer=40, table_cutoff=1) as ctx:
            ctx.push_block(bases, boff.astype(np.int32))
            res = ctx.finish()
        assert res.ninst == exp.ninst and res.ntable == exp.ntable
        assert np.array_equal(res.hist[1:], exp.hist[1:])
        assert np.array_equal(res.table, exp.table)


def test_copy_rate_helper_copies(ctx40):
    """fk_copy_rate (the copy-kernel ceiling bench.py reports) moves the bytes and returns a rate."""
    n = 64 << 20
    a, b = ctx40.alloc(n), ctx40.alloc(n)
    src = np.arange(n, dtype=np.uint8)
    a.upload(src)
    rate = ctx40.copy_rate(b.ptr, a.ptr, n, reps=2)
    assert rate > 100.0
    assert np.array_equal(b.download(n), src)
    a.free(); b.free()


def test_device_synth_matches_host_generator(ctx40):
    buf, n = ctx40.synth_reads(99, 50000, 150, 3000, 17, 500)
    got = buf.download(n)
    exp, _ = orc.synth_block(99, 50000, 150, 3000, 17, 500)
    assert np.array_equal(got, exp)
    buf.free()


def test_bc_prefix_matches_oracle():
    case, bases, boff = util.load_case("synth_tiny_k40_t1_T2")
    exp = orc.fastk(40, bases, boff, cutoff=1, bc_prefix=12)
    with fastk_amd.Context(kmer=40, table_cutoff=1, bc_prefix=12) as ctx:
        ctx.push_block(bases, boff.astype(np.int32))
        res = ctx.finish()
    assert res.ninst == exp.ninst
    assert np.array_equal(res.hist[1:], exp.hist[1:]) and np.array_equal(res.table, exp.table)


# ------------------------------------------------------------------------------ C host driver

@pytest.mark.parametrize("name,fmt,gz", [("edge_k40_t1_T4", "fasta", False),
                                         ("synth_illumina_k51_t1_T4", "fastq", True),
                                         ("synth_hifi_k40_t4_T8", "fasta", False)])
def test_cli_outputs_match_reference(name, fmt, gz, tmp_path):
    """fastk_amd/bin/FastK_amd (C host over the C-ABI) with FastK's flags: .hist bytes and the
    .ktab canonical stream equal the reference's; the reference-built Tabex/Histex accept them."""
    import gzip, hashlib, os, subprocess
    case, bases, boff = util.load_case(name)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = util.driver_exe("FastK_amd")
    assert os.path.exists(exe), "build fastk_amd/csrc first"
    path = str(tmp_path / ("reads." + fmt))
    (orc.write_fasta if fmt == "fasta" else orc.write_fastq)(path, bases, boff)
    if fmt == "fasta" and name.startswith("synth_hifi"):
        orc.write_fasta(path, bases, boff, width=80)          # multi-line FASTA
    if gz:
        with open(path, "rb") as f, gzip.open(path + ".gz", "wb") as g:
            g.write(f.read())
        os.remove(path)
        path += ".gz"
    subprocess.run([exe, "-k%d" % case["k"], "-t%d" % case["cutoff"], "-T%d" % case["T"], "-v", path],
                   check=True, cwd=str(tmp_path))
    exp = case["expected"]
    hist = open(tmp_path / "reads.hist", "rb").read()
    assert hashlib.sha256(hist).hexdigest() == exp["hist_sha256"]
    t = orc.read_ktab(str(tmp_path / "reads"))
    assert t["stream_sha256"] == exp["ktab"]["stream_sha256"]
    assert (t["nparts"], t["minval"], t["ibytes"], t["nels"]) == \
        (case["T"], case["cutoff"], exp["ktab"]["ibytes"], exp["ktab"]["nels"])
    tabex = os.path.join(orc.REF_DIR, "Tabex")
    if os.path.exists(tabex):
        out = subprocess.run([tabex, "-C", str(tmp_path / "reads")], capture_output=True, text=True)
        assert "Table is OK" in out.stdout + out.stderr
    # -x: every file, hidden parts included, is byte-identical to the reference's
    subprocess.run([exe, "-k%d" % case["k"], "-t%d" % case["cutoff"], "-T%d" % case["T"], "-x", path],
                   check=True, cwd=str(tmp_path))
    for fname, digest in exp["file_sha256"].items():
        mine = fname.replace("x.", "reads.", 1) if fname.startswith("x.") else fname.replace(".x.", ".reads.", 1)
        assert hashlib.sha256(open(tmp_path / mine, "rb").read()).hexdigest() == digest, fname


@pytest.mark.parametrize("name,fmt,width,piece", [("edge_k40_t1_T4", "fasta", 0, 0), ("edge_k40_t1_T4", "fastq", 0, 700),
                                                  ("synth_illumina_k51_t1_T4", "fastq", 0, 100000),
                                                  ("synth_hifi_k40_t4_T8", "fasta", 80, 300000),
                                                  ("synth_hifi_k40_t4_T8", "fasta", 0, 50000)])
def test_cli_reader_threads_pack_the_text(name, fmt, width, piece, tmp_path):
    """Plain FASTA / FASTQ files reach the GPU two bits per base: the reader threads of FastK_amd cut the file at record
    starts, resolve the lines and pack the bases (scan_text_packed -> fk_push_packed).  Same .hist bytes and .ktab
    stream as the reference, whatever the piece size; FASTK_AMD_DEVICE_TEXT=1 (text parsed on the device) agrees."""
    import hashlib, os, subprocess
    case, bases, boff = util.load_case(name)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = util.driver_exe("FastK_amd")
    path = str(tmp_path / ("reads." + fmt))
    if fmt == "fasta":
        orc.write_fasta(path, bases, boff, **({"width": width} if width else {}))
    else:
        orc.write_fastq(path, bases, boff)
    exp = case["expected"]
    for env_extra in ({"FASTK_AMD_PIECE": str(piece)} if piece else {}, {"FASTK_AMD_DEVICE_TEXT": "1"}):
        out = subprocess.run([exe, "-k%d" % case["k"], "-t%d" % case["cutoff"], "-T%d" % case["T"], "-v", path],
                             check=True, cwd=str(tmp_path), env=dict(os.environ, **env_extra),
                             capture_output=True, text=True)
        hist = open(tmp_path / "reads.hist", "rb").read()
        assert hashlib.sha256(hist).hexdigest() == exp["hist_sha256"], env_extra
        t = orc.read_ktab(str(tmp_path / "reads"))
        assert t["stream_sha256"] == exp["ktab"]["stream_sha256"], env_extra
        assert ("There are %d reads totalling %d bps" % (len(boff) - 1, int(boff[-1]) - (len(boff) - 1))) \
            in out.stdout + out.stderr, out.stderr
        for f in os.listdir(tmp_path):
            if f != "reads." + fmt:
                os.remove(tmp_path / f)


@pytest.mark.parametrize("name", ["synth_hifi_k40_t4_T8", "edge_k40_t1_T4"])
def test_cli_several_input_files(name, tmp_path):
    """FastK_amd a.fasta b.fasta.gz c.fasta (one type per run, io.c; outputs named after the first file, FastK.c:402-405):
    the reads of a golden case dealt over three files -- two plain ones, which the reader threads pack, around a gzipped
    one, whose text is parsed on the device -- give the golden's .hist bytes and table."""
    import gzip, hashlib, os, subprocess
    case, bases, boff = util.load_case(name)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = util.driver_exe("FastK_amd")
    raw = bytes(bases)
    reads = [raw[boff[i]:boff[i + 1] - 1] for i in range(len(boff) - 1)]
    cuts = [0, len(reads) // 3, len(reads) // 3 + len(reads) // 5, len(reads)]
    paths = []
    for f in range(3):
        b, o = orc.block_from_reads(reads[cuts[f]:cuts[f + 1]])
        path = str(tmp_path / ("part%d.fasta" % f))
        orc.write_fasta(path, b, o, width=70 if f == 2 else 0)
        if f == 1:
            with open(path, "rb") as fi, gzip.open(path + ".gz", "wb") as fo:
                fo.write(fi.read())
            os.remove(path)
            path += ".gz"
        paths.append(path)
    out = subprocess.run([exe, "-k%d" % case["k"], "-t%d" % case["cutoff"], "-T%d" % case["T"], "-v"] + paths, check=True,
                         cwd=str(tmp_path), capture_output=True, text=True)
    exp = case["expected"]
    assert hashlib.sha256(open(tmp_path / "part0.hist", "rb").read()).hexdigest() == exp["hist_sha256"]
    t = orc.read_ktab(str(tmp_path / "part0"))
    assert t["stream_sha256"] == exp["ktab"]["stream_sha256"] and t["nels"] == exp["ktab"]["nels"]
    assert ("There are %d reads" % len(reads)) in out.stdout + out.stderr


def test_cli_one_long_record(tmp_path):
    """A FASTA file that is ONE record of 12 Mbp in 60-base lines (a chromosome, not a read): the reader threads cannot
    cut it (a piece ends at a record start), the host scanner (-H) cuts it into blocks with K-1 bases of overlap
    (io.c:557-570), the device parser takes it as text -- same .hist bytes and table, and every 40-mer counted once."""
    import hashlib, os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = util.driver_exe("FastK_amd")
    rng = np.random.default_rng(7)
    L = 12_000_000
    seq = np.frombuffer(b"acgt", dtype=np.uint8)[rng.integers(0, 4, size=L)].copy()
    for at in (1_000_000, 7_777_777):
        seq[at:at + 500] = ord("N")
    rows = seq.reshape(-1, 60)
    text = np.empty((rows.shape[0], 61), dtype=np.uint8)
    text[:, :60] = rows
    text[:, 60] = ord("\n")
    path = str(tmp_path / "chr.fasta")
    with open(path, "wb") as f:
        f.write(b">chr1 one record\n")
        text.tofile(f)
    seen = {}
    for tag, extra, env in (("packed", [], {"FASTK_AMD_PIECE": "1048576"}), ("host scanner", ["-H"], {}),
                            ("device text", [], {"FASTK_AMD_DEVICE_TEXT": "1"})):
        subprocess.run([exe, "-k40", "-t1", "-T4", "-N" + str(tmp_path / "out")] + extra + [path], check=True,
                       cwd=str(tmp_path), env=dict(os.environ, **env), capture_output=True, text=True)
        hist = open(tmp_path / "out.hist", "rb").read()
        t = orc.read_ktab(str(tmp_path / "out"))
        seen[tag] = (hashlib.sha256(hist).hexdigest(), t["stream_sha256"], t["nels"])
        h = orc.read_hist(str(tmp_path / "out.hist"))
        for f in os.listdir(tmp_path):
            if f.startswith("out") or f.startswith(".out"):
                os.remove(tmp_path / f)
    assert seen["packed"] == seen["host scanner"] == seen["device text"], seen
    inst = (1_000_000 - 39) + (7_777_777 - 1_000_500 - 39) + (L - 7_778_277 - 39)
    counts = np.asarray(h["hist"], dtype=np.int64)                 # counts[i] = k-mers that occur low + i times
    assert int((counts * (h["low"] + np.arange(len(counts)))).sum()) == inst


def _odd_text(kind, quirks):
    """FASTA / FASTQ text with what real files hold beside acgt: upper and lower case, N runs, IUPAC codes, lines of
    several widths, a quality line that begins with '@'; with quirks also records without bases and no newline at
    the very end -- where the reference's scanner (io.c:685-738) does what no FASTA reader would: the line after a
    header is sequence whatever it begins with (a record without bases swallows the next header, whose letters then
    meet the bases behind it), and a last read that no newline ends is dropped."""
    import random
    rnd = random.Random(20251002 + len(kind))
    out = []
    def seq(n):
        s = []
        while len(s) < n:
            x = rnd.random()
            if x < 0.01:
                s.extend("N" * rnd.randint(1, 45))
            elif x < 0.013:
                s.append(rnd.choice("RYKMSWnryk"))
            else:
                s.append(rnd.choice("acgtACGT"))
        return "".join(s[:n])
    nrec = 400
    for r in range(nrec):
        n = rnd.choice(((0,) if quirks else ()) + (1, 39, 40, 41, 97, 150, 1500, 9000)) if r % 7 == 0 else rnd.randint(60, 3000)
        if r == nrec - 1:
            n = 2500                                  # the read the reference drops when no newline ends it
        b = seq(n)
        if kind == "fasta":
            out.append(">read%d some text > with marks @ in it\n" % r)
            w = rnd.choice((60, 80, 7, 100000))
            for i in range(0, len(b), w):
                out.append(b[i:i + w] + "\n")
        else:
            out.append("@read%d/1 +\n%s\n+\n%s\n" % (r, b, ("@" if r % 3 == 0 else "I") + "I" * max(len(b) - 1, 0)
                       if len(b) > 0 else ""))
    text = "".join(out)
    return text[:-1] if quirks else text              # quirks: the last line has no newline


@pytest.mark.parametrize("kind,quirks", [("fasta", False), ("fastq", False), ("fasta", True), ("fastq", True)])
def test_cli_text_parsers_agree_on_odd_files(kind, quirks, tmp_path):
    """The ways FastK_amd reads a plain text file -- packed by the reader threads (default), scanned by the host state
    machine (-H), parsed on the device (FASTK_AMD_DEVICE_TEXT=1) -- and the reference FastK itself give the same .hist
    bytes and table on a file with N runs, IUPAC codes, both cases and ragged lines; the packed path also when the file
    is cut into pieces of a few hundred bytes.  With quirks (records without bases, no final newline) the reader threads
    and the host scanner still follow the reference; the device parser reads FASTA as FASTA and is left out."""
    import hashlib, os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = util.driver_exe("FastK_amd")
    path = str(tmp_path / ("odd." + kind))
    with open(path, "w") as f:
        f.write(_odd_text(kind, quirks))
    seen = {}
    runs = [("packed", [], {}), ("packed, small pieces", [], {"FASTK_AMD_PIECE": "900"}),
            ("packed, scalar", [], {"FASTK_AMD_SCALAR": "1", "FASTK_AMD_PIECE": "5000"}),
            ("pread pieces", [], {"FASTK_AMD_MMAP": "0", "FASTK_AMD_PIECE": "3000"}),
            ("host scanner", ["-H"], {})]
    if not quirks:
        runs.append(("device text", [], {"FASTK_AMD_DEVICE_TEXT": "1"}))
    for tag, extra, env in runs:
        out = subprocess.run([exe, "-k40", "-t1", "-T4", "-v", "-N" + str(tmp_path / "out")] + extra + [path], check=True,
                             cwd=str(tmp_path), env=dict(os.environ, **env), capture_output=True, text=True)
        hist = hashlib.sha256(open(tmp_path / "out.hist", "rb").read()).hexdigest()
        t = orc.read_ktab(str(tmp_path / "out"))
        line = [x.strip() for x in (out.stdout + out.stderr).splitlines() if "reads totalling" in x]
        seen[tag] = (hist, t["stream_sha256"], t["nels"], line[0] if line else "")
        for f in os.listdir(tmp_path):
            if f.startswith("out") or f.startswith(".out"):
                os.remove(tmp_path / f)
    first = seen["packed"]
    assert first[2] > 1000
    for tag, v in seen.items():
        assert v == first, (tag, v, first)
    if orc.have_ref():
        orc.run_ref_fastk(path, 40, 1, 4, str(tmp_path))
        assert hashlib.sha256(open(tmp_path / ("odd.hist"), "rb").read()).hexdigest() == first[0]
        t = orc.read_ktab(str(tmp_path / "odd"))
        assert (t["stream_sha256"], t["nels"]) == first[1:3]


def test_group_records_brings_duplicates_together(ctx40):
    """fk_group_records: a permutation of the input in which equal records are adjacent."""
    rng = np.random.default_rng(21)
    n, rsize = 300000, 20
    distinct = rng.integers(0, 256, size=(20000, rsize), dtype=np.uint8)
    recs = distinct[rng.integers(0, len(distinct), size=n)]
    a = ctx40.alloc(recs.nbytes).upload(recs)
    b = ctx40.alloc(recs.nbytes)
    res = ctx40.group(a.ptr, b.ptr, n, rsize)
    got = a.download(recs.nbytes, ptr=res).reshape(n, rsize)
    # same multiset
    assert np.array_equal(orc.msd_sort(got, rsize), orc.msd_sort(recs, rsize))
    # number of runs == number of distinct records (no 40-bit collision expected at this size)
    runs = 1 + int(np.any(got[1:] != got[:-1], axis=1).sum())
    assert runs == len(np.unique(recs, axis=0))
    a.free(); b.free()


# ------------------------------------------------------------------------------ prefix-sorted count

def _shuffle_within_prefix(ks, nbytes, seed):
    """Keep the order of the first nbytes key bytes, shuffle records inside equal-prefix runs."""
    rng = np.random.default_rng(seed)
    pre = np.zeros(len(ks), dtype=np.int64)
    for b in range(nbytes):
        pre = (pre << 8) | ks[:, b].astype(np.int64)
    order = np.lexsort((rng.random(len(ks)), pre))
    return np.ascontiguousarray(ks[order])


@pytest.mark.parametrize("name,nbytes", [("edge_k40_t1_T4", 4), ("synth_illumina_k40_t1_T4", 4),
                                         ("edge_k51_t1_T4", 4), ("synth_illumina_k40_t1_T4", 2),
                                         ("edge_k21_t2_T3", 3)])
def test_count_on_prefix_sorted_input(name, nbytes):
    """k-mers ordered on their first bytes only (what four digit passes give): the count kernel
    must resolve heterogeneous prefix runs in LDS and still reproduce the reference exactly, or
    report FK_ESTATE (then the caller sorts on) -- never a wrong answer."""
    case, bases, boff = util.load_case(name)
    k, cutoff = case["k"], case["cutoff"]
    P = orc.params(k)
    smers, _ = orc.distribute(P, bases, boff)
    kl, ovf, _ = orc.kmer_list(P, orc.msd_sort(smers, P.smer_word))
    ks = _shuffle_within_prefix(orc.msd_sort(kl, P.kmer_bytes), nbytes, 5)
    with fastk_amd.Context(kmer=k, table_cutoff=cutoff) as ctx:
        w = ctx.w
        dev = _pad_kmers(ks, w.kmer_bytes, w.kmer_stride)
        a = ctx.alloc(dev.nbytes).upload(dev)
        t = ctx.alloc(dev.nbytes)
        out = ctx.count(a.ptr, len(ks), cutoff, t.ptr, len(ks), sorted_bytes=nbytes)
        if out is None:
            assert nbytes < 4, "a 4-byte prefix must be resolvable on these inputs"
            return
        hist, mi, nd, nt = out
        tab = _unpad_kmers(t.download(nt * w.kmer_stride).reshape(nt, w.kmer_stride), w.kmer_bytes)
    util.check_against_golden(case, hist, mi + ovf, tab)


def test_count_reports_unresolvable_prefix_runs(ctx40):
    """2,000 distinct k-mers sharing one 4-byte prefix, in random order: too long to fix in LDS."""
    rng = np.random.default_rng(3)
    n = 2000
    recs = np.zeros((n, 12), dtype=np.uint8)
    recs[:, 4:10] = rng.integers(0, 256, size=(n, 6))
    recs[:, 10] = 1
    a = ctx40.alloc(recs.nbytes).upload(recs)
    assert ctx40.count(a.ptr, n, 1, None, 0, sorted_bytes=4) is None
    a.free()


def test_sharded_path_on_one_gpu_matches_plain_path():
    """bench.py --force-shard runs the C shard engine on one rank (planned split into 4 exchange rounds' buckets, the
    local share of the exchange, per-round counting, C2, C3); totals must equal the single-context pipeline's.  (ASCII
    stripes: the packed splitter cuts a few super-mers elsewhere at tile edges, so only k-mer totals would compare.)"""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for extra in ([], ["--force-shard", "--ascii-stripes"]):
        cmd = [sys.executable, os.path.join(root, "bench.py"), "--config", "1", "--genome-mbp", "2", "--steps", "1",
               "--warmup", "0", "--no-cpu-baseline", "--no-device-leg", "--no-e2e"] + extra
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
        p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        line = [x for x in p.stdout.splitlines() if x.startswith("{")][-1]
        outs.append(json.loads(line)["config"])
    for key in ("kmer_instances", "supermers", "distinct_kmers"):
        assert outs[0][key] == outs[1][key], key


# failures of the several-ranks-on-one-GPU rig itself (RCCL refusing the set-up), as opposed to wrong results
_RCCL_RIG_ERRORS = ("ncclInvalidUsage", "ncclSystemError", "ncclUnhandledCudaError", "Duplicate GPU detected",
                    "unhandled system error", "invalid usage")


@pytest.mark.parametrize("name,fmt,ranks", [("synth_illumina_k40_t1_T4", "fastq", 2), ("synth_illumina_k40_t1_T4", "fastq", 4),
                                             ("synth_hifi_k40_t4_T8", "fasta", 2), ("edge_k40_t1_T4", "fasta", 4),
                                             ("synth_illumina_k51_t1_T4", "fastq", 2), ("configs0_k40_t1_T4", "fastq", 4)])
def test_c_driver_sharded_over_rccl_writes_the_one_gpu_files(name, fmt, ranks, tmp_path):
    """FastK_amd -G<n>: the C host starts one process per rank, every rank reads its stripe of the file,
    the super-mers travel by minimizer bucket and the table entries by first byte with RCCL called from
    C (fk_shard_count / fk_shard_write, no Python, no host staging of the payload), and every rank writes
    its own hidden .ktab parts.  On this one-GPU box the ranks share device 0 (FK_RANKS_SHARE_GPU=1: RCCL
    then moves the payload over its socket transport).  Every file must be byte-identical to the one-GPU
    run with the same -T, and .hist / the canonical stream must be the reference's."""
    import os, subprocess
    case, bases, boff = util.load_case(name)
    exp = case["expected"]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = util.driver_exe("FastK_amd")
    path = str(tmp_path / ("reads." + fmt))
    if case["kind"] == "edge":
        orc.write_fasta(path, bases, boff)
    else:
        util.write_fastx(path, bases, boff, fmt == "fastq")
    T = 4
    args = ["-k%d" % case["k"], "-t%d" % case["cutoff"], "-T%d" % T]
    one, many = tmp_path / "one", tmp_path / "many"
    one.mkdir(); many.mkdir()
    subprocess.run([exe] + args + ["-N" + str(one / "x"), path], check=True)
    env = dict(os.environ, FK_RANKS_SHARE_GPU="1")
    p = subprocess.run([exe] + args + ["-v", "-G%d" % ranks, "-N" + str(many / "x"), path], env=env, capture_output=True, text=True,
                       timeout=900)
    if p.returncode != 0 and any(m in p.stdout + p.stderr for m in _RCCL_RIG_ERRORS):
        if os.environ.get("FK_REQUIRE_RANKS") == "1":
            pytest.fail("RCCL would not bring up %d ranks on one GPU: %s" % (ranks, p.stderr[-500:]))
        pytest.skip("RCCL would not bring up several ranks on one GPU here: " + p.stderr[-300:])
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    files = sorted(os.listdir(one))
    assert files == sorted(os.listdir(many)) and len(files) == 2 + T
    for f in files:
        assert util.sha_file(one / f) == util.sha_file(many / f), f
    assert util.sha_file(many / "x.hist") == exp["hist_sha256"]
    t = orc.read_ktab(str(many / "x"))
    assert t["stream_sha256"] == exp["ktab"]["stream_sha256"] and t["nels"] == exp["ktab"]["nels"]


def _prof_stream(d, root):
    """the profile set of <d>/<root>.prof as (per-read byte strings in file order): parts concatenated"""
    kk, enc = orc.read_profiles(str(d), root)
    return kk, enc


@pytest.mark.parametrize("name,fmt,ranks,budget", [("synth_illumina_k51_t1_T4", "fastq", 2, True), ("synth_illumina_k40_t1_T4", "fastq", 4, True),
                                                    ("edge_k40_t1_T4", "fasta", 2, False), ("edge_k51_t1_T4", "fasta", 2, True),
                                                    # -t<n> beside -p (the reference's FastK.c:491-540 in one main()): counted with
                                                    # cutoff 1 for the look-ups, the .ktab keeps what reaches n (fk_shard_set_write_cutoff)
                                                    ("edge_k40_t4_T1", "fasta", 2, False), ("synth_hifi_k40_t4_T8", "fasta", 4, True),
                                                    ("edge_k21_t2_T3", "fasta", 2, True)])
def test_c_driver_sharded_with_profiles_and_budget(name, fmt, ranks, budget, tmp_path):
    """FastK_amd -G<n> -t1 -p [-M1]: BASELINE configs[4]'s options through the C host on several ranks (the one-GPU rig).
    The counting pass of every rank splits its stripe chunk by chunk (the test shrinks the chunks and the HBM share
    of the records so that a fixture is cut into several chunks, most of them spilled to pinned host memory) and the
    chunk store feeds the exchange rounds; the profile pass looks every k-mer up on the rank that owns its minimizer
    bucket (fk_shard_profiles) and every rank writes the .prof parts of its own range of the reads.  .hist, the
    canonical .ktab stream and the DECODED profiles must be the REFERENCE's (golden digests); the profile bytes must be
    those of the one-GPU run."""
    import os, subprocess
    case, bases, boff = util.load_case(name)
    exp = case["expected"]
    assert "prof" in exp
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = util.driver_exe("FastK_amd")
    path = str(tmp_path / ("reads." + fmt))
    if case["kind"] == "edge":
        orc.write_fasta(path, bases, boff)
    elif len(set(np.diff(boff))) == 1:
        util.write_fastx(path, bases, boff, fmt == "fastq")
    else:
        orc.write_fasta(path, bases, boff, width=100)
    T = 4
    args = ["-k%d" % case["k"], "-t%d" % case["cutoff"], "-p", "-T%d" % T]
    one, many = tmp_path / "one", tmp_path / "many"
    one.mkdir(); many.mkdir()
    subprocess.run([exe] + args + ["-N" + str(one / "x"), path], check=True)
    env = dict(os.environ, FK_RANKS_SHARE_GPU="1")
    extra = []
    if budget:
        extra = ["-M1"]
        env.update(FASTK_AMD_CHUNK_BYTES=str(max(len(bases) // 7, 4096)), FASTK_AMD_SPILL_LIMIT=str(max(len(bases) // 6, 4096)))
    p = subprocess.run([exe] + args + extra + ["-v", "-G%d" % ranks, "-N" + str(many / "x"), path], env=env, capture_output=True,
                       text=True, timeout=900)
    if p.returncode != 0 and any(m in p.stdout + p.stderr for m in _RCCL_RIG_ERRORS):
        if os.environ.get("FK_REQUIRE_RANKS") == "1":
            pytest.fail("RCCL would not bring up %d ranks on one GPU: %s" % (ranks, p.stderr[-500:]))
        pytest.skip("RCCL would not bring up several ranks on one GPU here: " + p.stderr[-300:])
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    assert util.sha_file(many / "x.hist") == exp["hist_sha256"]
    t = orc.read_ktab(str(many / "x"))
    assert t["stream_sha256"] == exp["ktab"]["stream_sha256"] and t["nels"] == exp["ktab"]["nels"]
    k1, enc1 = _prof_stream(one, "x")
    kn, encn = _prof_stream(many, "x")
    assert k1 == kn == case["k"] and len(encn) == exp["prof"]["nreads"]
    assert encn == enc1, "the ranks' profiles are not the one-GPU run's"
    assert orc.profiles_digest([orc.profile_decode(e) for e in encn]) == exp["prof"]["decoded_sha256"]


@pytest.mark.parametrize("how", ["gz", "hoco"])
def test_c_driver_sharded_deals_pieces_of_long_reads(how, tmp_path):
    """-G2 with input the HOST parses (gzipped FASTA; -c on plain FASTA) holding a read of 20 Mbp: the driver cuts it
    into 8 MB blocks with a K-1 overlap (rem > 0, io.c:557-570) and deals the pieces to the ranks like any other
    block.  (Round 2 pushed every cut piece on EVERY rank: the k-mers of long reads were counted once per rank and
    the conservation checks could not see it.)  Every file must equal the one-GPU run's, and the histogram must
    count every k-mer instance once."""
    import gzip, os, subprocess
    rng = np.random.default_rng(77)
    acgt = np.frombuffer(b"acgt", dtype=np.uint8)
    reads = [acgt[rng.integers(0, 4, size=n)] for n in (20_000_000, 3000, 9_000_000, 150, 40)]
    if how == "hoco":                            # no homopolymer runs: -c keeps every base, the counts stay checkable
        for r in reads:
            same = np.nonzero(r[1:] == r[:-1])[0] + 1
            while len(same):
                r[same] = acgt[(np.searchsorted(acgt, r[same]) + 1 + (same & 1)) % 4]
                same = np.nonzero(r[1:] == r[:-1])[0] + 1
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = util.driver_exe("FastK_amd")
    path = str(tmp_path / ("long.fa.gz" if how == "gz" else "long.fa"))
    with (gzip.open(path, "wb", compresslevel=1) if how == "gz" else open(path, "wb")) as f:
        for i, r in enumerate(reads):
            f.write(b">r%d\n" % i)
            f.write(r.tobytes())
            f.write(b"\n")
    k, T = 40, 4
    args = ["-k%d" % k, "-t1", "-T%d" % T] + (["-c"] if how == "hoco" else [])
    one, many = tmp_path / "one", tmp_path / "many"
    one.mkdir(); many.mkdir()
    subprocess.run([exe] + args + ["-N" + str(one / "x"), path], check=True)
    p = subprocess.run([exe] + args + ["-G2", "-N" + str(many / "x"), path], env=dict(os.environ, FK_RANKS_SHARE_GPU="1"),
                       capture_output=True, text=True, timeout=900)
    if p.returncode != 0 and any(m in p.stdout + p.stderr for m in _RCCL_RIG_ERRORS) and os.environ.get("FK_REQUIRE_RANKS") != "1":
        pytest.skip("RCCL would not bring up several ranks on one GPU here: " + p.stderr[-300:])
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    files = sorted(os.listdir(one))
    assert files == sorted(os.listdir(many)) and len(files) == 2 + T
    for f in files:
        assert util.sha_file(one / f) == util.sha_file(many / f), f
    h = orc.read_hist(str(many / "x.hist"))
    hist = np.asarray(h["hist"], dtype=np.int64)
    cnt = np.arange(h["low"], h["low"] + len(hist))
    inst = sum(len(r) - k + 1 for r in reads)
    assert int((hist[:-1] * cnt[:-1]).sum()) + int(h["ihigh"]) == inst


@pytest.mark.parametrize("ranks", [2, 4])
def test_ranks_on_one_gpu_match_one_context(ranks):
    """A real exchange between processes: `ranks` RCCL ranks share device 0 (tools/ranks_on_one_gpu.py
    gives each its own NCCL_HOSTID, so RCCL accepts them and moves the payload over its socket
    transport).  count_sharded, count_sharded_rounds, gather_table, write_table_sharded,
    profiles_sharded and profiles_exchanged with the HIP stages on every rank: histogram, totals,
    merged table, the .ktab files the ranks write and every read's profile bytes equal the
    one-context run over all reads."""
    import importlib.util, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("ranks_on_one_gpu", os.path.join(root, "tools", "ranks_on_one_gpu.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rc, text = mod.parent(world=ranks, port=29650 + ranks)
    if rc != 0 and "MISMATCH" not in text and any(m in text for m in _RCCL_RIG_ERRORS):
        if os.environ.get("FK_REQUIRE_RANKS") == "1":
            pytest.fail("RCCL would not bring up %d ranks on one GPU (FK_REQUIRE_RANKS=1): %s" % (ranks, text[-500:]))
        pytest.skip("RCCL would not bring up several ranks on one GPU here: " + text[-300:])
    assert rc == 0, text[-3000:]
    assert text.count("equal to the one-context run") == 4 and "MISMATCH" not in text, text[-3000:]


def test_bench_contract_with_two_ranks_on_one_gpu():
    """The driver's N > 1 command line (torch.distributed.run ... bench.py --gpus 2) end to end, both
    ranks on device 0: one JSON line from rank 0, n_gpus 2, twice the one-rank k-mer instances
    (weak scaling), exchange verified by the warm-up step's checksums."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29663", os.path.join(root, "bench.py"),
           "--gpus", "2", "--config", "1", "--steps", "2", "--warmup", "1", "--genome-mbp", "2"]
    env = dict(os.environ, FK_RANKS_SHARE_GPU="1")
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    if p.returncode != 0 and any(m in p.stdout + p.stderr for m in _RCCL_RIG_ERRORS):
        if os.environ.get("FK_REQUIRE_RANKS") == "1":
            pytest.fail("RCCL would not bring up two ranks on one GPU (FK_REQUIRE_RANKS=1): " + p.stderr[-500:])
        pytest.skip("RCCL would not bring up two ranks on one GPU here: " + p.stderr[-300:])
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    lines = [x for x in p.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["scaling"] == "weak"
    reads = int(50 * 2 * 2e6 / 150) // 2 * 2
    assert out["config"]["kmer_instances"] == reads * 111
    assert out["value"] > 0 and out["roofline"]["achieved"] > 0 and "cpu_baseline" not in out


@pytest.mark.parametrize("ranks", [1, 2])
def test_bench_config3_through_the_c_shard_engine(ranks, tmp_path):
    """bench.py --config 3 (BASELINE configs[3]: the HiFi-shaped set striped over the GPUs, strong scaling) drives
    fk_shard_count_device and fk_shard_gather -- RCCL called from C -- under the driver's torch.distributed.run command
    line, two ranks on the one GPU.  The workload is the golden case hifi50x20M (50x of a 20 Mbp genome in 15 kbp reads,
    reference FastK's files): the line's .hist digest and the canonical stream of the ranks' gathered table ranges,
    taken in rank order, must be the REFERENCE's; the timed step includes the final gather (C3), and the line says how
    many ranks RCCL's communicator held and what every rank sent."""
    import json, os, subprocess, sys
    case = json.load(open(os.path.join(util.GOLDEN, "hifi50x20M_k40_t4_T4.json")))
    sy, exp = case["synth"], case["expected"]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tail = [os.path.join(root, "bench.py"), "--gpus", str(ranks), "--genome-mbp", str(sy["genome_len"] / 1e6), "--seed", str(sy["seed"]),
            "--steps", "2", "--warmup", "1", "--dump-table", str(tmp_path)]
    if ranks == 1:
        tail += ["--config", "3"]                 # (with several ranks configs[3] is the default)
    if ranks == 1:
        cmd, env = [sys.executable] + tail, dict(os.environ)
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks),
               "--master-addr", "127.0.0.1", "--master-port", "29671"] + tail
        env = dict(os.environ, FK_RANKS_SHARE_GPU="1")
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    if p.returncode != 0 and any(m in p.stdout + p.stderr for m in _RCCL_RIG_ERRORS):
        if os.environ.get("FK_REQUIRE_RANKS") == "1":
            pytest.fail("RCCL would not bring up %d ranks on one GPU: %s" % (ranks, p.stderr[-500:]))
        pytest.skip("RCCL would not bring up several ranks on one GPU here: " + p.stderr[-300:])
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    lines = [x for x in p.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == ranks and out["scaling"] == "strong" and "configs[3]" in out["config"]["workload"]
    L, k = sy["read_len"], case["k"]
    assert sy["nreads"] % ranks == 0
    assert out["config"]["kmer_instances"] == sy["nreads"] * (L - k + 1)
    assert out["gather_ms"] > 0 and out["count_ms"] > 0 and abs(out["gather_ms"] + out["count_ms"] - out["ms_per_step"]) < 1.0
    assert out["config"]["gathered_entries"] == out["config"]["table_entries"] == exp["ktab"]["nels"] and "final gather" in out["metric"]
    assert out["hist_file_sha256"] == exp["hist_sha256"], "the sharded run's .hist is not the reference's"
    assert out["rccl_ranks"] == ranks and len(out["exchange"]) == ranks
    for r, ex in enumerate(out["exchange"]):
        assert ex["comm_ranks"] == ranks and ex["rounds"] >= 4 and ex["kept_bytes"] > 0
        assert (ex["sent_bytes"] > 0) == (ranks > 1) and ex["sent_bytes"] % 20 == 0 and ex["exchange_ms"] > 0
    assert sum(ex["sent_bytes"] for ex in out["exchange"]) == sum(ex["recv_bytes"] for ex in out["exchange"])
    kw = (2 * k + 7) // 8 + 2
    table = np.concatenate([np.fromfile(os.path.join(str(tmp_path), "table.%d" % r), dtype=np.uint8).reshape(-1, kw)
                            for r in range(ranks)])
    assert table.shape[0] == exp["ktab"]["nels"]
    assert orc.table_stream_sha256(k, table) == exp["ktab"]["stream_sha256"], "the gathered table is not the reference's"


def test_sharded_final_gather_writes_reference_files(tmp_path):
    """count_sharded(fetch_table) -> gather_table -> write_files on a one-rank RCCL group: the files
    are the golden ones; HipEngine.sort_table (the re-ordering rank 0 does for world > 1) restores
    a shuffled table."""
    import hashlib, os
    import torch
    import torch.distributed as dist
    from tests import shard_model as shard
    case, bases, boff = util.load_case("synth_illumina_k40_t1_T4")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29591")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        with fastk_amd.Context(kmer=case["k"], table_cutoff=case["cutoff"], nbuckets=1) as ctx:
            eng = shard.HipEngine(ctx, torch.device("cuda", 0))
            reads = torch.from_numpy(bases).cuda()
            out = shard.count_sharded(eng, reads, verify=True, fetch_table=True)
            table = out["local"]["result"].table
            merged = shard.gather_table(table, ctx.w.kmer_bytes, eng.sort_table)
            rng = np.random.default_rng(5)
            again = eng.sort_table(np.ascontiguousarray(merged[rng.permutation(len(merged))]))
            assert np.array_equal(again, merged)
            # profiles of the rank's reads against the union of all ranks' tables (here: one rank)
            shard.write_table_sharded(table, out["wfirst"], out["ntable"], case["k"], case["cutoff"],
                                      case["T"], str(tmp_path), "y", eng.sort_table)
            pdata, poffs = shard.profiles_sharded(eng, reads, table)
            mine = [orc.profile_decode(pdata.tobytes()[poffs[i]:poffs[i + 1]]) for i in range(len(poffs) - 1)]
            assert orc.profiles_digest(mine) == case["expected"]["prof"]["decoded_sha256"]
    finally:
        dist.destroy_process_group()
    util.check_against_golden(case, out["hist"], out["max_inst"], merged)
    fastk_amd.write_files(case["k"], case["cutoff"], case["T"], out["hist"], out["max_inst"], merged,
                          str(tmp_path), "x", wfirst=out["wfirst"])
    exp = case["expected"]
    assert hashlib.sha256(open(tmp_path / "x.hist", "rb").read()).hexdigest() == exp["hist_sha256"]
    assert orc.read_ktab(str(tmp_path / "x"))["stream_sha256"] == exp["ktab"]["stream_sha256"]
    for f in ["%s.ktab"] + [".%%s.ktab.%d" % (i + 1) for i in range(case["T"])]:     # written by the rank(s)
        assert open(tmp_path / (f % "y"), "rb").read() == open(tmp_path / (f % "x"), "rb").read(), f


@pytest.mark.parametrize("name,nb", [("synth_illumina_k40_t1_T4", 4), ("edge_k40_t1_T4", 8),
                                     ("edge_k51_t1_T4", 3)])
def test_bucketed_split_keeps_equal_kmers_together(name, nb):
    """nbuckets > 1 (the sharded path): records come out grouped by bucket, every k-mer instance is
    in exactly one record, and no canonical k-mer appears in two buckets -- so counting the buckets
    independently and merging the tables reproduces the reference."""
    case, bases, boff = util.load_case(name)
    k = case["k"]
    P = orc.params(k)
    with fastk_amd.Context(kmer=k, table_cutoff=case["cutoff"], nbuckets=nb) as ctx:
        w = ctx.w
        rd = ctx.alloc(len(bases) + 64).upload(bases)
        ns, ni, counts = ctx.split(rd.ptr, len(bases))
        assert sum(counts) == ns and len(counts) == nb
        out = ctx.alloc(max(ns, 1) * w.smer_stride)
        ctx.split_emit(rd.ptr, len(bases), out.ptr, ns, counts)
        recs = out.download(ns * w.smer_stride).reshape(ns, w.smer_stride)[:, :w.smer_word]
    hist = np.zeros(0x8000, dtype=np.int64)
    max_inst = 0
    tables = []
    lo = 0
    for c in counts:
        part = np.ascontiguousarray(recs[lo:lo + c])
        lo += c
        kl, ovf, _ = orc.kmer_list(P, orc.msd_sort(part, P.smer_word))
        res = orc.count_sorted(P, orc.msd_sort(kl, P.kmer_bytes), case["cutoff"])
        hist += res.hist
        max_inst += res.max_inst + ovf
        tables.append(res.table)
    merged = np.concatenate([t for t in tables if len(t)])
    order = np.lexsort(merged[:, :P.kmer_bytes].T[::-1])
    merged = merged[order]
    # disjoint buckets: no k-mer twice in the merged table
    assert not np.any(np.all(merged[1:, :P.kmer_bytes] == merged[:-1, :P.kmer_bytes], axis=1))
    util.check_against_golden(case, hist, max_inst, merged)


def test_planned_bucketed_split_matches_exact_counts():
    """fk_split_plan + fk_split_planned (one emit pass into sampled, padded regions) must deliver the
    same per-bucket multisets as the exact count-then-emit pair."""
    case, bases, boff = util.load_case("synth_illumina_k40_t1_T4")
    with fastk_amd.Context(kmer=40, nbuckets=4) as ctx:
        w = ctx.w
        rd = ctx.alloc(len(bases) + 64).upload(bases)
        ns, ni, counts = ctx.split(rd.ptr, len(bases))
        exact = ctx.alloc(ns * w.smer_stride)
        ctx.split_emit(rd.ptr, len(bases), exact.ptr, ns, counts)
        ex = exact.download(ns * w.smer_stride).reshape(ns, w.smer_stride)
        cap, offs = ctx.split_plan(rd.ptr, len(bases))
        assert cap >= ns and len(offs) == 5
        buf = ctx.alloc(cap * w.smer_stride)
        got = ctx.split_planned(rd.ptr, len(bases), buf.ptr, cap, offs)
        assert got is not None
        pcounts, pni = got
        assert pcounts == counts and pni == ni
        pl = buf.download(cap * w.smer_stride).reshape(cap, w.smer_stride)
    lo = 0
    for b, c in enumerate(counts):
        a = orc.msd_sort(np.ascontiguousarray(ex[lo:lo + c]), w.smer_stride)
        p = orc.msd_sort(np.ascontiguousarray(pl[offs[b]:offs[b] + c]), w.smer_stride)
        assert np.array_equal(a, p), b
        lo += c


def test_planned_split_reports_a_region_that_is_too_small():
    """The streamed emit (eight interleaved cursors per bucket, ragged ends closed afterwards) must notice a
    region that cannot hold its bucket -- FK_ESTATE, the caller then takes the exact two-call path -- and must
    not write outside the buffer while doing so (the next call on the same context still gives exact results)."""
    case, bases, boff = util.load_case("synth_hifi_k40_t4_T8")
    with fastk_amd.Context(kmer=40, nbuckets=6) as ctx:
        w = ctx.w
        rd = ctx.alloc(len(bases) + 64).upload(bases)
        ns, ni, counts = ctx.split(rd.ptr, len(bases))
        cap, offs = ctx.split_plan(rd.ptr, len(bases))
        tight = [0]
        for c in counts:                       # half of what each bucket needs
            tight.append(tight[-1] + max(c // 2, 1))
        buf = ctx.alloc(cap * w.smer_stride)
        assert ctx.split_planned(rd.ptr, len(bases), buf.ptr, tight[-1], tight) is None
        got = ctx.split_planned(rd.ptr, len(bases), buf.ptr, cap, offs)
        assert got is not None and got[0] == counts and got[1] == ni


# ------------------------------------------------------------------------------ exact part files

@pytest.mark.parametrize("name", util.golden_names())
def test_exact_parts_mode_reproduces_reference_files(name, tmp_path):
    """exact_parts=1 replays the reference's own super-mer rule on the GPU, so the weighted k-mer
    first-byte census -- and with it every hidden .ktab part boundary -- is the reference's:
    all output files are byte-identical to reference FastK's (sha256 from tests/golden)."""
    import hashlib
    case, bases, boff = util.load_case(name)
    k, T = case["k"], case["T"]
    with fastk_amd.Context(kmer=k, table_cutoff=case["cutoff"], nthreads=T, exact_parts=True) as ctx:
        nreads = len(boff) - 1
        step = max(1, nreads // 3)
        for s in range(0, nreads, step):
            e = min(nreads, s + step)
            ctx.push_block(bases[boff[s]:boff[e]], (boff[s:e + 1] - boff[s]).astype(np.int32))
        res = ctx.finish()
        util.check_against_golden(case, res.hist, res.max_inst, res.table)
        ctx.write_hist(res, str(tmp_path / "x.hist"))
        ctx.write_ktab(res, str(tmp_path), "x")
    for fname, digest in case["expected"]["file_sha256"].items():
        got = hashlib.sha256(open(tmp_path / fname, "rb").read()).hexdigest()
        assert got == digest, fname


@pytest.mark.parametrize("knobs", [dict(exact_chain=1), dict(exact_chain=2), dict(exact_segments=0),
                                   dict(exact_chain=1, exact_segments=0)], ids=lambda d: ",".join("%s=%d" % kv for kv in d.items()))
@pytest.mark.parametrize("name", ["edge_k40_t1_T4", "synth_hifi_k40_t4_T8", "synth_illumina_k51_t1_T4", "edge_k21_t2_T3"])
def test_exact_splitter_paths_give_the_same_files(name, knobs, tmp_path):
    """k_split_exact keeps the chain of minimizers behind the current one in registers (6 entries) and walks its ring, as
    the reference does at every forced closing (split.c:1304-1320), only when that chain ran out; long reads are cut into
    segments where the reference's state is known.  With the chain shortened to 1 or 2 entries the ring walk is the common
    path, without segments a thread follows a whole read: every output file is still the reference's."""
    case, bases, boff = util.load_case(name)
    with fastk_amd.Context(kmer=case["k"], table_cutoff=case["cutoff"], nthreads=case["T"], exact_parts=True) as ctx:
        for key, v in knobs.items():
            ctx.debug_set(key, v)
        _push_in_pieces(ctx, bases, boff, 2)
        res = ctx.finish()
        util.check_against_golden(case, res.hist, res.max_inst, res.table)
        ctx.write_hist(res, str(tmp_path / "x.hist"))
        ctx.write_ktab(res, str(tmp_path), "x")
    for fname, digest in case["expected"]["file_sha256"].items():
        assert util.sha_file(tmp_path / fname) == digest, fname


@pytest.mark.parametrize("exact", [False, True], ids=["default", "exact_parts"])
@pytest.mark.parametrize("k", [8, 9, 11, 12])
def test_smallest_kmer_sizes(k, exact):
    """k = 8 is the lower limit fk_create accepts: its super-mers are records of ONE 32-bit word (11 bases and the length
    byte), the only k whose records are -- the expansion was not built for them until round 5 (`super-mer stride 4 not
    built`).  Random reads of 5 to 400 bases against the oracle, default splitter and exact_parts."""
    rng = np.random.default_rng(800 + k)
    reads = [bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(rng.integers(5, 400)))])
             for _ in range(300)]
    bases, boff = orc.block_from_reads(reads)
    o = orc.fastk(k, bases, boff, cutoff=1, nthreads=2)
    with fastk_amd.Context(kmer=k, table_cutoff=1, nthreads=2, exact_parts=exact) as ctx:
        ctx.push_block(bases, boff.astype(np.int32))
        res = ctx.finish()
    assert np.array_equal(res.hist, o.hist) and res.max_inst == o.max_inst
    assert np.array_equal(res.table, o.table)


@pytest.mark.parametrize("what", ["three reads, one shorter than k, -p", "short reads only, -p", "two reads for four ranks",
                                  "reads shorter than k among others, -t2 -p"])
def test_cli_degenerate_inputs_one_gpu_and_sharded(what):
    """Files so small that a rank of FastK_amd -G4 gets no read, or only reads without a k-mer (their profiles have no
    bytes: round 5 found the driver calling that `Out of memory`): FastK_amd, -G2 and -G4 (ranks sharing the GPU) and the
    reference itself agree on .hist, the .ktab stream and the decoded profiles (tools/cli_degenerate_probe.py holds more
    of these).  Where the reference dies of the input (only reads shorter than k) ours agree among themselves."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("cli_degenerate_probe", os.path.join(root, "tools", "cli_degenerate_probe.py"))
    probe = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(probe)
    if not os.path.exists(probe.REF):
        util.no_reference("oracle/_ref/FastK not built")
    rng = np.random.default_rng(11)
    r = lambda n: probe.rnd(rng, n)
    reads, flags = {"three reads, one shorter than k, -p": ([r(100), r(30), r(3000)], ("-t1", "-p")),
                    "short reads only, -p": ([r(30), r(20)], ("-t1", "-p")),
                    "two reads for four ranks": ([r(200), r(200)], ("-t1",)),
                    "reads shorter than k among others, -t2 -p": ([r(100)] * 2 + [r(30), r(3000), r(41)], ("-t2", "-p"))}[what]
    ok, res = probe.case(what, reads, flags=flags)
    if not ok and any(m in str(res) for m in _RCCL_RIG_ERRORS):
        pytest.skip("RCCL would not bring up several ranks on one GPU here")
    assert ok, res


def test_exact_parts_long_reads_in_several_buckets():
    """The two halves of the exact splitter that the other tests take one at a time: long reads (cut into segments) AND a
    sort memory so small that the reference deals the minimizers to several buckets (the trie walk per super-mer).
    200 M bases in 15 kbp reads, FastK_amd -x -M1 (-T4: 3 buckets) and -M2 (-T3) against the reference run live with the
    same options: every output file, hidden parts included (tools/exact_long_reads_buckets_probe.py)."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not orc.have_ref():
        util.no_reference("oracle/_ref/FastK not built")
    spec = importlib.util.spec_from_file_location("xlb", os.path.join(root, "tools", "exact_long_reads_buckets_probe.py"))
    probe = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(probe)
    assert probe.run(200.0, 15000) == 0
    assert probe.run(200.0, 15000, ("-p",)) == 0         # ... and with profiles: the .prof / .pidx parts of four input threads too


def test_exact_parts_with_profiles_every_file_of_the_reference():
    """-x -p: under -p the reference keeps its super-mers on the read's strand (split.c:1245, Stuff_Seq(..., 0, ...)), so a
    super-mer and its reverse complement are two records, the weighted k-mer list is another one and the hidden .ktab
    parts are cut at other first bytes than in a run without -p -- fk_params.exact_parts = 2; and its .prof parts are the
    byte ranges its input threads read (io.c:2420-2521, fast_nearest), with run numbers per thread.  FastK_amd -x -p and
    the reference's main() over the shim (FASTK_AMD_EXACT=1 -p) against the reference run live on reads full of ties,
    one and several input threads, FASTA and FASTQ: every file (tools/exact_prof_low_complexity_probe.py)."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not (orc.have_ref() and os.path.exists(os.path.join(orc.REF_DIR, "FastK_gpu"))):
        util.no_reference("oracle/_ref/FastK, FastK_gpu not built")
    spec = importlib.util.spec_from_file_location("xp", os.path.join(root, "tools", "exact_prof_low_complexity_probe.py"))
    probe = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(probe)
    assert probe.run() == 0


def test_exact_parts_with_the_other_options():
    """-x with -bc<n>, -c, -t<n>, each with and without -p, FASTA and FASTQ: FastK_amd and the reference's main() over the
    shim against the reference run live with the same options on reads full of ties -- every output file
    (tools/exact_flags_probe.py)."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not (orc.have_ref() and os.path.exists(os.path.join(orc.REF_DIR, "FastK_gpu"))):
        util.no_reference("oracle/_ref/FastK, FastK_gpu not built")
    spec = importlib.util.spec_from_file_location("xf", os.path.join(root, "tools", "exact_flags_probe.py"))
    probe = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(probe)
    assert probe.run() == 0


def _low_complexity_reads(seed, nreads=260, lengths=(40, 60, 150, 400, 1500, 6000)):
    """Reads made of what breaks ties in a minimizer scheme: homopolymers, di-/tri-/tetra-nucleotide repeats (among
    them the ones equal to their own reverse complement), copies of one short motif with a few substitutions, runs of N,
    random stretches in between -- 40 to 6000 bases, so that some are cut into segments by the exact splitter."""
    rng = np.random.default_rng(seed)
    units = ["A", "T", "C", "AC", "AT", "CG", "GA", "AAT", "ACG", "ACGT", "AATT", "GATC", "AACCGGTT", "ACACACGT"]
    reads = []
    for _ in range(nreads):
        want = int(rng.choice(lengths))
        parts, n = [], 0
        while n < want:
            kind = rng.integers(0, 10)
            ln = int(rng.integers(5, 300))
            if kind < 6:
                u = units[int(rng.integers(0, len(units)))]
                piece = np.frombuffer((u * (ln // len(u) + 1))[:ln].encode(), dtype=np.uint8).copy()
                if kind >= 4 and ln > 10:                       # a few substitutions
                    for q in rng.integers(0, ln, size=max(1, ln // 40)):
                        piece[q] = ord("ACGT"[int(rng.integers(0, 4))])
            elif kind < 9:
                piece = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=ln)]
            else:
                piece = np.full(int(rng.integers(1, 60)), ord("N"), dtype=np.uint8)
            parts.append(piece)
            n += len(piece)
        reads.append(np.concatenate(parts)[:want].tobytes())
    return orc.block_from_reads(reads)


@pytest.mark.parametrize("knobs", [dict(), dict(exact_chain=1), dict(exact_segments=0)],
                         ids=lambda d: ",".join("%s=%d" % kv for kv in d.items()) or "default")
@pytest.mark.parametrize("k,T,seed", [(40, 4, 1), (21, 1, 2), (51, 3, 3), (64, 2, 4), (33, 8, 5), (40, 2, 106), (27, 4, 107),
                                      (8, 2, 8), (12, 4, 9), (16, 1, 10)])     # (k = 8: one-word records, packed in the walk)
def test_exact_splitter_on_low_complexity_reads(k, T, seed, knobs, tmp_path):
    """The reference run live (oracle/_ref/FastK) on reads full of ties -- equal minimizer values inside one window are
    what the `<` on arrival / `<=` on the forced rescan of split.c:1149,1306-1315 decide, and what the register chain
    of k_split_exact has to reproduce -- against exact_parts: every output file byte for byte."""
    if not orc.have_ref():
        util.no_reference("oracle/_ref/FastK not built (needs the reference sources at build time)")
    bases, boff = _low_complexity_reads(20260000 + seed) if seed < 100 else \
                  _low_complexity_reads(20260000 + seed, 60, (39, 5000, 30000, 120000))     # (many segments per read)
    ref = tmp_path / "ref"
    ref.mkdir()
    orc.write_fasta(str(ref / "x.fasta"), bases, boff, width=0)
    orc.run_ref_fastk(str(ref / "x.fasta"), k, 1, T, str(ref))
    ours = tmp_path / "ours"
    ours.mkdir()
    with fastk_amd.Context(kmer=k, table_cutoff=1, nthreads=T, exact_parts=True) as ctx:
        for key, v in knobs.items():
            ctx.debug_set(key, v)
        _push_in_pieces(ctx, bases, boff, 2)
        res = ctx.finish()
        ctx.write_hist(res, str(ours / "x.hist"))
        ctx.write_ktab(res, str(ours), "x")
    names = sorted(f for f in os.listdir(ref) if f != "x.fasta")
    assert "x.hist" in names and "x.ktab" in names and len(names) == 2 + T, names
    for f in names:
        assert open(ours / f, "rb").read() == open(ref / f, "rb").read(), f


@pytest.mark.parametrize("name,fmt", [("edge_k40_t1_T4", "fasta"), ("synth_illumina_k51_t1_T4", "fasta"),
                                      ("synth_hifi_k40_t4_T8", "fasta")])
def test_reference_main_over_gpu_shim(name, fmt, tmp_path):
    """oracle/_ref/FastK_gpu = the REFERENCE's own main(), option parser and multi-threaded input layer
    (FastK.c, io.c compiled where they lie) linked against libfastk_amd.so through the INTEGRATION.md
    shim.  By default it takes the fast splitter: .hist bytes, the .ktab stub fields and the canonical stream are
    the reference's.  With FASTK_AMD_EXACT=1 every output file is the reference's, byte for byte."""
    import hashlib, os, subprocess
    exe = os.path.join(orc.REF_DIR, "FastK_gpu")
    if not os.path.exists(exe):
        util.no_reference("oracle/_ref/FastK_gpu not built (needs the reference sources at build time)")
    case, bases, boff = util.load_case(name)
    exp = case["expected"]
    for mode in ("fast", "exact"):
        d = tmp_path / mode
        d.mkdir()
        path = str(d / ("x." + fmt))
        orc.write_fasta(path, bases, boff, width=0 if case["kind"] == "edge" else 100)
        env = dict(os.environ)
        env.pop("FASTK_AMD_EXACT", None)
        if mode == "exact":
            env["FASTK_AMD_EXACT"] = "1"
        subprocess.run([exe, "-k%d" % case["k"], "-t%d" % case["cutoff"], "-T%d" % case["T"],
                        "-P" + str(d), path], check=True, cwd=str(d), env=env,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        assert util.sha_file(d / "x.hist") == exp["hist_sha256"], mode
        t = orc.read_ktab(str(d / "x"))
        assert t["stream_sha256"] == exp["ktab"]["stream_sha256"], mode
        assert (t["kmer"], t["nparts"], t["minval"], t["ibytes"], t["nels"]) == \
            (case["k"], case["T"], case["cutoff"], exp["ktab"]["ibytes"], exp["ktab"]["nels"]), mode
        if mode == "exact":
            for fname, digest in exp["file_sha256"].items():
                assert util.sha_file(d / fname) == digest, fname


@pytest.mark.parametrize("name", util.golden_names())
def test_reference_main_with_profiles_over_gpu_shim(name, tmp_path):
    """The reference's main() with -p on the GPU path: .hist and every .ktab file are still the
    reference's bytes (table filtered to -t on the way out), the profiles decode to the reference's
    (golden digest), and -p:<table> of the same reads against that table gives its counts."""
    import hashlib, os, subprocess
    exe = os.path.join(orc.REF_DIR, "FastK_gpu")
    if not os.path.exists(exe):
        util.no_reference("oracle/_ref/FastK_gpu not built (needs the reference sources at build time)")
    case, bases, boff = util.load_case(name)
    k = case["k"]
    path = str(tmp_path / ("x." + case["fmt"]))          # the golden's own file layout: io.c deals the reads to its
    if case["fmt"] == "fasta":                            # threads by file bytes, and the profile parts follow
        orc.write_fasta(path, bases, boff, width=0 if case["kind"] == "edge" else 100)
    else:
        orc.write_fastq(path, bases, boff)
    subprocess.run([exe, "-k%d" % k, "-t%d" % case["cutoff"], "-T%d" % case["T"], "-p",
                    "-P" + str(tmp_path), path], check=True, cwd=str(tmp_path), env=dict(os.environ, FASTK_AMD_EXACT="1"),
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    # (the golden's .ktab digests are those of a run WITHOUT -p: .hist, the stub and the canonical stream do not depend on
    # -p, the hidden parts do -- under -p the reference keeps its super-mers on the read's strand, split.c:1245, and cuts
    # the parts by that list's census; they are compared with the reference run live below)
    for fname, digest in case["expected"]["file_sha256"].items():
        if not fname.startswith(".x.ktab."):
            assert hashlib.sha256(open(tmp_path / fname, "rb").read()).hexdigest() == digest, fname
    assert orc.read_ktab(str(tmp_path / "x"))["stream_sha256"] == case["expected"]["ktab"]["stream_sha256"]
    if orc.have_ref():
        live = tmp_path / "live"
        live.mkdir()
        os.link(path, live / os.path.basename(path))
        subprocess.run([os.path.join(orc.REF_DIR, "FastK"), "-k%d" % k, "-t%d" % case["cutoff"], "-T%d" % case["T"], "-p",
                        "-P" + str(live), str(live / os.path.basename(path))], check=True, cwd=str(live),
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        names = sorted(f for f in os.listdir(live) if f != os.path.basename(path))
        assert len(names) >= 4 + case["T"]
        for f in names:
            assert util.sha_file(live / f) == util.sha_file(tmp_path / f), f
    # in exact mode the .prof files are the reference's too, byte for byte: its super-mer junctions (a difference of -31
    # takes two bytes there, merge.c:456,590) and its panel flushes (a pending run is written out every 1024 NPARTS
    # runs per input thread, merge.c:263-267,706-716) are replayed by k_pf_exact
    for fname, digest in case["expected"]["prof"]["file_sha256"].items():
        assert util.sha_file(tmp_path / fname) == digest, fname
    kk, enc = orc.read_profiles(str(tmp_path), "x")
    assert kk == k and len(enc) == case["expected"]["prof"]["nreads"]
    assert orc.profiles_digest([orc.profile_decode(e) for e in enc]) == case["expected"]["prof"]["decoded_sha256"]
    # relative to its own cutoff-t table: counts below the cutoff read as 0
    rel = str(tmp_path / ("y." + case["fmt"]))
    os.link(path, rel)
    subprocess.run([exe, "-k%d" % k, "-T2", "-p:x", "-P" + str(tmp_path), rel], check=True, cwd=str(tmp_path),
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    assert not os.path.exists(tmp_path / "y.hist")
    k2, renc = orc.read_profiles(str(tmp_path), "y")
    full = [orc.profile_decode(e) for e in enc]
    want = [[c if c >= case["cutoff"] else 0 for c in r] for r in full]
    assert [orc.profile_decode(e) for e in renc] == want


def test_trained_bucket_assignment_balances_and_keeps_results():
    """fk_bucket_census + fk_set_bucket_weights: the buckets get within a few percent of each other
    and the bucket-streamed result is still the reference's (any assignment is valid)."""
    case, bases, boff = util.load_case("synth_illumina_k40_t1_T4")
    with fastk_amd.Context(kmer=40, table_cutoff=case["cutoff"], nthreads=case["T"], nbuckets=8) as ctx:
        ctx.set_bucket_weights(ctx.bucket_census(bases[:1 << 20]))
        rd = ctx.alloc(len(bases) + 64).upload(bases)
        ns, ni, counts = ctx.split(rd.ptr, len(bases))
        c = np.array(counts, dtype=float)
        assert c.max() / c.mean() < 1.12 and c.min() / c.mean() > 0.88   # work, not records, is balanced
        ctx.push_block(bases, boff.astype(np.int32))
        res = ctx.finish()
        util.check_against_golden(case, res.hist, res.max_inst, res.table)


@pytest.mark.parametrize("name,nb,passes", [("synth_illumina_k40_t1_T4", 8, 2), ("synth_illumina_k40_t1_T4", 5, 5),
                                            ("synth_hifi_k40_t4_T8", 12, 3), ("edge_k40_t1_T4", 7, 2),
                                            ("synth_illumina_k51_t1_T4", 6, 4)])
def test_multi_pass_split_over_resident_reads(name, nb, passes):
    """fk_params.split_passes: the reads stay resident and are split `passes` times, each pass keeping the
    super-mers of one group of minimizer buckets only (how BASELINE configs[2] fits one GPU: 150 GB of
    reads + 166 GB of records do not).  Histogram, totals and table must be the reference's."""
    case, bases, boff = util.load_case(name)
    with fastk_amd.Context(kmer=case["k"], table_cutoff=case["cutoff"], nthreads=case["T"], nbuckets=nb,
                           split_passes=passes) as ctx:
        rd = ctx.alloc(len(bases) + 64).upload(bases)
        res = ctx.count_device_reads(rd.ptr, len(bases), fetch_table=True)
        assert 1 < res.split_passes <= passes and res.buckets_counted <= nb
        assert res.replay_passes == res.split_passes - 1       # later passes rebuild records from recorded entries
        util.check_against_golden(case, res.hist, res.max_inst, res.table)
        ctx.debug_set("split_replay", 0)                        # every pass recomputes the minimizers: same result
        full = ctx.count_device_reads(rd.ptr, len(bases), fetch_table=True)
        assert full.replay_passes == 0 and full.split_passes == res.split_passes
        assert np.array_equal(full.hist, res.hist) and np.array_equal(full.table, res.table) and full.nsuper == res.nsuper
        ctx.debug_set("split_replay", 1)
        one = ctx.count_device_reads(rd.ptr, len(bases), fetch_table=True)      # arena re-use: same again
        assert np.array_equal(one.hist, res.hist) and np.array_equal(one.table, res.table)
        assert one.nsuper == res.nsuper and one.ninst == res.ninst
    with fastk_amd.Context(kmer=case["k"], table_cutoff=case["cutoff"], nthreads=case["T"], nbuckets=nb) as ctx:
        rd = ctx.alloc(len(bases) + 64).upload(bases)
        ref = ctx.count_device_reads(rd.ptr, len(bases), fetch_table=False)
        assert ref.split_passes == 1
        assert (ref.nsuper, ref.ninst, ref.nweighted, ref.ndistinct) == (res.nsuper, res.ninst, res.nweighted,
                                                                         res.ndistinct)


def test_multi_pass_split_automatic_from_budget():
    """split_passes = 0 with an hbm_budget: as many passes as it takes for one pass's records to fit
    half the budget."""
    case, bases, boff = util.load_case("synth_illumina_k40_t1_T4")
    with fastk_amd.Context(kmer=40, table_cutoff=case["cutoff"], nthreads=case["T"], nbuckets=16,
                           hbm_budget=6 << 20) as ctx:
        rd = ctx.alloc(len(bases) + 64).upload(bases)
        res = ctx.count_device_reads(rd.ptr, len(bases), fetch_table=True)
        assert res.split_passes > 1
        util.check_against_golden(case, res.hist, res.max_inst, res.table)


@pytest.mark.parametrize("rsize,n,keys", [(12, 400003, 10), (20, 150001, 19), (12, 100000, 5), (28, 30000, 25)])
def test_lsd_sort_equals_reference_engine(ctx40, rsize, n, keys):
    """fk_lsd_sort_records against the REFERENCE's LSD_Sort itself (oracle/_ref/libfkref.so, LSDsort.c:115
    compiled where it lies), not only against the restatement: same bytes, ties in input order."""
    if not orc.have_fkref():
        util.no_reference("oracle/_ref/libfkref.so not built (needs the reference sources at build time)")
    rng = np.random.default_rng(n)
    recs = rng.integers(0, 256, size=(n, rsize), dtype=np.uint8)
    recs[:, 0] = rng.integers(0, 4, size=n)
    order = list(range(keys - 1, -1, -1))
    a = ctx40.alloc(recs.nbytes).upload(recs)
    b = ctx40.alloc(recs.nbytes)
    res = ctx40.lsd_sort(a.ptr, b.ptr, n, rsize, order)
    got = a.download(recs.nbytes, ptr=res).reshape(n, rsize)
    a.free(); b.free()
    assert np.array_equal(got, orc.ref_lsd_sort(recs, order, 5))


# ------------------------------------------------------------------------------ above fixture size
# Digest-only golden cases made by the reference (tests/golden/make_golden.py --large): BASELINE.json
# configs[0] at its stated size and the two samples bench.py times the reference on.

LARGE = ["configs0_k40_t1_T4", "hifi50x20M_k40_t4_T4", "illumina50x20M_k40_t1_T4"]


def _push_in_pieces(ctx, bases, boff, pieces):
    nreads = len(boff) - 1
    step = max(1, (nreads + pieces - 1) // pieces)
    for s0 in range(0, nreads, step):
        e = min(nreads, s0 + step)
        ctx.push_block(bases[boff[s0]:boff[e]], (boff[s0:e + 1] - boff[s0]).astype(np.int32))


@pytest.mark.parametrize("name", LARGE)
def test_reference_digests_above_fixture_size(name, tmp_path):
    """Library path against the reference's digests at BASELINE configs[0] size and above: the default
    pipeline (position-parallel split, hash aggregation; resident and bucket-streamed with a budget)
    must give the reference's .hist bytes and .ktab canonical stream, and exact_parts every file."""
    case, bases, boff = util.load_case(name)
    k, T, cutoff = case["k"], case["T"], case["cutoff"]
    with fastk_amd.Context(kmer=k, table_cutoff=cutoff, nthreads=T) as ctx:
        _push_in_pieces(ctx, bases, boff, 7)
        res = ctx.finish()
        util.check_against_golden(case, res.hist, res.max_inst, res.table)
    with fastk_amd.Context(kmer=k, table_cutoff=cutoff, nthreads=T, nbuckets=5, hbm_budget=2 << 30) as ctx:
        _push_in_pieces(ctx, bases, boff, 7)
        res2 = ctx.finish()
        assert np.array_equal(res2.hist, res.hist) and res2.max_inst == res.max_inst
        assert np.array_equal(res2.table, res.table)
    del res, res2
    with fastk_amd.Context(kmer=k, table_cutoff=cutoff, nthreads=T, exact_parts=True) as ctx:
        _push_in_pieces(ctx, bases, boff, 3)
        res = ctx.finish()
        ctx.write_hist(res, str(tmp_path / "x.hist"))
        ctx.write_ktab(res, str(tmp_path), "x")
    for fname, digest in case["expected"]["file_sha256"].items():
        assert util.sha_file(tmp_path / fname) == digest, fname


@pytest.mark.parametrize("name", ["configs0_k40_t1_T4_M1", "illumina50x20M_k40_t1_T4_M3", "illumina50x20M_k40_t1_T4_M1",
                                  "illumina50x30M_k40_t1_T4_M1"])
def test_exact_parts_follows_the_reference_scheme_to_several_buckets(name, tmp_path):
    """with a small sort memory the reference cuts its input into NPARTS > 1 buckets -- 2 at
    BASELINE configs[0] with -M1 (unpadded trie, leaves dealt by assign_pieces' drand48 draws), 3 and 9 at the 6.7 M-read
    sample with -M3 / -M1, 14 at a 10 M-read sample with -M1 (first block = two thirds of the input, and Determine_Scheme
    pads seven heavy minimizers to 7 bases) -- and cuts the hidden .ktab part files by the threads that sort BUCKET 0's
    weighted k-mers.  exact_parts with the same sort memory retrains that
    scheme (fk_scheme.hip) and writes every file of the reference byte for byte: through the library, FastK_amd -x -M
    and the reference's own main() over the shim (FASTK_AMD_EXACT=1)."""
    import os, subprocess
    case, bases, boff = util.load_case(name)
    exp = case["expected"]
    k, T, cutoff = case["k"], case["T"], case["cutoff"]
    mem = int(case["ref_extra"][0][2:])
    lib_dir = tmp_path / "lib"
    lib_dir.mkdir()
    with fastk_amd.Context(kmer=k, table_cutoff=cutoff, nthreads=T, exact_parts=True) as ctx:
        ctx._ck(ctx.L.fk_set_sort_memory(ctx.h, mem * 1000000000, 0.0))
        _push_in_pieces(ctx, bases, boff, 3)
        res = ctx.finish()
        assert ctx.debug_get("scheme_nparts") > 1
        util.check_against_golden(case, res.hist, res.max_inst, res.table)
        ctx.write_hist(res, str(lib_dir / "x.hist"))
        ctx.write_ktab(res, str(lib_dir), "x")
        del res
    for fname, digest in exp["file_sha256"].items():
        assert util.sha_file(lib_dir / fname) == digest, ("library", fname)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = str(tmp_path / ("x." + case["fmt"]))
    util.write_fastx(path, bases, boff, case["fmt"] == "fastq")
    args = ["-k%d" % k, "-t%d" % cutoff, "-T%d" % T, "-M%d" % mem]
    for cmd in ([util.driver_exe("FastK_amd")] + args + ["-x", path],
                [os.path.join(orc.REF_DIR, "FastK_gpu")] + args + ["-P" + str(tmp_path), path]):
        if not os.path.exists(cmd[0]):
            continue
        for f in os.listdir(tmp_path):
            if f not in (os.path.basename(path), "lib"):
                os.remove(tmp_path / f)
        subprocess.run(cmd, check=True, cwd=str(tmp_path), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                       env=dict(os.environ, FASTK_AMD_EXACT="1"))
        for fname, digest in exp["file_sha256"].items():
            assert util.sha_file(tmp_path / fname) == digest, (os.path.basename(cmd[0]), fname)


@pytest.mark.parametrize("name", LARGE[:2])
def test_drivers_against_reference_digests_above_fixture_size(name, tmp_path):
    """The C driver (FastK_amd, text parsed on the GPU; then -x) and the reference's own main() over the
    shim (FastK_gpu) on the same inputs as files: stream digests, then every file digest."""
    import os, subprocess
    case, bases, boff = util.load_case(name)
    exp = case["expected"]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = util.driver_exe("FastK_amd")
    path = str(tmp_path / ("x." + case["fmt"]))
    util.write_fastx(path, bases, boff, case["fmt"] == "fastq")
    args = ["-k%d" % case["k"], "-t%d" % case["cutoff"], "-T%d" % case["T"]]
    subprocess.run([exe] + args + [path], check=True, cwd=str(tmp_path))
    assert util.sha_file(tmp_path / "x.hist") == exp["hist_sha256"]
    t = orc.read_ktab(str(tmp_path / "x"))
    assert t["stream_sha256"] == exp["ktab"]["stream_sha256"] and t["nels"] == exp["ktab"]["nels"]
    del t
    for cmd in ([exe] + args + ["-x", path], [os.path.join(orc.REF_DIR, "FastK_gpu")] + args + ["-P" + str(tmp_path), path]):
        if not os.path.exists(cmd[0]):
            continue
        for f in os.listdir(tmp_path):
            if f != os.path.basename(path):
                os.remove(tmp_path / f)
        subprocess.run(cmd, check=True, cwd=str(tmp_path), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                       env=dict(os.environ, FASTK_AMD_EXACT="1"))
        for fname, digest in exp["file_sha256"].items():
            assert util.sha_file(tmp_path / fname) == digest, (os.path.basename(cmd[0]), fname)


def test_reference_digests_above_4GiB(tmp_path):
    """hifi50x200M_k40_t4_T8 (tests/golden/make_golden.py --huge): 50x of a 200 Mbp genome in 15 kbp reads with
    0.2 % substitutions -- 666,666 reads, 10.0 G bases, 9.97 G k-mer instances, 11.6 GB of super-mer records,
    18 GB of weighted k-mers, 200.0 M table entries -- counted by the REFERENCE FastK in the build container; only
    its digests travel, the reads are regenerated by fk_synth_reads.  Every buffer of the resident path lies
    beyond 2^32 bytes here.  Against the reference's .hist bytes, .ktab canonical stream, entry count and stub:
      (a) the setting bench.py times: reads resident, 48 minimizer buckets, 3 split passes with entry replay;
      (b) the same without replay (every pass recomputes the minimizers);
      (c) 4 buckets in one pass: 2.9 GB of super-mers and 4.5 GB of weighted k-mers per bucket;
      (d) value_device's route: reads in host memory pushed in 1 GB blocks with a budget that spills;
      (e) the C driver on the FASTA file, on one GPU and as two RCCL ranks (-G2)."""
    import ctypes as C, json, os, subprocess
    case = json.load(open(os.path.join(util.GOLDEN, "hifi50x200M_k40_t4_T8.json")))
    s, exp = case["synth"], case["expected"]
    k, cutoff, T, L, nreads = case["k"], case["cutoff"], case["T"], s["read_len"], s["nreads"]
    nbytes = nreads * (L + 1)
    assert nbytes > (1 << 32)

    def check(res, what):
        assert res.ninst == nreads * (L - k + 1), what
        util.check_against_golden(case, res.hist, res.max_inst, res.table)

    with fastk_amd.Context(kmer=k) as gen:
        buf = gen.alloc(nbytes + 64)
        piece = 1 << 16
        for first in range(0, nreads, piece):
            n = min(piece, nreads - first)
            gen._ck(gen.L.fk_synth_reads(gen.h, s["seed"], s["genome_len"], L, s["err_ppm"], first, n, buf.ptr + first * (L + 1)))
        gen._ck(gen.L.fk_synchronize(gen.h))
        sample = buf.download(8 << 20)
        for what, nb, passes, replay in (("a", 48, 3, 1), ("b", 48, 3, 0), ("c", 4, 1, 1)):
            with fastk_amd.Context(kmer=k, table_cutoff=cutoff, nthreads=T, nbuckets=nb, split_passes=passes) as ctx:
                ctx.set_bucket_weights(ctx.bucket_census(sample))
                ctx.debug_set("split_replay", replay)
                res = ctx.count_device_reads(buf.ptr, nbytes, fetch_table=True)
                assert res.split_passes == passes and res.buckets_counted == nb, what
                assert res.replay_passes == ((passes - 1) if replay else 0), what
                check(res, what)
                del res
        # (d) the reads go to pinned host memory and come back block by block
        lib = gen.L
        host = C.c_void_p()
        gen._ck(lib.fk_host_alloc(nbytes + 64, C.byref(host)))
        gen._ck(lib.fk_copy_to_host(gen.h, host.value, buf.ptr, nbytes))
        buf.free()
    try:
        with fastk_amd.Context(kmer=k, table_cutoff=cutoff, nthreads=T, nbuckets=16, hbm_budget=8 << 30) as ctx:
            ctx.set_bucket_weights(ctx.bucket_census(sample))
            per = 65536                                         # reads per block: 1 GB
            boff = (np.arange(per + 1, dtype=np.int64) * (L + 1)).astype(np.int32)
            for first in range(0, nreads, per):
                n = min(per, nreads - first)
                ctx._ck(lib.fk_push_block(ctx.h, host.value + first * (L + 1), boff.ctypes.data, n, 0, 0))
            res = ctx.finish()
            assert ctx.debug_get("spilled_bytes") > (1 << 30)
            check(res, "d")
            del res
        # (e) the same reads as a FASTA file, one line per read, through the C driver
        path = str(tmp_path / "x.fasta")
        view = np.ctypeslib.as_array(C.cast(host.value, C.POINTER(C.c_uint8)), shape=(nbytes,))
        with open(path, "wb") as f:
            for first in range(0, nreads, 20000):
                n = min(20000, nreads - first)
                mat = np.empty((n, 3 + L + 1), dtype=np.uint8)
                mat[:, 0:3] = np.frombuffer(b">r\n", dtype=np.uint8)
                mat[:, 3:3 + L] = view[first * (L + 1):(first + n) * (L + 1)].reshape(n, L + 1)[:, :L]
                mat[:, 3 + L] = ord("\n")
                mat.tofile(f)
        del view
    finally:
        lib.fk_host_free(host)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = util.driver_exe("FastK_amd")
    args = ["-k%d" % k, "-t%d" % cutoff, "-T%d" % T]
    for what, extra, env in (("one GPU", ["-M64"], None), ("-G2", ["-G2"], dict(os.environ, FK_RANKS_SHARE_GPU="1"))):
        d = tmp_path / what.strip("-").replace(" ", "")
        d.mkdir()
        p = subprocess.run([exe] + args + extra + ["-N" + str(d / "x"), path], env=env, capture_output=True, text=True, timeout=1500)
        if p.returncode != 0 and env is not None and any(m in p.stdout + p.stderr for m in _RCCL_RIG_ERRORS) \
                and os.environ.get("FK_REQUIRE_RANKS") != "1":
            pytest.skip("RCCL would not bring up two ranks on one GPU here: " + p.stderr[-300:])
        assert p.returncode == 0, (what, (p.stdout + p.stderr)[-3000:])
        assert util.sha_file(d / "x.hist") == exp["hist_sha256"], what
        t = orc.read_ktab(str(d / "x"))
        assert (t["kmer"], t["nparts"], t["minval"], t["ibytes"], t["nels"]) == (k, T, cutoff, exp["ktab"]["ibytes"], exp["ktab"]["nels"]), what
        assert t["stream_sha256"] == exp["ktab"]["stream_sha256"], what
        del t
        for f in os.listdir(d):
            os.remove(d / f)


@pytest.mark.parametrize("name", ["edge_k40_t1_T4", "synth_hifi_k40_t4_T8", "synth_illumina_k51_t1_T4"])
def test_reference_readers_accept_our_files(name, tmp_path):
    """SURVEY 8(f)-1 reader conformance: the reference-built Histex, Tabex and Logex (oracle/_ref) print
    for OUR files (default pipeline: our own part boundaries) exactly what they print for the files of
    the reference FastK run in place on the same input."""
    import os, subprocess
    tools = {t: os.path.join(orc.REF_DIR, t) for t in ("FastK", "Histex", "Tabex", "Logex")}
    if not all(os.path.exists(p) for p in tools.values()):
        util.no_reference("oracle/_ref tools not built (needs the reference sources at build time)")
    case, bases, boff = util.load_case(name)
    k, T, cutoff = case["k"], case["T"], case["cutoff"]
    ours, theirs = tmp_path / "ours", tmp_path / "theirs"
    ours.mkdir(); theirs.mkdir()
    orc.write_fasta(str(theirs / "x.fasta"), bases, boff)
    orc.run_ref_fastk(str(theirs / "x.fasta"), k, cutoff, T, str(theirs))
    with fastk_amd.Context(kmer=k, table_cutoff=cutoff, nthreads=T) as ctx:
        ctx.push_block(bases, boff.astype(np.int32))
        res = ctx.finish()
        ctx.write_hist(res, str(ours / "x.hist"))
        ctx.write_ktab(res, str(ours), "x")

    def run(tool, *args):
        outs = []
        for d in (ours, theirs):
            p = subprocess.run([tools[tool]] + list(args), cwd=str(d), capture_output=True, text=True)
            assert p.returncode == 0, (tool, args, p.stderr[-500:])
            outs.append(p.stdout + p.stderr)
        assert outs[0] == outs[1], (tool, args)
        return outs[0]

    assert len(run("Histex", "x")) > 0                       # the default range
    run("Histex", "-h1:20", "x")
    run("Histex", "-A", "-k", "-h1:32767", "x")              # k-mer instances, every count, ASCII
    run("Histex", "-G", "x")                                 # the GeneScope form
    assert "OK" in run("Tabex", "-C", "x")
    assert len(run("Tabex", "-A", "-t%d" % max(cutoff, 2), "x")) > 0      # the whole table as text
    run("Tabex", "x", "100-200")
    # Logex: a one-table expression (k-mers with count >= 2), table and histogram output
    for d in (ours, theirs):
        p = subprocess.run([tools["Logex"], "-T%d" % T, "-h", "sel = A[2-]", "x"], cwd=str(d), capture_output=True, text=True)
        assert p.returncode == 0, p.stderr[-500:]
    for f in ("sel.hist",):
        assert util.sha_file(ours / f) == util.sha_file(theirs / f), f
    a, b = orc.read_ktab(str(ours / "sel")), orc.read_ktab(str(theirs / "sel"))
    assert a["stream_sha256"] == b["stream_sha256"] and a["nels"] == b["nels"]


# ------------------------------------------------------------------------------ full size
def test_full_size_properties_configs1():
    """BASELINE configs[1] at full size (33.3 M reads of 150 bp, 3.7 G k-mer instances): no oracle
    finishes this in seconds, so the run is checked through properties that do not depend on size:
    conservation (every k-mer instance is counted exactly once: sum_c c*hist[c] + max_inst ==
    instances), distinct k-mers == sum(hist) == table entries at cutoff 1, strictly increasing
    table, table counts reproduce the histogram, and the bucket-streamed run (4 buckets, other code
    path: planned split, per-bucket aggregation, union sort) gives the identical histogram and table."""
    import json, os
    L, glen, k = 150, 100_000_000, 40
    nreads = int(50 * glen / L)
    nbytes = nreads * (L + 1)
    inst = nreads * (L - k + 1)
    results = []
    for nb in (1, 4):
        with fastk_amd.Context(kmer=k, table_cutoff=1, nbuckets=nb) as ctx:
            buf, n = ctx.synth_reads(20251001, glen, L, 1000, 0, nreads)
            assert n == nbytes
            res = ctx.count_device_reads(buf.ptr, nbytes, fetch_table=True)
            buf.free()
        h = res.hist.astype(np.int64)
        assert res.ninst == inst
        assert int((h[1:0x7fff] * np.arange(1, 0x7fff)).sum()) + int(res.max_inst) == inst
        assert res.ndistinct == int(h.sum()) == res.ntable == len(res.table)
        results.append(res)
    a, b = results
    assert np.array_equal(a.hist, b.hist) and a.max_inst == b.max_inst
    assert np.array_equal(a.table, b.table)
    # ... and against reference FastK itself on these very reads (tests/golden/make_golden.py --only=configs1_k40_t1_T4:
    # 33.3 M reads through oracle/_ref/FastK -k40 -t1 -T4): histogram bins, .hist file bytes, entry count, index width
    # and the canonical .ktab stream (count.c:1893-1910, table.c:485-498)
    case = json.load(open(os.path.join(util.GOLDEN, "configs1_k40_t1_T4.json")))
    assert case["synth"] == dict(seed=20251001, genome_len=glen, read_len=L, err_ppm=1000, nreads=nreads)
    util.check_against_golden(case, a.hist, a.max_inst, a.table)
    t = a.table
    # strictly increasing keys: compare as big-endian integers, 8 + 2 bytes
    hi = t[:, :8].copy().view(">u8").ravel()
    lo = t[:, 8:10].copy().view(">u2").ravel()
    assert np.all((hi[1:] > hi[:-1]) | ((hi[1:] == hi[:-1]) & (lo[1:] > lo[:-1])))
    cnt = t[:, 10:12].copy().view("<u2").ravel()
    assert np.array_equal(np.bincount(cnt, minlength=0x8000)[1:], a.hist[1:])


def test_full_size_properties_configs2():
    """BASELINE configs[2] at full size on one GPU: 50x of a 3 Gbp genome in 15 kbp reads with 0.2 %
    substitutions, k=40 -t4 (10 M reads, 150 G bases, 149.61 G k-mer instances, ~8.6 G super-mers,
    ~22 G weighted k-mers, 3.0 G table entries).  The 150 GB of reads stay resident; the run takes 3
    split passes over them (one that computes the minimizers and records the other groups' entries, two that
    replay them) and 48 minimizer buckets one after the other (bench.py's setting), then again with 40 buckets
    and three full passes (other group boundaries, other bucket contents, no replay).  Checked: instance
    count, conservation (sum c*hist[c] + max_inst == instances), sum(hist) == distinct k-mers, table
    entries == sum(hist[4:]), strictly increasing table, table counts reproduce the histogram from the
    cutoff up, and both settings give the identical histogram and table.  A summary goes to
    gpurun_out/full_size_configs2.json."""
    import json, os, time
    L, glen, k, cutoff = 15000, 3_000_000_000, 40, 4
    nreads = int(50 * glen / L)
    nbytes = nreads * (L + 1)
    inst = nreads * (L - k + 1)
    summary = dict(reads=nreads, bases=nreads * L, kmer_instances=inst, runs=[])
    tables = []
    with fastk_amd.Context(kmer=k) as gen:
        buf = gen.alloc(nbytes + 64)
        piece = 1 << 20                                    # reads per generator call
        for first in range(0, nreads, piece):
            n = min(piece, nreads - first)
            gen._ck(gen.L.fk_synth_reads(gen.h, 20251001, glen, L, 2000, first, n, buf.ptr + first * (L + 1)))
        gen._ck(gen.L.fk_synchronize(gen.h))
        sample = buf.download(8 << 20)
        for nb, passes in ((48, 3), (40, 3)):
            with fastk_amd.Context(kmer=k, table_cutoff=cutoff, nbuckets=nb, split_passes=passes) as ctx:
                ctx.set_bucket_weights(ctx.bucket_census(sample))
                if nb == 40:
                    ctx.debug_set("split_replay", 0)          # this run recomputes the minimizers in every pass
                t0 = time.perf_counter()
                res = ctx.count_device_reads(buf.ptr, nbytes, fetch_table=True)
                dt = time.perf_counter() - t0
            h = res.hist.astype(np.int64)
            assert res.ninst == inst
            assert int((h[1:0x7fff] * np.arange(1, 0x7fff)).sum()) + int(res.max_inst) == inst
            assert res.ndistinct == int(h.sum())
            assert res.ntable == int(h[cutoff:].sum()) == len(res.table)
            assert res.split_passes == passes and res.buckets_counted == nb
            assert res.replay_passes == (passes - 1 if nb == 48 else 0)
            summary["runs"].append(dict(buckets=nb, split_passes=passes, seconds_first_run_with_table_fetch=round(dt, 2),
                                        supermers=res.nsuper, weighted_kmers=res.nweighted, distinct_kmers=res.ndistinct,
                                        table_entries=res.ntable, device_ms=res.ms))
            tables.append((res.hist, res.max_inst, res.table))
            del res
        buf.free()
    (ha, ma, ta), (hb, mb, tb) = tables
    assert np.array_equal(ha, hb) and ma == mb
    step = 1 << 27                                         # entries per slice of the 36 GB tables
    cnt_hist = np.zeros(0x8000, dtype=np.int64)
    prev_hi, prev_lo = None, None
    for o in range(0, len(ta), step):
        a, b = ta[o:o + step], tb[o:o + step]
        assert np.array_equal(a, b), "tables differ in entries %d.." % o
        hi = np.ascontiguousarray(a[:, :8]).view(">u8").ravel()
        lo = np.ascontiguousarray(a[:, 8:10]).view(">u2").ravel()
        assert np.all((hi[1:] > hi[:-1]) | ((hi[1:] == hi[:-1]) & (lo[1:] > lo[:-1]))), "table not increasing near %d" % o
        if prev_hi is not None:
            assert (hi[0] > prev_hi) or (hi[0] == prev_hi and lo[0] > prev_lo)
        prev_hi, prev_lo = hi[-1], lo[-1]
        cnt_hist += np.bincount(np.ascontiguousarray(a[:, 10:12]).view("<u2").ravel(), minlength=0x8000)
    assert np.array_equal(cnt_hist[cutoff:], ha[cutoff:]) and cnt_hist[:cutoff].sum() == 0
    # ... and against the REFERENCE: tests/golden/configs2_k40_t4.json holds what oracle/_ref/FastK -k40 -t4 left for
    # these 150 G bases (tools/cpu_baseline_full.py --golden, run on a GPU box's host: 272 s) -- the sha256 of the .hist
    # file, the number of table entries and the sha256 of the .ktab canonical stream (prefix index + the 9-byte
    # payloads of all parts, 27 GB).  VERDICT r5: until round 5 this test compared two runs of our own.
    import hashlib
    gold = json.load(open(os.path.join(util.GOLDEN, "configs2_k40_t4.json")))
    exp = gold["expected"]
    assert gold["synth"] == dict(seed=20251001, genome_len=glen, read_len=L, err_ppm=2000, nreads=nreads)
    assert gold["k"] == k and gold["cutoff"] == cutoff and exp["kmer_instances"] == inst
    raw = orc.hist_file_bytes(k, ha, ma)
    assert len(raw) == exp["hist_len"]
    assert hashlib.sha256(raw).hexdigest() == exp["hist_sha256"], ".hist bytes differ from reference FastK's at configs[2]"
    assert len(ta) == exp["ktab"]["nels"], "table entries differ from reference FastK's at configs[2]"
    ib = orc.idx_bytes(k, len(ta))
    assert ib == exp["ktab"]["ibytes"] == 3
    pre_cnt = np.zeros(1 << 24, dtype=np.int64)
    for o in range(0, len(ta), step):
        a = ta[o:o + step]
        pre = (a[:, 0].astype(np.int64) << 16) | (a[:, 1].astype(np.int64) << 8) | a[:, 2].astype(np.int64)
        pre_cnt += np.bincount(pre, minlength=1 << 24)
    hs = hashlib.sha256()
    hs.update(np.cumsum(pre_cnt).astype(np.int64).tobytes())
    for o in range(0, len(ta), 1 << 24):
        hs.update(np.ascontiguousarray(ta[o:o + (1 << 24), ib:]).tobytes())
    assert hs.hexdigest() == exp["ktab"]["stream_sha256"], ".ktab canonical stream differs from reference FastK's at configs[2]"
    summary["reference"] = dict(golden="tests/golden/configs2_k40_t4.json", hist_sha256=exp["hist_sha256"],
                                ktab_stream_sha256=exp["ktab"]["stream_sha256"], table_entries=exp["ktab"]["nels"], equal=True)
    summary["checks"] = "instances, conservation, sum(hist) = distinct, ntable = sum(hist[4:]), table strictly increasing, " \
                        "table counts = histogram, identical histogram and table for (48 buckets, 3 passes with entry replay) and (40 buckets, 3 full passes), " \
                        ".hist sha256 / table entries / .ktab stream sha256 equal to reference FastK's (golden fixture)"
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "full_size_configs2.json"), "w") as f:
            json.dump(summary, f, indent=1)


def test_configs4_scaled_slice_k51_profiles_with_spill(tmp_path):
    """BASELINE configs[4] (100x of a 32 Gbp genome, k=51 -t1 -p, HBM-spill stress) at a STATED SCALE of
    1/1000: 100x coverage of a 32 Mbp genome in 150 bp reads (21.3 M reads, 3.2 G bases, 2.13 G k-mer
    instances), through the C driver on one GPU with -M2: 2 GB budget, so the reads are split chunk by chunk,
    ~3/4 of the 28-byte super-mer records go through pinned host memory and come back bucket by bucket, and
    the profiles take the second pass over the input.  Checked: conservation, sum(hist) = table entries at
    -t1, and .hist / .ktab / every .prof file byte-identical to the all-resident run of the same command."""
    import json, os, subprocess
    L, glen, k = 150, 32_000_000, 51
    nreads = int(100 * glen / L)
    bases, boff = orc.synth_block(4051, glen, L, 1000, 0, nreads)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = util.driver_exe("FastK_amd")
    path = str(tmp_path / "x.fastq")
    util.write_fastx(path, bases, boff, True)
    del bases
    outs = {}
    for name, mem in (("res", []), ("mem", ["-M2"])):
        d = tmp_path / name
        d.mkdir()
        p = subprocess.run([exe, "-k%d" % k, "-t1", "-T4", "-p", "-v"] + mem + ["-N" + str(d / "x"), path], capture_output=True,
                           text=True)
        assert p.returncode == 0, p.stderr[-2000:]
        outs[name] = {f: util.sha_file(d / f) for f in sorted(os.listdir(d))}
        if mem:
            assert "minimizer bucket" in p.stderr                     # the budget really chunked the run
    assert outs["res"] == outs["mem"] and len(outs["res"]) == 2 + 4 + 1 + 8
    # ... and the REFERENCE's results for these very reads (tests/golden/make_golden.py --only=configs4slice_k51_t1_T4_p:
    # oracle/_ref/FastK -k51 -t1 -T4 -p on the 21.3 M reads): .hist bytes, the canonical .ktab stream, and the digest of
    # every read's DECODED profile (the reference's bytes depend on its super-mer cuts; merge.c:880-930)
    gold = json.load(open(os.path.join(util.GOLDEN, "configs4slice_k51_t1_T4_p.json")))
    assert gold["synth"] == dict(seed=4051, genome_len=glen, read_len=L, err_ppm=1000, nreads=nreads) and gold["k"] == k

    def against_reference(dd, what):
        exp = gold["expected"]
        assert util.sha_file(dd / "x.hist") == exp["hist_sha256"], what + ": .hist is not the reference's"
        tt = orc.read_ktab(str(dd / "x"))
        assert (tt["nels"], tt["ibytes"]) == (exp["ktab"]["nels"], exp["ktab"]["ibytes"]), what
        assert tt["stream_sha256"] == exp["ktab"]["stream_sha256"], what + ": the .ktab stream is not the reference's"
        del tt
        nr, _, npos, dig = orc.profiles_digest_files(str(dd), "x")
        assert (nr, npos) == (exp["prof"]["nreads"], exp["prof"]["kmer_positions"]), what
        assert dig == exp["prof"]["decoded_sha256"], what + ": the decoded profiles are not the reference's"

    against_reference(tmp_path / "mem", "one GPU, -M2")
    # ... and as BASELINE states it, on several GPUs: -G2 -p -M2 on the one-GPU rig (two ranks share the device, each
    # with a 2 GB budget for its stripe).  The ranks cut their parts elsewhere, so the comparison is on contents:
    # .hist bytes, the canonical .ktab stream, and every read's profile bytes in file order.
    d = tmp_path / "g2"
    d.mkdir()
    p = subprocess.run([exe, "-k%d" % k, "-t1", "-T4", "-p", "-v", "-M2", "-G2", "-N" + str(d / "x"), path], capture_output=True, text=True,
                       env=dict(os.environ, FK_RANKS_SHARE_GPU="1"), timeout=1800)
    if p.returncode != 0 and any(m in p.stdout + p.stderr for m in _RCCL_RIG_ERRORS):
        if os.environ.get("FK_REQUIRE_RANKS") == "1":
            pytest.fail("RCCL would not bring up 2 ranks on one GPU: %s" % p.stderr[-500:])
    else:
        assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
        assert util.sha_file(d / "x.hist") == outs["mem"]["x.hist"]
        assert orc.read_ktab(str(d / "x"))["stream_sha256"] == orc.read_ktab(str(tmp_path / "mem" / "x"))["stream_sha256"]

        def prof_bytes(dd):                       # (offsets per read, data) of the whole set, parts in order
            import struct
            offs, data, base = [np.zeros(1, dtype=np.int64)], [], 0
            nparts = struct.unpack("<ii", open(dd / "x.prof", "rb").read(8))[1]
            for tt in range(1, nparts + 1):
                px = open(dd / (".x.pidx.%d" % tt), "rb").read()
                first, n = struct.unpack("<qq", px[4:20])
                assert first == sum(len(o) for o in offs) - 1
                o = np.frombuffer(px[20:20 + 8 * n], dtype=np.int64)
                offs.append(o + base)
                dat = np.fromfile(dd / (".x.prof.%d" % tt), dtype=np.uint8)
                base += len(dat)
                data.append(dat)
            return np.concatenate(offs), np.concatenate(data)
        oa, da = prof_bytes(tmp_path / "mem")
        ob, db = prof_bytes(d)
        assert len(oa) == nreads + 1 and np.array_equal(oa, ob) and np.array_equal(da, db), "the ranks' profiles differ from the one-GPU run's"
        del oa, da, ob, db
        against_reference(d, "-G2 -M2")
    h = orc.read_hist(str(tmp_path / "mem" / "x.hist"))
    hist = np.asarray(h["hist"], dtype=np.int64)
    cnt = np.arange(h["low"], h["low"] + len(hist))
    inst = nreads * (L - k + 1)
    assert int((hist[:-1] * cnt[:-1]).sum()) + int(h["ihigh"]) == inst
    t = orc.read_ktab(str(tmp_path / "mem" / "x"))
    assert t["nels"] == int(hist.sum())


# ------------------------------------------------------------------------------ profiles (-p)

def _check_profiles(k, bases, boff, data, offs, table=None):
    if table is None:
        table = orc.fastk(k, bases, boff, cutoff=1).table
    exp = orc.profile_counts(k, bases, boff, table)
    assert len(offs) == len(exp) + 1 and offs[0] == 0 and offs[-1] == len(data)
    raw = data.tobytes()
    for i, x in enumerate(exp):
        got = raw[offs[i]:offs[i + 1]]
        assert got == orc.profile_encode(x), "read %d" % i
    return exp


@pytest.mark.parametrize("nb", [1, 4])
@pytest.mark.parametrize("name", ["synth_tiny_k40_t1_T2", "edge_k40_t1_T4", "edge_k51_t1_T4",
                                  "synth_illumina_k40_t1_T4"])
def test_profiles_match_oracle_and_reference(name, nb, tmp_path):
    """Profiles from the device (table lookup per position + codec) are the oracle's canonical streams,
    decode to what the reference's own -p files decode to, and the reference's Profex lists the files
    written by fk_write_prof exactly like its own."""
    import os
    import subprocess
    case, bases, boff = util.load_case(name)
    k = case["k"]
    d = str(tmp_path)
    with fastk_amd.Context(kmer=k, table_cutoff=1, nthreads=case["T"], nbuckets=nb) as ctx:
        nreads = len(boff) - 1
        step = max(1, nreads // 3)
        for s in range(0, nreads, step):
            e = min(nreads, s + step)
            ctx.push_block(bases[boff[s]:boff[e]], (boff[s:e + 1] - boff[s]).astype(np.int32))
        res = ctx.finish()
        data, offs = ctx.make_profiles(outdir=d, root="x", nparts=case["T"])
    exp = _check_profiles(k, bases, boff, data, offs, res.table)
    raw = data.tobytes()
    mine = [orc.profile_decode(raw[offs[i]:offs[i + 1]]) for i in range(len(offs) - 1)]
    assert orc.profiles_digest(mine) == case["expected"]["prof"]["decoded_sha256"]   # the reference's
    if not orc.have_ref() or nb != 1:
        return
    rd = os.path.join(d, "ref")
    os.mkdir(rd)
    path = os.path.join(rd, "x.fastq")
    orc.write_fastq(path, bases, boff)
    orc.run_ref_fastk(path, k, 1, case["T"], rd, extra=("-p",))
    kk, enc = orc.read_profiles(rd, "x")
    assert kk == k and len(enc) == len(exp)
    for e, x in zip(enc, exp):
        assert orc.profile_decode(e) == x.tolist()
    profex = os.path.join(orc.REF_DIR, "Profex")
    if os.path.exists(profex):
        a = subprocess.run([profex, os.path.join(d, "x"), "1-#"], check=True, capture_output=True).stdout
        b = subprocess.run([profex, os.path.join(rd, "x"), "1-#"], check=True, capture_output=True).stdout
        assert a == b and len(a) > 0


@pytest.mark.parametrize("k", [12, 16, 17, 31, 32, 33, 48, 63, 64])
def test_profiles_across_k(k):
    rng = np.random.default_rng(100 + k)
    genome = rng.integers(0, 4, size=20000)
    reads = []
    for _ in range(800):
        s0 = int(rng.integers(0, len(genome) - 400))
        r = genome[s0:s0 + int(rng.integers(max(1, k - 3), 400))].copy()
        if rng.random() < 0.5:
            r = (3 - r)[::-1]
        if rng.random() < 0.3 and len(r) > 5:
            r[int(rng.integers(0, len(r)))] = 4
        reads.append("".join("acgtn"[x] for x in r))
    reads += ["", "a" * 300, "acgt" * 60, "n" * 50, "ac" * 90] * 3
    reads += ["a" * 9000]                                           # a run > 63 many times over
    bases, boff = orc.block_from_reads(reads)
    with fastk_amd.Context(kmer=k, table_cutoff=1) as ctx:
        ctx.push_block(bases, boff.astype(np.int32))
        ctx.finish()
        data, offs = ctx.make_profiles()
    _check_profiles(k, bases, boff, data, offs)


def test_profiles_saturated_counts_and_call_order():
    k = 40
    rng = np.random.default_rng(9)
    unit = "".join("acgt"[x] for x in rng.integers(0, 4, size=60))
    reads = [unit] * 40000 + ["".join("acgt"[x] for x in rng.integers(0, 4, size=100)) for _ in range(50)]
    bases, boff = orc.block_from_reads(reads)
    with fastk_amd.Context(kmer=k, table_cutoff=1) as ctx:
        with pytest.raises(fastk_amd.FastKError):
            ctx.make_profiles()                                     # nothing counted yet
        ctx.push_block(bases, boff.astype(np.int32))
        ctx.finish()
        data, offs = ctx.make_profiles()
    exp = _check_profiles(k, bases, boff, data, offs)
    assert int(exp[0][0]) == 32767                                  # capped like the table's counts
    with fastk_amd.Context(kmer=k, table_cutoff=2) as ctx:
        ctx.push_block(bases, boff.astype(np.int32))
        ctx.finish()
        with pytest.raises(fastk_amd.FastKError):
            ctx.make_profiles()                                     # needs the cutoff-1 table


@pytest.mark.parametrize("name,fmt", [("edge_k40_t4_T1", "fasta"), ("synth_hifi_k40_t4_T8", "fastq"),
                                      ("edge_k51_t1_T4", "fastq")])
def test_cli_profiles_option(name, fmt, tmp_path):
    """FastK_amd -p -t<n>: profiles decode to the reference's, the .hist and the cutoff-n table are the
    usual ones (the engine counts with cutoff 1 for the look-ups and filters on the way out)."""
    import hashlib, os, subprocess
    case, bases, boff = util.load_case(name)
    k = case["k"]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = util.driver_exe("FastK_amd")
    path = str(tmp_path / ("reads." + fmt))
    (orc.write_fasta if fmt == "fasta" else orc.write_fastq)(path, bases, boff)
    subprocess.run([exe, "-k%d" % k, "-t%d" % case["cutoff"], "-T%d" % case["T"], "-p", "-v", path],
                   check=True, cwd=str(tmp_path))
    exp = case["expected"]
    assert hashlib.sha256(open(tmp_path / "reads.hist", "rb").read()).hexdigest() == exp["hist_sha256"]
    t = orc.read_ktab(str(tmp_path / "reads"))
    assert t["stream_sha256"] == exp["ktab"]["stream_sha256"] and t["minval"] == case["cutoff"]
    kk, enc = orc.read_profiles(str(tmp_path), "reads")
    table = orc.fastk(k, bases, boff, cutoff=1).table
    want = orc.profile_counts(k, bases, boff, table)
    assert kk == k and len(enc) == len(want)
    for e, x in zip(enc, want):
        assert e == orc.profile_encode(x)
    if orc.have_ref():
        rd = tmp_path / "ref"
        rd.mkdir()
        rp = str(rd / ("reads." + fmt))
        (orc.write_fasta if fmt == "fasta" else orc.write_fastq)(rp, bases, boff)
        orc.run_ref_fastk(rp, k, case["cutoff"], case["T"], str(rd), extra=("-p",))
        k2, renc = orc.read_profiles(str(rd), "reads")
        assert [orc.profile_decode(e) for e in renc] == [x.tolist() for x in want]


def test_relative_profiles_against_another_table(tmp_path):
    """-p:<table> (README.md:112-120): counts come from another data set's table, 0 for k-mers it does
    not hold; API (fk_set_table + fk_make_profiles, no counting run) and CLI against the reference."""
    import os, subprocess
    k, T = 40, 3
    _, basesB, boffB = util.load_case("synth_illumina_k40_t1_T4")        # table source
    rng = np.random.default_rng(77)
    nB = len(boffB) - 1
    readsA = []
    for i in rng.integers(0, nB, size=600):                              # reads of B, some mutated
        r = basesB[boffB[i]:boffB[i + 1] - 1].copy()
        if rng.random() < 0.5:
            r[int(rng.integers(0, len(r)))] = ord("acgt"[int(rng.integers(0, 4))])
        if rng.random() < 0.1:
            r[int(rng.integers(0, len(r)))] = ord("N")
        readsA.append(r.tobytes().decode())
    readsA += ["".join("acgt"[x] for x in rng.integers(0, 4, size=200)) for _ in range(50)]   # unrelated
    readsA += ["acgt" * 5, ""]
    basesA, boffA = orc.block_from_reads(readsA)
    tabB = orc.fastk(k, basesB, boffB, cutoff=2).table                   # a cutoff-2 table: absent k-mers -> 0
    want = orc.profile_counts(k, basesA, boffA, tabB)
    assert any((w == 0).any() and (w > 0).any() for w in want)
    rng.shuffle(tabB)                                                    # any order is accepted
    with fastk_amd.Context(kmer=k, table_cutoff=0) as ctx:
        ctx.push_block(basesA, boffA.astype(np.int32))
        ctx.set_table(tabB)
        data, offs = ctx.make_profiles()
    raw = data.tobytes()
    for i, x in enumerate(want):
        assert raw[offs[i]:offs[i + 1]] == orc.profile_encode(x), i

    # CLI + the reference on files
    d = str(tmp_path)
    pb, pa = os.path.join(d, "B.fasta"), os.path.join(d, "A.fasta")
    orc.write_fasta(pb, basesB, boffB)
    orc.write_fasta(pa, basesA, boffA)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = util.driver_exe("FastK_amd")
    subprocess.run([exe, "-k%d" % k, "-t2", "-T2", pb], check=True, cwd=d)
    subprocess.run([exe, "-k%d" % k, "-T%d" % T, "-p:B", "-v", pa], check=True, cwd=d)
    assert not os.path.exists(os.path.join(d, "A.hist"))                 # only profiles are produced
    kk, enc = orc.read_profiles(d, "A")
    assert kk == k and enc == [orc.profile_encode(x) for x in want]
    if orc.have_ref():
        rd = os.path.join(d, "ref")
        os.mkdir(rd)
        for f in ("A.fasta", "B.fasta"):
            os.link(os.path.join(d, f), os.path.join(rd, f))
        orc.run_ref_fastk(os.path.join(rd, "B.fasta"), k, 2, 2, rd)
        orc.run_ref_fastk(os.path.join(rd, "A.fasta"), k, 0, T, rd, extra=("-p:B",))
        k2, renc = orc.read_profiles(rd, "A")
        assert [orc.profile_decode(e) for e in renc] == [x.tolist() for x in want]


def test_profiles_after_chunked_run_piece_by_piece():
    """A chunked (HBM-budgeted, bucketed) run drops the reads; its table still serves look-ups for
    reads handed over again in pieces: concatenated, the profiles are the data set's."""
    case, bases, boff = util.load_case("synth_illumina_k40_t1_T4")
    k = case["k"]
    with fastk_amd.Context(kmer=k, table_cutoff=1, nbuckets=3, hbm_budget=1 << 30) as ctx:
        ctx.debug_set("chunk_bytes", len(bases) // 4)
        ctx.push_block(bases, boff.astype(np.int32))
        res = ctx.finish()
        util.check_against_golden(case, res.hist, res.max_inst, res.table)
        with pytest.raises(fastk_amd.FastKError):
            ctx.make_profiles()
        got = []
        nreads = len(boff) - 1
        for lo in range(0, nreads, 7001):
            hi = min(nreads, lo + 7001)
            piece = np.ascontiguousarray(bases[boff[lo]:boff[hi]])
            buf = ctx.alloc(len(piece) + 64).upload(piece)
            data, offs = ctx.make_profiles(buf.ptr, len(piece))
            raw = data.tobytes()
            got += [orc.profile_decode(raw[offs[i]:offs[i + 1]]) for i in range(len(offs) - 1)]
            buf.free()
    assert len(got) == nreads
    assert orc.profiles_digest(got) == case["expected"]["prof"]["decoded_sha256"]


def test_profiles_follow_input_thread_order(tmp_path):
    """Blocks of three input threads pushed interleaved (what io.c's threads do): profiles come out in
    data-set order (thread 0's reads, then thread 1's, ...) and the part files are the threads' ranges."""
    import struct
    case, bases, boff = util.load_case("synth_illumina_k40_t1_T4")
    k = case["k"]
    nreads = len(boff) - 1
    cuts = [0, nreads // 5, nreads // 2, nreads]                  # uneven thread ranges
    cur = cuts[:3]
    with fastk_amd.Context(kmer=k, table_cutoff=1, nthreads=3) as ctx:
        step = 611
        while any(cur[t] < cuts[t + 1] for t in range(3)):
            for t in (2, 0, 1):
                lo, hi = cur[t], min(cuts[t + 1], cur[t] + step)
                if lo < hi:
                    ctx.push_block(bases[boff[lo]:boff[hi]], (boff[lo:hi + 1] - boff[lo]).astype(np.int32), tid=t)
                    cur[t] = hi
        ctx.finish()
        data, offs = ctx.make_profiles(outdir=str(tmp_path), root="x", nparts=3)
    raw = data.tobytes()
    got = [orc.profile_decode(raw[offs[i]:offs[i + 1]]) for i in range(nreads)]
    assert orc.profiles_digest(got) == case["expected"]["prof"]["decoded_sha256"]
    for t in range(3):
        px = open(tmp_path / (".x.pidx.%d" % (t + 1)), "rb").read()
        b, n = struct.unpack("<qq", px[4:20])
        assert (b, n) == (cuts[t], cuts[t + 1] - cuts[t])


def test_randomised_differential_run():
    """tests/fuzz_parity.py: random k, read mixes, buckets, chunking / host spill, block sizes, thread
    ids, cut-offs and profiles against the oracle (1,650 configurations were run when it was written;
    a short slice stays in the suite)."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tests", "fuzz_parity.py"), "40", "7"],
                       capture_output=True, text=True, timeout=900, cwd=root)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "all 40 iterations equal to the oracle" in p.stdout


@pytest.mark.parametrize("name", ["synth_illumina_k40_t1_T4", "edge_k40_t1_T4", "edge_k51_t1_T4"])
def test_profiles_with_lookups_on_the_owning_rank(name):
    """shard.profiles_exchanged on a one-rank RCCL group: split with positions -> exchange -> count ->
    owner-side look-ups of the received records -> counts back -> scatter to positions -> codec.  Totals
    are the golden ones and the profiles decode to the reference's."""
    import os
    import torch
    import torch.distributed as dist
    from tests import shard_model as shard
    case, bases, boff = util.load_case(name)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29593")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        with fastk_amd.Context(kmer=case["k"], table_cutoff=1, nbuckets=1) as ctx:
            eng = shard.HipEngine(ctx, torch.device("cuda", 0))
            reads = torch.from_numpy(bases).cuda()
            tot, data, offs = shard.profiles_exchanged(eng, reads, fetch_table=True)
    finally:
        dist.destroy_process_group()
    util.check_against_golden(case, tot["hist"], tot["max_inst"], tot["local"]["result"].table)
    raw = data.tobytes()
    got = [orc.profile_decode(raw[offs[i]:offs[i + 1]]) for i in range(len(offs) - 1)]
    assert orc.profiles_digest(got) == case["expected"]["prof"]["decoded_sha256"]


@pytest.mark.parametrize("kind", ["sam", "bam"])
def test_cli_reads_sam_and_bam(kind, tmp_path):
    """FastK_amd on .sam / .bam input (io.c:1314-1495): secondary / supplementary records are skipped, SAM
    letters go through the reference's everything-is-a-base rule, BAM's 4-bit codes keep n & co. as
    breaks.  .hist bytes and the table stream equal the reference's on the same file; -p too."""
    import hashlib, os, subprocess
    case, bases, boff = util.load_case("edge_k40_t1_T4")
    reads = [bases[boff[i]:boff[i + 1] - 1].tobytes().decode() for i in range(len(boff) - 1)]
    reads = [r for r in reads if len(r) > 0][:6000] + ["acgtNRYKMacgt" * 8, "ac" * 70000]   # a read longer than a block piece
    rng = np.random.default_rng(3)
    flags = [int(rng.choice([0, 4, 16, 0x100, 0x800, 0x904])) for _ in reads]
    flags[-1] = 0
    k, T = 40, 3
    d = str(tmp_path)
    path = os.path.join(d, "x." + kind)
    (orc.write_sam if kind == "sam" else orc.write_bam)(path, reads, flags)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = util.driver_exe("FastK_amd")
    subprocess.run([exe, "-k%d" % k, "-t1", "-T%d" % T, "-p", "-Nmine", path], check=True, cwd=d)
    kept = [r for r, f in zip(reads, flags) if not (f & 0x900)]
    kept = [orc.sam_bases(r) for r in kept] if kind == "sam" else \
        ["".join(ch if ch.upper() in "=ACMGRSVTWYHKDBN" else "N" for ch in r).lower() for r in kept]
    b2, o2 = orc.block_from_reads(kept)
    exp = orc.fastk(k, b2, o2, cutoff=1)
    assert open(os.path.join(d, "mine.hist"), "rb").read() == orc.hist_file_bytes(k, exp.hist, exp.max_inst)
    t = orc.read_ktab(os.path.join(d, "mine"))
    assert t["stream_sha256"] == orc.table_stream_sha256(k, exp.table)
    kk, enc = orc.read_profiles(d, "mine")
    want = orc.profile_counts(k, b2, o2, exp.table)
    assert len(enc) == len(want) and all(e == orc.profile_encode(x) for e, x in zip(enc, want))
    if orc.have_ref():
        rd = os.path.join(d, "ref")
        os.mkdir(rd)
        rp = os.path.join(rd, "x." + kind)
        os.link(path, rp)
        orc.run_ref_fastk(rp, k, 1, T, rd)
        assert open(os.path.join(rd, "x.hist"), "rb").read() == open(os.path.join(d, "mine.hist"), "rb").read()
        assert orc.read_ktab(os.path.join(rd, "x"))["stream_sha256"] == t["stream_sha256"]


def test_cli_output_naming_follows_reference(tmp_path):
    """Output names and places (FastK.c:361-409): next to the FIRST input with its root by default, -N
    overrides both; several inputs make one data set.  Same file lists and same .hist as the reference."""
    import hashlib, os, subprocess
    case, bases, boff = util.load_case("synth_tiny_k40_t1_T2")
    nreads = len(boff) - 1
    half = nreads // 2
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = util.driver_exe("FastK_amd")
    ref = os.path.join(orc.REF_DIR, "FastK")
    if not os.path.exists(ref):
        util.no_reference("reference build not available")

    def layout(d):
        os.makedirs(os.path.join(d, "in", "sub"))
        os.makedirs(os.path.join(d, "out"))
        orc.write_fasta(os.path.join(d, "in", "a.part.fasta"), bases[:boff[half]], boff[:half + 1])
        orc.write_fasta(os.path.join(d, "in", "sub", "b.fa"), bases[boff[half]:], boff[half:] - boff[half], width=60)
        orc.write_fastq(os.path.join(d, "in", "sub", "c.fq"), bases[boff[half]:], boff[half:] - boff[half])

    def listing(d):
        out = []
        for dp, _, fs in os.walk(d):
            out += [os.path.relpath(os.path.join(dp, f), d) for f in fs]
        return sorted(out)

    # the extension may be left out, and foo.fastq finds foo.fastq.gz (Fetch_File, io.c:136-160)
    import gzip
    gdirs = []
    for tool in (exe, ref):
        d = str(tmp_path / ("g%d" % len(gdirs)))
        os.makedirs(d)
        layout(d)
        with open(os.path.join(d, "in", "sub", "c.fq"), "rb") as f, gzip.open(os.path.join(d, "in", "z.fastq.gz"), "wb") as g:
            g.write(f.read())
        for args in (["in/a.part"], ["in/z.fastq"], ["in/z"]):
            cmd = [tool, "-k40", "-t1", "-T2"] + args
            if tool == ref:
                cmd.insert(1, "-P" + d)
            subprocess.run(cmd, check=True, cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        p = subprocess.run([tool, "-k40", "in/nothing"], cwd=d, capture_output=True, text=True)
        assert p.returncode == 1 and "Cannot open in/nothing as a" in p.stderr
        gdirs.append(d)
    assert listing(gdirs[0]) == listing(gdirs[1])
    for f in listing(gdirs[0]):
        if f.endswith(".hist"):
            assert open(os.path.join(gdirs[0], f), "rb").read() == open(os.path.join(gdirs[1], f), "rb").read(), f

    for n, extra in enumerate(([], ["-Nout/named"], ["-Nplain"])):
        dirs = []
        for tool in (exe, ref):
            d = str(tmp_path / ("t%d_%d" % (n, len(dirs))))
            os.makedirs(d)
            layout(d)
            cmd = [tool, "-k40", "-t1", "-T2"] + extra + ["in/a.part.fasta", "in/sub/b.fa"]
            if tool == ref:
                cmd.insert(1, "-P" + d)
            subprocess.run(cmd, check=True, cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            dirs.append(d)
        assert listing(dirs[0]) == listing(dirs[1]), extra
        hist = [f for f in listing(dirs[0]) if f.endswith(".hist")]
        assert len(hist) == 1
        mine = open(os.path.join(dirs[0], hist[0]), "rb").read()
        assert mine == open(os.path.join(dirs[1], hist[0]), "rb").read()
        assert hashlib.sha256(mine).hexdigest() == case["expected"]["hist_sha256"]
        for tool, d in zip((exe, ref), dirs):                       # one input type per run, both tools
            p = subprocess.run([tool, "-k40", "-T2", "in/a.part.fasta", "in/sub/c.fq"], cwd=d, capture_output=True, text=True)
            assert p.returncode == 1 and "All files must be of the same type" in p.stderr


def test_profiles_of_reads_pushed_in_pieces(tmp_path):
    """A read longer than a DATA_BLOCK arrives as pieces: every block but the last is pushed with rem = 1
    and the next block of the same thread repeats the last K-1 bases (io.c:557-570).  Counting sees every
    k-mer once, and the profile is the whole read's, not one per piece."""
    import os, subprocess
    k = 40
    rng = np.random.default_rng(21)
    genome = "".join("acgt"[x] for x in rng.integers(0, 4, size=30000))
    reads = []
    for i in range(40):
        L = int(rng.integers(30, 6000))
        s0 = int(rng.integers(0, len(genome) - L))
        reads.append(genome[s0:s0 + L] if i % 7 else genome[s0:s0 + L].replace("a", "N", 1))
    bases, boff = orc.block_from_reads(reads)
    exp = orc.fastk(k, bases, boff, cutoff=1)
    want = orc.profile_counts(k, bases, boff, exp.table)
    with fastk_amd.Context(kmer=k, table_cutoff=1) as ctx:
        owner = [int(rng.integers(0, 3)) for _ in reads]
        order = sorted(range(len(reads)), key=lambda i: owner[i])   # data-set order: thread 0's reads, then 1's, 2's
        pending = {t: [i for i in order if owner[i] == t] for t in range(3)}
        state = {t: None for t in range(3)}                  # (read index, next start) of a read in progress
        while any(pending[t] or state[t] for t in range(3)):
            t = int(rng.integers(0, 3))
            if state[t] is None:
                if not pending[t]:
                    continue
                state[t] = (pending[t].pop(0), 0)
            i, st = state[t]
            r = reads[i]
            cut = len(r) if len(r) - st < 300 or rng.random() < 0.3 else st + int(rng.integers(k, len(r) - st))
            piece = r[st:cut].encode() + b"\x00"
            arr = np.frombuffer(piece, dtype=np.uint8)
            last = (cut == len(r))
            ctx.push_block(arr, np.array([0, len(arr)], dtype=np.int32), rem=0 if last else 1, tid=t)
            state[t] = None if last else (i, cut - (k - 1))
        res = ctx.finish()
        assert res.ninst == exp.ninst and np.array_equal(res.hist, exp.hist) and np.array_equal(res.table, exp.table)
        data, offs = ctx.make_profiles()
    raw = data.tobytes()
    assert len(offs) == len(reads) + 1
    for j, i in enumerate(order):
        assert raw[offs[j]:offs[j + 1]] == orc.profile_encode(want[i]), "read %d" % i

    # the reference's main() over the shim cuts a 2.6 Mbp read into 1 MB blocks itself: same profiles as
    # the reference's own run
    exe = os.path.join(orc.REF_DIR, "FastK_gpu")
    ref = os.path.join(orc.REF_DIR, "FastK")
    if not (os.path.exists(exe) and os.path.exists(ref)):
        return
    big = "".join("acgt"[x] for x in rng.integers(0, 4, size=2600000))
    b2, o2 = orc.block_from_reads([reads[0], big, reads[1], big[:1500000]])
    outs = []
    for tool, name in ((exe, "g"), (ref, "r")):
        d = str(tmp_path / name)
        os.makedirs(d)
        orc.write_fasta(os.path.join(d, "x.fasta"), b2, o2, width=100)
        subprocess.run([tool, "-k%d" % k, "-t1", "-T2", "-p", "-P" + d, os.path.join(d, "x.fasta")], check=True,
                       cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        kk, enc = orc.read_profiles(d, "x")
        outs.append([orc.profiles_digest([orc.profile_decode(e)]) for e in enc])
    assert outs[0] == outs[1] and len(outs[0]) == 4


@pytest.mark.parametrize("fmt", ["fastq", "fasta"])
def test_cli_profiles_of_homopolymer_compressed_reads(fmt, tmp_path):
    """-c -p: the profiles are those of the compressed reads (io.c:284-294 applied first), as in the reference."""
    import os, subprocess
    case, bases, boff = util.load_case("synth_illumina_k40_t1_T4")
    ref = os.path.join(orc.REF_DIR, "FastK")
    if not os.path.exists(ref):
        util.no_reference("reference build not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = util.driver_exe("FastK_amd")
    outs = []
    for tool, name in ((exe, "g"), (ref, "r")):
        d = str(tmp_path / name)
        os.makedirs(d)
        path = os.path.join(d, "x." + fmt)
        if fmt == "fastq":
            orc.write_fastq(path, bases, boff)
        else:
            orc.write_fasta(path, bases, boff, width=70)
        cmd = [tool, "-k40", "-t1", "-T3", "-c", "-p", path]
        if tool == ref:
            cmd.insert(1, "-P" + d)
        subprocess.run(cmd, check=True, cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        kk, enc = orc.read_profiles(d, "x")
        outs.append((orc.profiles_digest([orc.profile_decode(e) for e in enc]), len(enc),
                     open(os.path.join(d, "x.hist"), "rb").read()))
    assert outs[0] == outs[1]


@pytest.mark.parametrize("flags", [["-bc8", "-p"], ["-c"], ["-c", "-bc5", "-p"]])
def test_barcode_prefix_and_compression_with_profiles(flags, tmp_path):
    """-bc<n> with -p: the profile is the trimmed read's (reference behaviour); -c: homopolymer-compressed
    reads.  FastK_amd and the reference's main() over the shim against the reference itself: .hist bytes,
    table stream and decoded profiles."""
    import os, subprocess
    case, bases, boff = util.load_case("edge_k40_t1_T4")
    ref = os.path.join(orc.REF_DIR, "FastK")
    shim = os.path.join(orc.REF_DIR, "FastK_gpu")
    if not os.path.exists(ref):
        util.no_reference("reference build not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tools = [("r", ref), ("a", util.driver_exe("FastK_amd"))]
    if os.path.exists(shim):
        tools.append(("s", shim))
    outs = {}
    for name, tool in tools:
        d = str(tmp_path / name)
        os.makedirs(d)
        path = os.path.join(d, "x.fasta")
        orc.write_fasta(path, bases[:boff[6000]], boff[:6001], width=80)
        cmd = [tool, "-k40", "-t1", "-T2"] + flags + [path]
        if name != "a":
            cmd.insert(1, "-P" + d)
        subprocess.run(cmd, check=True, cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        prof = None
        if "-p" in flags:
            kk, enc = orc.read_profiles(d, "x")
            prof = (len(enc), orc.profiles_digest([orc.profile_decode(e) for e in enc]))
        outs[name] = (open(os.path.join(d, "x.hist"), "rb").read(), orc.read_ktab(os.path.join(d, "x"))["stream_sha256"], prof)
    for name in outs:
        assert outs[name] == outs["r"], name


@pytest.mark.parametrize("fmt,extra", [("fastq", []), ("fasta", ["-bc6"]), ("bam", ["-c"])])
def test_cli_profiles_with_memory_budget(fmt, extra, tmp_path):
    """-p with -M: the counting pass is chunked (reads dropped on the way), a second pass over the input looks
    the reads up piece by piece.  Every output file equals the resident run's."""
    import os, subprocess
    case, bases, boff = util.load_case("synth_illumina_k40_t1_T4")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = util.driver_exe("FastK_amd")
    reads = [bases[boff[i]:boff[i + 1] - 1].tobytes().decode() for i in range(len(boff) - 1)]
    outs = []
    for name, mem in (("res", []), ("mem", ["-M1"])):
        d = str(tmp_path / name)
        os.makedirs(d)
        path = os.path.join(d, "x." + fmt)
        if fmt == "fastq":
            orc.write_fastq(path, bases, boff)
        elif fmt == "fasta":
            orc.write_fasta(path, bases, boff, width=60)
        else:
            orc.write_bam(path, reads)
        subprocess.run([exe, "-k40", "-t2", "-T3", "-p"] + extra + mem + [path], check=True, cwd=d,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        files = sorted(f for f in os.listdir(d) if f != "x." + fmt)
        outs.append({f: open(os.path.join(d, f), "rb").read() for f in files})
    assert sorted(outs[0]) == sorted(outs[1]) and len(outs[0]) == 1 + 4 + 1 + 6
    for f in outs[0]:
        assert outs[0][f] == outs[1][f], f


# ------------------------------------------------------------------------------ round 6: the k-mer stage by references

def _domain_reads(k, seed, n=2500):
    """reads with everything the cut into minimizer domains has to get right: both strands, errors (pieces of different
    super-mers that hold the same k-mers), reads of k-1 .. k+2, N runs, homopolymers and short tandem repeats (one
    16-mer at every start: a domain as long as the super-mer; the same minimizer VALUE at several starts), a read
    that is its own reverse complement, one read copied 40,000 times (weights beyond 0x7fff)."""
    rng = np.random.default_rng(seed)
    genome = rng.integers(0, 4, size=60000)
    reads = []
    for _ in range(n):
        L = int(rng.choice([k - 1, k, k + 1, k + 2, 90, 150, 400, 1500]))
        s0 = int(rng.integers(0, len(genome) - L))
        r = genome[s0:s0 + L].copy()
        for j in range(L):
            if rng.random() < 0.004:
                r[j] = rng.integers(0, 4)
        if rng.random() < 0.5:
            r = (3 - r)[::-1]
        if rng.random() < 0.1 and L > 5:
            a = int(rng.integers(0, L))
            r[a:a + int(rng.integers(1, 4))] = 4
        reads.append("".join("acgtn"[x] for x in r))
    unit = "".join("acgt"[x] for x in genome[100:100 + 3 * k])
    comp = {"a": "t", "c": "g", "g": "c", "t": "a"}
    pal = unit[:k] + "".join(comp[c] for c in reversed(unit[:k]))
    reads += ["a" * 300] * 20 + ["ac" * 150] * 20 + ["acg" * 100] * 20 + ["aacgt" * 60] * 20 + [pal] * 7
    reads += [unit] * 40000
    return orc.block_from_reads(reads)


@pytest.mark.parametrize("k", [32, 33, 37, 40, 41, 48, 51, 56, 63, 64])
def test_kmer_stage_by_references_matches_oracle_and_hashed_grouping(k):
    """Round 6 (fk_recut.hip): for k from 32 the weighted k-mers are not grouped by two passes any more -- distinct
    super-mers are cut into minimizer domains, 8-byte references are sorted, the expansion writes the k-mers grouped.
    Against the oracle bit for bit and against the hashed grouping (fk_debug_set("kmer_stage", 2)), one and several
    buckets, with fills of 8192 and of 24 records (every group larger than a fill: chunks and selections), table
    cutoffs 1 and 4; res.nrefs says which stage ran."""
    bases, boff = _domain_reads(k, 600 + k)
    for cutoff in (1, 4):
        exp = orc.fastk(k, bases, boff, cutoff=cutoff)
        for nb, stage, limit in ((1, 0, 0), (3, 0, 0), (1, 2, 0), (2, 0, 24)):
            with fastk_amd.Context(kmer=k, table_cutoff=cutoff, nbuckets=nb) as ctx:
                if stage:
                    ctx.debug_set("kmer_stage", stage)
                if limit:
                    ctx.debug_set("aggr_limit", limit)
                ctx.push_block(bases, boff.astype(np.int32))
                res = ctx.finish()
                what = (k, cutoff, nb, stage, limit)
                by_refs = (stage == 0 and ctx.w.smer_stride // 4 <= 7)        # (wider super-mer records, k from 61: no LDS
                assert (res.nrefs > 0) == by_refs, what                       #  de-duplication, hence no references)
                if by_refs:
                    assert res.ndistinct_super <= res.nrefs <= res.nweighted, what
                assert res.ninst == exp.ninst, what
                assert np.array_equal(res.hist, exp.hist) and res.max_inst == exp.max_inst, what
                assert res.ntable == exp.ntable and np.array_equal(res.table, exp.table), what


def test_kmer_stage_below_32_keeps_the_hashed_grouping():
    with fastk_amd.Context(kmer=31, table_cutoff=1) as ctx:
        bases, boff = _domain_reads(31, 5, n=300)
        exp = orc.fastk(31, bases, boff, cutoff=1)
        ctx.push_block(bases, boff.astype(np.int32))
        res = ctx.finish()
        assert res.nrefs == 0
        assert np.array_equal(res.hist, exp.hist) and np.array_equal(res.table, exp.table)


def _device_count():
    import ctypes
    for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"):
        try:
            lib = ctypes.CDLL(name)
        except OSError:
            continue
        n = ctypes.c_int(0)
        if lib.hipGetDeviceCount(ctypes.byref(n)) == 0:
            return n.value
    import torch
    return torch.cuda.device_count()


@pytest.mark.parametrize("ranks,opts,name,fmt", [(2, [], "synth_hifi_k40_t4_T8", "fasta"), (2, ["-p"], "synth_illumina_k51_t1_T4", "fastq"),
                                                 (4, [], "configs0_k40_t1_T4", "fastq"), (8, ["-M2"], "configs0_k40_t1_T4", "fastq")])
def test_c_driver_sharded_on_distinct_devices(ranks, opts, name, fmt, tmp_path):
    """FastK_amd -G<n> with every rank on a DEVICE OF ITS OWN (device = local rank, RCCL's peer-to-peer transports over
    xGMI): what no box of this pool can run -- they have one GPU, where the -G tests let the ranks share it over RCCL's
    socket transport -- and what the driver's 8-GPU node runs before its scaling bench does.
    Skipped below `ranks` devices.  Every file byte-identical to the one-GPU run with the same -T, .hist and canonical
    .ktab stream the REFERENCE's (golden digests), decoded profiles the reference's where -p is given."""
    import os, subprocess
    ndev = _device_count()
    if ndev < ranks:
        pytest.skip("%d device(s) here; the test needs %d" % (ndev, ranks))
    case, bases, boff = util.load_case(name)
    exp = case["expected"]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = util.driver_exe("FastK_amd")
    path = str(tmp_path / ("reads." + fmt))
    util.write_fastx(path, bases, boff, fmt == "fastq")
    T = 8
    args = ["-k%d" % case["k"], "-t%d" % case["cutoff"], "-T%d" % T] + opts
    one, many = tmp_path / "one", tmp_path / "many"
    one.mkdir(); many.mkdir()
    subprocess.run([exe] + [a for a in args if not a.startswith("-M")] + ["-N" + str(one / "x"), path], check=True)
    env = {k: v for k, v in os.environ.items() if k != "FK_RANKS_SHARE_GPU"}
    p = subprocess.run([exe] + args + ["-v", "-G%d" % ranks, "-N" + str(many / "x"), path], env=env, capture_output=True, text=True,
                       timeout=900)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    assert util.sha_file(many / "x.hist") == exp["hist_sha256"]
    t = orc.read_ktab(str(many / "x"))
    assert t["stream_sha256"] == exp["ktab"]["stream_sha256"] and t["nels"] == exp["ktab"]["nels"]
    for f in sorted(os.listdir(one)):
        if "ktab" in f or f.endswith(".hist"):
            assert util.sha_file(one / f) == util.sha_file(many / f), f
    if "-p" in opts:
        assert _prof_stream(one, "x") == _prof_stream(many, "x")
