"""Pin the oracle against the REFERENCE itself, run here: oracle/_ref/FastK (built by
oracle/Makefile from /root/reference).  The module is skipped only where neither /root/reference nor a
prebuilt oracle/_ref exists; a missing piece of the reference build fails (tests/util.py, no_reference)."""
import os
import random

import numpy as np
import pytest

from oracle import orc
from tests import util

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
    # the streamed digest bench.py and tools/cpu_baseline_full.py take of tables too large to load (the reference's
    # 27 GB of parts at configs[2]) is the canonical-stream digest of the golden fixtures
    import bench
    dig, nels, nparts = bench.ktab_stream_sha256(d, "x")
    t = orc.read_ktab(os.path.join(d, "x"))
    assert (dig, nels, nparts) == (t["stream_sha256"], t["nels"], T)
    # the reference's own checker accepts the oracle's table
    import subprocess
    out = subprocess.run([os.path.join(orc.REF_DIR, "Tabex"), "-C", os.path.join(od, "x")],
                         capture_output=True, text=True)
    assert "Table is OK" in out.stdout + out.stderr


def _tie_heavy_reads(seed, nreads, lengths):
    """homopolymers, tandem repeats (some equal to their reverse complement), N runs, random stretches"""
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
                if kind >= 4 and ln > 10:
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


@pytest.mark.parametrize("profile", [False, True], ids=["plain", "with -p"])
@pytest.mark.parametrize("k,T,seed,shape", [(40, 4, 31, (260, (40, 60, 150, 400, 1500, 6000))), (21, 2, 32, (260, (40, 60, 150, 400, 1500))),
                                            (51, 3, 33, (40, (39, 5000, 30000))), (12, 1, 34, (200, (5, 12, 13, 100, 900))),
                                            (8, 2, 35, (200, (8, 9, 40, 300)))])
def test_oracle_on_reads_full_of_ties_and_in_profile_mode(k, T, seed, shape, profile, tmp_path):
    """Pins two things the random reads above do not reach.  Ties: equal minimizer values inside one window (`<` on
    arrival, `<=` on the forced rescan, split.c:1149,1306-1315) decide the super-mer cuts, hence the distinct super-mers,
    hence the first-byte census that places the hidden .ktab part boundaries.  And -p: the reference then keeps its
    super-mers on the read's strand (split.c:1245: Stuff_Seq(..., 0, ...) under DO_PROFILE) -- a super-mer and its reverse
    complement are two records and the boundaries move (orc.fastk(profile=True)).  Every .hist / .ktab file of the
    reference run live with and without -p."""
    import subprocess
    bases, boff = _tie_heavy_reads(20260000 + seed, *shape)
    d = str(tmp_path)
    path = os.path.join(d, "x.fasta")
    orc.write_fasta(path, bases, boff)
    subprocess.run([os.path.join(orc.REF_DIR, "FastK"), "-k%d" % k, "-t1", "-T%d" % T, "-P" + d] + (["-p"] if profile else []) + [path],
                   check=True, cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    res = orc.fastk(k, bases, boff, cutoff=1, nthreads=T, profile=profile)
    od = os.path.join(d, "o")
    os.mkdir(od)
    orc.write_outputs(res, 1, T, od, "x")
    for f in ["x.hist", "x.ktab"] + [".x.ktab.%d" % (t + 1) for t in range(T)]:
        assert open(os.path.join(d, f), "rb").read() == open(os.path.join(od, f), "rb").read(), f
    if profile:                                # ... and what a profile IS (orc.profile_counts) against the reference's files
        kk, enc = orc.read_profiles(d, "x")
        want = orc.profile_counts(k, bases, boff, res.table)
        assert kk == k and len(enc) == len(want)
        for i, e in enumerate(enc):
            assert list(orc.profile_decode(e)) == list(want[i]), i


@pytest.mark.parametrize("k,T,bc,profile,seed", [(40, 4, 8, False, 71), (40, 3, 5, True, 72), (21, 2, 20, False, 73),
                                                 (51, 5, 1, True, 74), (33, 4, 40, False, 75)])
def test_oracle_with_a_barcode_prefix_on_reads_full_of_ties(k, T, bc, profile, seed, tmp_path):
    """-bc<n> (the first n bases of every read do not count, io.c / split.c:1077-1079) beside the ties and -p."""
    import subprocess
    bases, boff = _tie_heavy_reads(20260000 + seed, 300, (40, 60, 150, 400, 1500, 6000))
    d = str(tmp_path)
    path = os.path.join(d, "x.fasta")
    orc.write_fasta(path, bases, boff)
    subprocess.run([os.path.join(orc.REF_DIR, "FastK"), "-k%d" % k, "-t1", "-T%d" % T, "-bc%d" % bc, "-P" + d]
                   + (["-p"] if profile else []) + [path], check=True, cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    res = orc.fastk(k, bases, boff, cutoff=1, nthreads=T, bc_prefix=bc, profile=profile)
    od = os.path.join(d, "o")
    os.mkdir(od)
    orc.write_outputs(res, 1, T, od, "x")
    for f in ["x.hist", "x.ktab"] + [".x.ktab.%d" % (t + 1) for t in range(T)]:
        assert open(os.path.join(d, f), "rb").read() == open(os.path.join(od, f), "rb").read(), f


def test_oracle_scheme_on_long_reads_full_of_ties(tmp_path):
    """100 M bases of long tie-heavy reads with -M1: the reference deals the minimizers to two buckets (padded-minimizer
    trie, assign_pieces) and cuts the hidden parts by bucket 0's census; orc.fastk_parts follows -- all six files.  (~40 s:
    the one slow test of the CPU suite; it is what test_exact_parts_long_reads_in_several_buckets' oracle-free check and
    the fuzz harness's exact leg lean on for inputs beyond the fixtures.)"""
    import subprocess
    bases, boff = _tie_heavy_reads(20260099, 6000, (5000, 15000, 30000))
    k, T = 40, 4
    d = str(tmp_path)
    path = os.path.join(d, "x.fasta")
    orc.write_fasta(path, bases, boff)
    p = subprocess.run([os.path.join(orc.REF_DIR, "FastK"), "-k%d" % k, "-t1", "-T%d" % T, "-M1", "-v", "-P" + d, path],
                       cwd=d, capture_output=True, text=True)
    assert p.returncode == 0 and "Dividing data into 2 blocks" in p.stderr
    res = orc.fastk_parts(k, bases, boff, 1000000000, cutoff=1, nthreads=T)
    od = os.path.join(d, "o")
    os.mkdir(od)
    orc.write_outputs(res, 1, T, od, "x")
    for f in ["x.hist", "x.ktab"] + [".x.ktab.%d" % (t + 1) for t in range(T)]:
        assert util.sha_file(os.path.join(d, f)) == util.sha_file(os.path.join(od, f)), f


@pytest.mark.parametrize("rsize,n,nbytes,T", [(12, 50000, 10, 4), (20, 30011, 19, 3), (16, 1000, 5, 1), (12, 7, 10, 4)])
def test_oracle_lsd_engine_equals_reference_lsd_sort(rsize, n, nbytes, T):
    """Sort-engine unit parity against the reference's own LSD_Sort (libfkref.so = LSDsort.c compiled
    where it lies): same records in, same bytes out, ties (partial keys) in input order."""
    if not orc.have_fkref():
        util.no_reference("oracle/_ref/libfkref.so not built (needs the reference sources at build time)")
    rng = np.random.default_rng(rsize * 1000 + n)
    recs = rng.integers(0, 256, size=(n, rsize), dtype=np.uint8)
    recs[:, 1] = rng.integers(0, 3, size=n)
    order = list(range(nbytes - 1, -1, -1))
    assert np.array_equal(orc.lsd_sort(recs, order), orc.ref_lsd_sort(recs, order, T))


@pytest.mark.parametrize("kmer,n,distinct,T,heavy", [(40, 200000, 50000, 4, False), (40, 30011, 30011, 3, False),
                                                     (25, 60000, 900, 1, True), (51, 5000, 4000, 4, True), (40, 3, 2, 4, False)])
def test_oracle_msd_engine_and_counting_equal_reference_weighted_kmer_sort(kmer, n, distinct, T, heavy):
    """The MSD engine and the counting it drives, unit level: the reference's own Weighted_Kmer_Sort (MSDsort.c:536-544:
    msd_sort -> radix_sort / shell_sort, hist_kmers on every run of equal k-mers; libfkref.so = MSDsort.c compiled where
    it lies) against the oracle's restatement (orc_msd_sort + orc_count_sorted) on the same weighted k-mers: same key
    order, the same summed count in every run's first record, the same histogram and max_inst -- also with runs whose
    sum passes 0x7fff."""
    if not orc.have_fkref():
        util.no_reference("oracle/_ref/libfkref.so not built (needs the reference sources at build time)")
    rng = np.random.default_rng(kmer * 100000 + n)
    kb = (kmer + 3) // 4
    keys = rng.integers(0, 256, size=(distinct, kb), dtype=np.uint8)
    keys[:, 0] = rng.integers(0, 7, size=distinct) * 37                 # few first bytes: long partitions
    if kmer & 3:
        keys[:, kb - 1] &= (0xff << (2 * (4 - (kmer & 3)))) & 0xff       # unused low bits are zero in a real list
    recs = np.zeros((n, kb + 2), dtype=np.uint8)
    pick = rng.integers(0, distinct, size=n)
    if heavy:
        pick[: n // 2] = pick[0]                                         # one k-mer in half of the records
    recs[:, :kb] = keys[pick]
    w = rng.integers(1, 40 if heavy else 5, size=n)
    recs[:, kb] = w & 0xff
    recs[:, kb + 1] = w >> 8
    out, hist, max_inst = orc.ref_weighted_kmer_sort(recs, kmer, T)
    P = orc.params(kmer)
    mine = orc.msd_sort(recs, kb)
    res = orc.count_sorted(P, mine, 1)
    ref_keys = out[:, :kb].copy()
    ref_keys[:, 0] = np.sort(recs[:, 0], kind="stable")                  # the engine flags run heads in byte 0
    assert np.array_equal(ref_keys, mine[:, :kb])
    heads = np.ones(n, dtype=bool)
    heads[1:] = (mine[1:, :kb] != mine[:-1, :kb]).any(axis=1)
    assert (out[heads, 0] == 1).all()
    ref_counts = out[heads, kb].astype(np.int64) | (out[heads, kb + 1].astype(np.int64) << 8)
    table = np.asarray(res.table)
    assert table.shape[0] == int(heads.sum())
    assert np.array_equal(table[:, :kb], mine[heads, :kb])
    assert np.array_equal(table[:, kb].astype(np.int64) | (table[:, kb + 1].astype(np.int64) << 8), ref_counts)
    assert np.array_equal(hist[1:], np.asarray(res.hist, dtype=np.int64)[1:0x8000])
    assert max_inst == int(res.max_inst)
    if heavy:
        assert max_inst > 0


def test_scheme_equals_the_reference_trainer(tmp_path):
    """Determine_Scheme restated (orc.scheme: trainer census, refine_tree, assign_pieces with glibc's unseeded
    drand48) against the reference's own DEBUG_SCHEME print-out (oracle/_ref/FastK_scheme: split.c compiled with that
    macro on): the prefix trie and the bucket of every leaf, for BASELINE configs[0] with -M1 (two buckets)."""
    import subprocess
    from tests import util
    exe = os.path.join(orc.REF_DIR, "FastK_scheme")
    if not os.path.exists(exe):
        import subprocess as sp
        sp.run(["make", "-s", "-C", os.path.dirname(orc.REF_DIR), "ref_scheme"], check=False)
    if not os.path.exists(exe):
        util.no_reference("oracle/_ref/FastK_scheme not built (needs the reference sources)")
    bases, boff = orc.synth_block(20251001, 10000000, 150, 1000, 0, 1000000)
    path = os.path.join(str(tmp_path), "x.fastq")
    util.write_fastx(path, bases, boff, True)
    out = subprocess.run([exe, "-k40", "-t1", "-T4", "-M1", "-v", "-P" + str(tmp_path), path], capture_output=True, text=True,
                         cwd=str(tmp_path))
    assert out.returncode == 0 and "Dividing data into 2 blocks" in out.stderr
    P, S, part = orc.scheme(40, bases, boff, 2)
    inv = {P.tran[0]: "a", P.tran[1]: "c", P.tran[2]: "g", P.tran[3]: "t"}

    def show(lev, i):                                     # print_ass of split.c:420-434
        if part[i] >= 0:
            return " %d\n" % part[i]
        return "\n" + "".join("%*s -> %c:" % (2 * lev, "", inv[a]) + show(lev + 1, -part[i] + a) for a in range(4))

    mine = "".join(" %5d:" % i + show(0, i) for i in range(1024))
    ref = out.stdout[out.stdout.index("Padded Assignments"):].split("\n", 1)[1]
    assert S.nparts == 2 and S.pad == 0
    assert ref.strip() == mine.strip()


def test_reference_build_links_the_real_htslib():
    """oracle/_ref is the reference with NOTHING resolved to address 0: the seven HTSLIB functions
    io.c calls for .cram input are defined by libhts.a, compiled from the reference's HTSLIB sources by
    oracle/Makefile, and libfkref.so links with --no-undefined."""
    import subprocess
    syms = ["cram_close", "cram_get_seq", "cram_open", "hgetc2", "hread2", "hseek", "itf8_decode"]
    ref = os.path.join(os.path.dirname(os.path.abspath(orc.__file__)), "_ref")
    assert os.path.exists(os.path.join(ref, "libhts.a")), "oracle/_ref/libhts.a missing"
    for binary in ("FastK", "libfkref.so"):
        out = subprocess.run(["nm", "--defined-only", os.path.join(ref, binary)], capture_output=True, text=True,
                             check=True).stdout
        table = {l.split()[2]: (l.split()[0], l.split()[1]) for l in out.splitlines() if len(l.split()) == 3}
        for s in syms:
            assert s in table, (binary, s, "not defined")
            addr, kind = table[s]
            assert kind in "Tt" and int(addr, 16) != 0, (binary, s, addr, kind)
    mk = open(os.path.join(os.path.dirname(ref), "Makefile")).read()
    assert "--defsym" not in mk and "unresolved-symbols" not in mk
