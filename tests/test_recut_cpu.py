"""The rule behind round 6's k-mer stage (fastk_amd/csrc/fk_recut.hip), restated in numpy and checked on the CPU.

The device no longer moves the W weighted k-mer records to bring equal k-mers together (the role of
Weighted_Kmer_Sort, MSDsort.c:536-544, in front of hist_kmers, MSDsort.c:491-509): every distinct super-mer is cut
where M changes -- M(x) = the smallest rank of the canonical 16-mers inside the k-mer x -- 8-byte references to the
pieces are sorted on a 22-bit mix of M, the expansion writes the k-mers in that order and the aggregation's fills are
cut between key groups.  That is only right if

  (1) M is a function of the k-mer alone and the same for a k-mer and its reverse complement -- so that EVERY copy of a
      canonical k-mer, from whichever read, strand and super-mer, carries the same key;
  (2) the pieces of a super-mer partition its k-mers, each piece's k-mers sharing M;
  (3) fills cut at key-group boundaries never part two records of one k-mer (a consequence of (1)).

The kernels' own output was checked against the oracle bit for bit on an MI355X (tests/test_gpu_parity.py::
test_kmer_stage_by_references_matches_oracle_and_hashed_grouping; log: profiles/r06_b_kmer_stage_by_references_tests.log);
this file pins the arithmetic -- rc_rank, the rolled reverse complement, the doubling window minimum with the look
d = w - 2^lg further on, the reference word -- so that a change of any of them has a CPU test to answer to."""
import numpy as np
import pytest

MLEN, KEY_BITS, IDX_BITS = 16, 22, 28          # FK_REF_MLEN, FK_REF_KEY_BITS, FK_REF_IDX_BITS (fk_common.h)
M32 = np.uint64(0xffffffff)


def rc_rank(c):
    """fk_recut.hip rc_rank: (c ^ 0x5bd1e995) * 0x9E3779B1, xor-shifted by 15 (32-bit arithmetic)"""
    x = ((c ^ np.uint64(0x5bd1e995)) * np.uint64(0x9E3779B1)) & M32
    return x ^ (x >> np.uint64(15))


def codes16(seq):
    """forward and reverse-complement codes of every 16-mer start of seq (values 0..3), first base in the high bits --
    the forward code as the kernel's funnel shift of two record words gives it, the reverse complement as it rolls it:
    rc' = (rc >> 2) | (complement of the entering base) << 30"""
    n = len(seq) - MLEN + 1
    s = seq.astype(np.uint64)
    f = np.zeros(n, dtype=np.uint64)
    for i in range(MLEN):
        f = (f << np.uint64(2)) | s[i:i + n]
    rc = np.zeros(n, dtype=np.uint64)
    r0 = 0
    for i in range(MLEN):                         # rc of the first 16-mer: complement, order of the bases reversed
        r0 |= (3 - int(seq[i])) << (2 * i)
    rc[0] = r0
    for p in range(1, n):
        rc[p] = (rc[p - 1] >> np.uint64(2)) | (np.uint64(3 - int(seq[p + MLEN - 1])) << np.uint64(30))
    return f, rc


def window_min_like_the_kernel(h, w):
    """min over h[j .. j + w - 1] the way k_recut takes it: minima over 2^lg consecutive starts by doubling, then
    min(A[j], A[j + d]) with d = w - 2^lg"""
    lg = int(w).bit_length() - 1
    A = h.copy()
    s = 1
    while s < (1 << lg):
        B = A.copy()
        B[:len(A) - s] = np.minimum(A[:len(A) - s], A[s:])
        A = B
        s <<= 1
    d = w - (1 << lg)
    n = len(h) - w + 1
    return np.minimum(A[:n], A[d:d + n])


def kmer_M(seq, k):
    f, rc = codes16(seq)
    h = rc_rank(np.minimum(f, rc))
    return window_min_like_the_kernel(h, k - MLEN + 1)


def key_of(M):
    return ((M * np.uint64(0x9E3779B1)) & M32) >> np.uint64(32 - KEY_BITS)


def canonical(seq, j, k):
    a = seq[j:j + k]
    b = (3 - a)[::-1]
    return min(bytes(a.astype(np.uint8)), bytes(b.astype(np.uint8)))


@pytest.mark.parametrize("k", [32, 40, 51, 64])
def test_window_minimum_is_the_plain_minimum(k):
    rng = np.random.default_rng(k)
    seq = rng.integers(0, 4, size=400)
    f, rc = codes16(seq)
    # the rolled reverse complement is the reverse complement
    for p in (0, 1, 17, 200, len(f) - 1):
        a = seq[p:p + MLEN]
        want = 0
        for x in (3 - a)[::-1]:
            want = (want << 2) | int(x)
        assert int(rc[p]) == want
    h = rc_rank(np.minimum(f, rc))
    w = k - MLEN + 1
    M = window_min_like_the_kernel(h, w)
    plain = np.array([h[j:j + w].min() for j in range(len(seq) - k + 1)], dtype=np.uint64)
    assert np.array_equal(M, plain)


@pytest.mark.parametrize("k", [32, 37, 40, 51, 64])
def test_every_copy_of_a_kmer_carries_the_same_key(k):
    """(1): reads from both strands with errors, tandem repeats and homopolymers -- whatever copy of a canonical k-mer
    is met, its M (hence its key) is the same; and keys separate: the k-mers of 2,000 reads use many of them"""
    rng = np.random.default_rng(100 + k)
    genome = rng.integers(0, 4, size=20000)
    reads = []
    for _ in range(2000):
        L = int(rng.choice([k, k + 1, 90, 150, 400]))
        s0 = int(rng.integers(0, len(genome) - L))
        r = genome[s0:s0 + L].copy()
        for j in range(L):
            if rng.random() < 0.004:
                r[j] = rng.integers(0, 4)
        if rng.random() < 0.5:
            r = (3 - r)[::-1].copy()
        reads.append(r)
    reads += [np.zeros(200, dtype=np.int64), np.tile(np.array([0, 1]), 100), np.tile(np.array([0, 1, 2]), 70),
              np.tile(np.array([0, 0, 1, 2, 3]), 40)]
    pal = genome[500:500 + k]
    reads.append(np.concatenate([pal, (3 - pal)[::-1]]))            # a read that is its own reverse complement
    seen = {}
    keys = set()
    for r in reads:
        if len(r) < k:
            continue
        M = kmer_M(r, k)
        K = key_of(M)
        for j in range(len(r) - k + 1):
            c = canonical(r, j, k)
            m = int(M[j])
            if c in seen:
                assert seen[c] == m, "two copies of one k-mer with different minimizer ranks"
            else:
                seen[c] = m
            keys.add(int(K[j]))
    assert len(seen) > 10000 and len(keys) > 1000


@pytest.mark.parametrize("k", [32, 40, 51])
def test_pieces_partition_a_supermer_and_the_reference_word_holds_them(k):
    """(2) and the reference layout: key on top (the sort takes bytes 5, 6, 7 of the little-endian word), then which
    super-mer, the piece's first k-mer, its k-mers"""
    rng = np.random.default_rng(7 * k)
    for trial in range(300):
        n = int(rng.integers(1, k - 3))                             # k-mers of a super-mer: 1 .. k - 4
        sm = rng.integers(0, 4, size=n + k - 1)
        M = kmer_M(sm, k)
        assert len(M) == n
        starts = [0] + [j for j in range(1, n) if M[j] != M[j - 1]]
        ends = starts[1:] + [n]
        covered = []
        for a, b in zip(starts, ends):
            assert len(set(int(x) for x in M[a:b])) == 1
            covered += list(range(a, b))
            key = int(key_of(M[a:a + 1])[0])
            idx = trial
            ref = (key << 42) | (idx << 14) | (a << 7) | (b - a)    # fk_ref_pack
            assert ref < (1 << 64) and key < (1 << KEY_BITS) and idx < (1 << IDX_BITS) and b - a <= k - 4 <= 127
            assert (ref & 127, (ref >> 7) & 127, (ref >> 14) & ((1 << IDX_BITS) - 1), ref >> 42) == (b - a, a, idx, key)
            raw = int(ref).to_bytes(8, "little")
            assert (raw[7] << 14) | (raw[6] << 6) | (raw[5] >> 2) == key     # bytes 5..7 order the references by key
        assert covered == list(range(n))


def test_fills_cut_between_key_groups_never_part_a_kmer():
    """(3), with k_ref_bounds' rule restated: fill f begins at the first key group that begins at or behind record
    f * target of the expansion order"""
    k, target = 40, 96
    rng = np.random.default_rng(11)
    genome = rng.integers(0, 4, size=6000)
    pieces = []                                                      # (key, canonical k-mers of the piece)
    for _ in range(400):
        L = int(rng.integers(k, 300))
        s0 = int(rng.integers(0, len(genome) - L))
        r = genome[s0:s0 + L].copy()
        if rng.random() < 0.5:
            r = (3 - r)[::-1].copy()
        M = kmer_M(r, k)
        K = key_of(M)
        a = 0
        for j in range(1, len(M) + 1):
            if j == len(M) or M[j] != M[j - 1]:
                pieces.append((int(K[a]), [canonical(r, x, k) for x in range(a, j)]))
                a = j
    pieces.sort(key=lambda p: p[0])                                  # the sort of the references
    recs, group_start = [], []
    for i, (key, kms) in enumerate(pieces):
        if i == 0 or pieces[i - 1][0] != key:
            group_start.append(len(recs))
        recs += kms
    W = len(recs)
    nf = (W + target - 1) // target
    bounds = []
    for f in range(nf):
        g = f * target
        nxt = [s for s in group_start if s >= g]
        bounds.append(nxt[0] if nxt else W)
    bounds.append(W)
    assert bounds[0] == 0 and all(b1 >= b0 for b0, b1 in zip(bounds, bounds[1:]))
    where = {}
    for f in range(nf):
        for c in recs[bounds[f]:bounds[f + 1]]:
            assert where.setdefault(c, f) == f, "a k-mer's records lie in two fills"
    assert len(where) > 1000
