#!/usr/bin/env python3
"""Randomised differential run of the whole path against the CPU oracle: random k, read mixes (both
strands, N's, upper/lower case, reads shorter than k, duplicates, low complexity), bucket counts,
chunked ingest with and without host spill, block sizes and input-thread ids, table cut-offs, and the
profile stage.  Every iteration compares histogram, max_inst, instance count, table and (cut-off 1)
profiles bit for bit.

  python tests/fuzz_parity.py [iterations=100] [seed=1]
  FUZZ_GUARD=0                       the reads of an iteration in ordinary memory (default: in a mapping of their own between
                                     two inaccessible pages, read-only once filled -- a host-side writer dies with a stack
                                     instead of flipping bytes; the array is also compared with its digest after every
                                     iteration, which catches a writer that does not go through the page tables)
  FUZZ_ONLY=<n> [FUZZ_FROM=<m>] python tests/fuzz_parity.py <iterations> <seed>
                                     only iteration n (or m .. n) of that run: the random draws of the iterations before
                                     are replayed without the GPU
"""
import ctypes
import faulthandler
import hashlib
import mmap
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("FASTK_AMD_TEST_KNOBS", "1")                   # (chunk sizes and spill limits are test knobs)
import fastk_amd                                                     # noqa: E402
from oracle import orc                                               # noqa: E402


_libc = ctypes.CDLL(None, use_errno=True)
_libc.mprotect.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
_PAGE = mmap.PAGESIZE


class Guarded:
    """A byte array in a mapping of its own: [inaccessible page][data, read-only][inaccessible page].  Any CPU store
    into it -- a late write of the library's host side, of the runtime's staging, of the harness -- is a SIGSEGV with a
    stack (faulthandler); `intact()` compares with the digest taken at the fill, for writers that bypass the page tables
    (a copy engine)."""

    def __init__(self, data):
        n = len(data)
        self.body = (max(n, 1) + _PAGE - 1) // _PAGE * _PAGE
        self.map = mmap.mmap(-1, self.body + 2 * _PAGE)
        self.base = ctypes.addressof(ctypes.c_char.from_buffer(self.map))
        whole = np.frombuffer(self.map, dtype=np.uint8)
        self.array = whole[_PAGE:_PAGE + n]
        self.array[:] = data
        self.sha = hashlib.sha256(self.array).digest()
        for off, ln, prot in ((0, _PAGE, 0), (_PAGE, self.body, 1), (_PAGE + self.body, _PAGE, 0)):
            if _libc.mprotect(self.base + off, ln, prot) != 0:
                raise OSError(ctypes.get_errno(), "mprotect")

    def intact(self):
        return hashlib.sha256(self.array).digest() == self.sha

    def release(self):
        _libc.mprotect(self.base, self.body + 2 * _PAGE, 3)
        del self.array                       # (the mapping itself goes when the last view of it does)


def device_reads_report(ctx, pushed):
    """the context's device copy of the pushed reads against what was pushed, block by block in push order: a text of
    what differs where (None: the reads are not resident)"""
    try:
        ptr, n = ctx.debug_get("reads_ptr"), ctx.debug_get("reads_len")
    except Exception:                                                # noqa: BLE001
        return None
    want = np.concatenate([np.asarray(b) for b in pushed]) if pushed else np.zeros(0, dtype=np.uint8)
    if not ptr or n != len(want):
        return "reads_len %d, %d bytes were pushed" % (n, len(want))
    dev = np.empty(n, dtype=np.uint8)
    ctx._ck(ctx.L.fk_copy_to_host(ctx.h, dev.ctypes.data, ptr, n))
    d = np.nonzero(dev != want)[0]
    starts = np.cumsum([0] + [len(b) for b in pushed])
    txt = "%d of %d bytes differ from the %d blocks pushed (zeros: device %d, pushed %d)" \
          % (len(d), n, len(pushed), int((dev == 0).sum()), int((want == 0).sum()))
    for p in d[:12].tolist():
        blk = int(np.searchsorted(starts, p, side="right") - 1)
        txt += "\n    at %d (block %d + %d of %d; address mod 64 = %d): device %s  pushed %s" \
               % (p, blk, p - int(starts[blk]), len(pushed[blk]), (ptr + p) % 64,
                  bytes(dev[max(0, p - 6):p + 10]), bytes(want[max(0, p - 6):p + 10]))
    return txt


class Canaries:
    """A ring of byte arrays filled with 'a' that turns over with the heap: a few are checked, freed and allocated again every
    iteration.  A byte that changes was written by something that does not own it (round 5: aligned 32-bit zeros turned up
    in the harness's own data about once per 2,000 iterations with 32 processes on one GPU)."""

    def __init__(self, n=64, seed=1):
        self.rng = np.random.default_rng(seed)
        self.ring = [self._new() for _ in range(n)]
        self.hits = 0

    def _new(self):
        return np.full(int(self.rng.integers(1024, 1 << 21)), 0x61, dtype=np.uint8)

    def turn(self, where, k=6):
        for _ in range(k):
            i = int(self.rng.integers(0, len(self.ring)))
            a = self.ring[i]
            bad = np.nonzero(a != 0x61)[0]
            if len(bad):
                self.hits += 1
                print("CANARY %s: %d bytes changed in an array of %d at 0x%x: %s" % (
                    where, len(bad), len(a), a.ctypes.data,
                    ", ".join("+%d (mod 64 = %d) = 0x%02x" % (int(p), (a.ctypes.data + int(p)) % 64, int(a[p])) for p in bad[:12])),
                    flush=True)
            self.ring[i] = self._new()


def make_reads(rng, k):
    glen = int(rng.integers(500, 40000))
    genome = rng.integers(0, 4, size=glen)
    if rng.random() < 0.3:                                           # a repeat-rich genome
        unit = genome[:int(rng.integers(1, 300))]                   # (from a homopolymer and di- / tri-nucleotide repeats up: ties
                                                                     # between the minimizers of one window)
        genome = np.concatenate([unit] * (glen // len(unit) + 1))[:glen]
    reads = []
    n = int(rng.integers(1, 1500))
    maxlen = int(rng.choice([k + 5, 150, 400, 3000]))
    for _ in range(n):
        L = int(rng.integers(max(1, k - 4), maxlen + 1))
        L = min(L, glen - 1)
        s0 = int(rng.integers(0, glen - L))
        r = genome[s0:s0 + L].copy()
        if rng.random() < 0.5:
            r = (3 - r)[::-1]
        for _ in range(int(rng.poisson(L * 0.003))):
            r[int(rng.integers(0, L))] = int(rng.integers(0, 4))
        if rng.random() < 0.15:
            a = int(rng.integers(0, L))
            r[a:a + int(rng.integers(1, 40))] = 4
        t = "".join("acgtn"[x] for x in r)
        if rng.random() < 0.3:
            t = t.upper()
        reads.append(t)
    if rng.random() < 0.3:
        reads += [reads[0]] * int(rng.integers(2, 40000 if rng.random() < 0.1 else 300))
    if rng.random() < 0.2:
        reads += ["a" * int(rng.integers(k, 400))] * int(rng.integers(1, 50))
    if rng.random() < 0.2:
        reads += ["", "acg", "n" * 30]
    order = rng.permutation(len(reads))
    return [reads[i] for i in order]


def pack_reads(bases, boff):
    """the 2-bit form fk_push_packed takes, from a DATA_BLOCK: (codes, nbases, rlen, inv)"""
    code = np.full(256, 255, dtype=np.uint8)
    for i, c in enumerate(b"acgt"):
        code[c] = i
        code[c - 32] = i
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
    flat = np.where(flat == 255, 2, flat).astype(np.uint8)       # (what stands under an invalid base does not matter)
    nb = len(flat)
    pad = np.zeros((nb + 3) // 4 * 4, dtype=np.uint8)
    pad[:nb] = flat
    q = pad.reshape(-1, 4)
    return ((q[:, 0] << 6) | (q[:, 1] << 4) | (q[:, 2] << 2) | q[:, 3]).astype(np.uint8), nb, rlen, inv


def run(iters, seed, only=None, first=None, guard=True, quiet=False, max_bytes=None, budget_s=None):
    """iters iterations of the draw sequence of `seed` (only / first: see the module text).  guard: the reads of every
    iteration in a Guarded mapping.  max_bytes: iterations whose reads take more are drawn but not run (the bounded
    pytest leg).  budget_s: stop after that many seconds.  Returns the number of iterations that ran on the GPU."""
    rng = np.random.default_rng(seed)
    if first is None:
        first = only if only is not None else 0
    t0 = time.time()
    ran = 0
    can = Canaries(seed=seed) if os.environ.get("FUZZ_CANARY", "1") != "0" else None
    for it in range(iters):
        if can is not None:
            can.turn("seed %d before iteration %d" % (seed, it))
        if budget_s is not None and time.time() - t0 > budget_s:
            break
        dry = only is not None and not (first <= it <= only)
        if only is not None and it > only:
            break
        k = int(rng.choice([12, 15, 16, 17, 21, 25, 31, 32, 33, 40, 47, 48, 51, 55, 63, 64]))
        cutoff = int(rng.choice([1, 1, 2, 3, 4, 6]))
        nb = int(rng.choice([1, 1, 2, 3, 7]))
        reads = make_reads(rng, k)
        bases, boff = orc.block_from_reads(reads)
        if max_bytes is not None and len(bases) > max_bytes:
            dry = True
        g = None
        if guard and not dry:
            g = Guarded(bases)
            bases = g.array
        exp = None if dry else orc.fastk(k, bases, boff, cutoff=cutoff)
        chunk = int(rng.choice([0, 0, max(4096, len(bases) // 5)]))
        spill = chunk > 0 and rng.random() < 0.5
        desc = dict(it=it, k=k, cutoff=cutoff, nb=nb, nreads=len(reads), nbytes=len(bases), chunk=chunk, spill=bool(spill))
        if dry or "FUZZ_ORACLE_ONLY" in os.environ:                  # the draws of this iteration, nothing else (FUZZ_ORACLE_ONLY:
                                                                     # the oracle has run on it -- for runs of the oracle under a sanitizer)
            if g is not None:
                g.release()
            nreads = len(boff) - 1
            nthreads = int(rng.integers(1, 4))
            cuts = sorted(int(x) for x in rng.integers(0, nreads + 1, size=nthreads - 1))
            cuts = [0] + cuts + [nreads]
            cur = cuts[:-1].copy()
            while any(cur[t] < cuts[t + 1] for t in range(nthreads)):
                t = int(rng.integers(0, nthreads))
                if cur[t] >= cuts[t + 1]:
                    continue
                cur[t] = min(cuts[t + 1], cur[t] + int(rng.integers(1, 400)))
            if it % 2 == 1:
                lo = 0
                while lo < nreads:
                    lo = min(nreads, lo + int(rng.integers(1, 600)))
            if it % 3 == 0 and all(len(r) > 0 for r in reads):
                fastq = bool(rng.random() < 0.5)
                width = int(rng.choice([0, 0, 7, 60, 100]))
                tlen = 0
                for i, r in enumerate(reads):
                    if fastq:
                        rng.integers(33, 75, size=len(r), dtype=np.uint8)
                        tlen += len(b"@r%d extra+@ text\n" % i) + 2 * len(r) + 4
                    else:
                        tlen += len(b">r%d ACGT>acgt\n" % i)
                        tlen += (len(r) + (len(r) + width - 1) // width) if width else len(r) + 1
                if rng.random() < 0.3 and not fastq:
                    tlen -= 1
                p0 = 0
                while p0 < tlen:
                    p0 += int(rng.integers(1, max(2, tlen // int(rng.integers(1, 12)))))
            continue
        try:
            with fastk_amd.Context(kmer=k, table_cutoff=cutoff, nbuckets=nb) as ctx:
                if chunk:
                    ctx.debug_set("chunk_bytes", chunk)
                    ctx.debug_set("slab_bytes", 32 << 20)             # (not the 8 GiB a real run's store begins with)
                    if spill:
                        ctx.debug_set("spill_limit", max(4096, len(bases) // 20))
                nreads = len(boff) - 1
                nthreads = int(rng.integers(1, 4))
                cuts = sorted(int(x) for x in rng.integers(0, nreads + 1, size=nthreads - 1))
                cuts = [0] + cuts + [nreads]
                cur = cuts[:-1].copy()
                pushed = []
                while any(cur[t] < cuts[t + 1] for t in range(nthreads)):
                    t = int(rng.integers(0, nthreads))
                    if cur[t] >= cuts[t + 1]:
                        continue
                    hi = min(cuts[t + 1], cur[t] + int(rng.integers(1, 400)))
                    lo = cur[t]
                    ctx.push_block(bases[boff[lo]:boff[hi]], (boff[lo:hi + 1] - boff[lo]).astype(np.int32), tid=t)
                    pushed.append(bases[boff[lo]:boff[hi]])
                    cur[t] = hi
                res = ctx.finish()
                assert res.ninst == exp.ninst, "ninst"
                assert np.array_equal(res.hist, exp.hist), "hist"
                assert res.max_inst == exp.max_inst, "max_inst"
                assert res.ntable == exp.ntable and np.array_equal(res.table, exp.table), "table"
                if cutoff == 1 and not chunk:
                    data, offs = ctx.make_profiles()
                    want = orc.profile_counts(k, bases, boff, exp.table)
                    raw = data.tobytes()
                    if len(offs) != len(want) + 1:
                        print("profile count %d for %d reads; device copy of the reads: %s"
                              % (len(offs) - 1, len(want), device_reads_report(ctx, pushed)), flush=True)
                        data2, offs2 = ctx.make_profiles()          # ... and once more, now
                        print("    a second fk_make_profiles right away: %d profiles; device copy now: %s"
                              % (len(offs2) - 1, device_reads_report(ctx, pushed)), flush=True)
                    assert len(offs) == len(want) + 1, "profile count"
                    for i, x in enumerate(want):
                        assert raw[offs[i]:offs[i + 1]] == orc.profile_encode(x), "profile of read %d" % i
            if it % 2 == 1:
                # the same reads in two bits per base (fk_push_packed), in random pieces, chunked or not
                with fastk_amd.Context(kmer=k, table_cutoff=cutoff, nbuckets=nb) as ctx:
                    if chunk:
                        ctx.debug_set("chunk_bytes", chunk)
                        ctx.debug_set("slab_bytes", 32 << 20)
                    nreads = len(boff) - 1
                    lo = 0
                    while lo < nreads:
                        hi = min(nreads, lo + int(rng.integers(1, 600)))
                        codes, nbp, rlen, inv = pack_reads(bases[boff[lo]:boff[hi]], boff[lo:hi + 1] - boff[lo])
                        ctx.push_packed(codes, nbp, rlen, inv)
                        lo = hi
                    res = ctx.finish()
                    assert res.ninst == exp.ninst and np.array_equal(res.hist, exp.hist) and res.max_inst == exp.max_inst, \
                        "packed push: histogram"
                    assert res.ntable == exp.ntable and np.array_equal(res.table, exp.table), "packed push: table"
            if cutoff == 1 and it % 2 == 0:
                # the sharded profile pieces on one context: split with positions, count the records,
                # owner-side look-ups, scatter back, codec
                with fastk_amd.Context(kmer=k, table_cutoff=1, nbuckets=1) as ctx:
                    n = len(bases)
                    rbuf = ctx.alloc(n + 64).upload(bases)
                    ns, ni, counts = ctx.split(rbuf.ptr, n)
                    w = ctx.w
                    recs = ctx.alloc(max(ns, 1) * w.smer_stride)
                    keep = ctx.alloc(max(ns, 1) * w.smer_stride)
                    pos = ctx.alloc(max(ns, 1) * 8)
                    if ns:
                        ctx.split_emit_pos(rbuf.ptr, n, recs.ptr, ns, counts, pos.ptr)
                        host = recs.download(ns * w.smer_stride)
                        keep.upload(host)
                    r2 = ctx.count_device_supermers(recs.ptr if ns else None, ns, fetch_table=True)
                    assert np.array_equal(r2.hist, exp.hist) and np.array_equal(r2.table, exp.table), "supermer run"
                    ninst = ctx.profile_lookup_supermers(keep.ptr if ns else None, ns)
                    assert ninst == exp.ninst, "look-up instance count"
                    cbuf = ctx.alloc(max(ninst, 1) * 2)
                    if ns:
                        ctx.profile_lookup_supermers(keep.ptr, ns, cbuf.ptr, ninst)
                    ctx.profile_scatter(keep.ptr if ns else None, pos.ptr if ns else None, ns,
                                        cbuf.ptr if ns else None, n, reset=True)
                    data, offs = ctx.profile_encode(rbuf.ptr, n)
                    want = orc.profile_counts(k, bases, boff, exp.table)
                    raw = data.tobytes()
                    assert len(offs) == len(want) + 1, "exchanged profile count"
                    for i, x in enumerate(want):
                        assert raw[offs[i]:offs[i + 1]] == orc.profile_encode(x), "exchanged profile of read %d" % i
            if it % 4 == 1 and exp.ntable > 0:
                # exact_parts (no random draw: the sequence of the other legs stays what it was): the reference's own
                # super-mer rule, files against the oracle's writers -- hidden .ktab parts included
                import shutil
                import tempfile
                T = 1 + it % 5
                prof = (it % 8 == 5)            # as in a run with -p: the super-mers keep the read's strand (exact_parts = 2)
                ex = orc.fastk(k, bases, boff, cutoff=cutoff, nthreads=T, profile=prof)
                d = tempfile.mkdtemp(prefix="fkfz")
                try:
                    os.mkdir(os.path.join(d, "o"))
                    os.mkdir(os.path.join(d, "g"))
                    orc.write_outputs(ex, cutoff, T, os.path.join(d, "o"), "x")
                    with fastk_amd.Context(kmer=k, table_cutoff=cutoff, nthreads=T, exact_parts=2 if prof else 1) as ctx:
                        if it % 8 == 1:
                            ctx.debug_set("exact_chain", 1)
                        nreads = len(boff) - 1
                        step = max(1, nreads // 3)
                        for lo in range(0, nreads, step):
                            hi = min(nreads, lo + step)
                            ctx.push_block(bases[boff[lo]:boff[hi]], (boff[lo:hi + 1] - boff[lo]).astype(np.int32))
                        res = ctx.finish()
                        assert np.array_equal(res.hist, ex.hist) and np.array_equal(res.table, ex.table), "exact_parts: counts"
                        ctx.write_hist(res, os.path.join(d, "g", "x.hist"))
                        ctx.write_ktab(res, os.path.join(d, "g"), "x")
                    names = sorted(os.listdir(os.path.join(d, "o")))
                    assert names == sorted(os.listdir(os.path.join(d, "g"))), "exact_parts: file names"
                    for f in names:
                        assert open(os.path.join(d, "o", f), "rb").read() == open(os.path.join(d, "g", f), "rb").read(), \
                            "exact_parts: " + f
                finally:
                    shutil.rmtree(d, ignore_errors=True)
            if it % 3 == 0 and all(len(r) > 0 for r in reads):
                # the device text parsers: the same reads as FASTQ / FASTA text, cut at random bytes
                fastq = bool(rng.random() < 0.5)
                width = int(rng.choice([0, 0, 7, 60, 100]))
                parts = []
                for i, r in enumerate(reads):
                    seq = r.encode()
                    if fastq:
                        parts.append(b"@r%d extra+@ text\n" % i + seq + b"\n+\n" +
                                     bytes(rng.integers(33, 75, size=len(seq), dtype=np.uint8)) + b"\n")
                    else:
                        parts.append(b">r%d ACGT>acgt\n" % i)
                        if width:
                            parts.extend(seq[j:j + width] + b"\n" for j in range(0, len(seq), width))
                        else:
                            parts.append(seq + b"\n")
                text = b"".join(parts)
                if rng.random() < 0.3 and not fastq:
                    text = text[:-1]                                 # no newline at the end of the file
                with fastk_amd.Context(kmer=k, table_cutoff=cutoff, nbuckets=nb) as ctx:
                    st, nr, p0 = (0 if fastq else 2), 0, 0
                    while p0 < len(text):
                        n = int(rng.integers(1, max(2, len(text) // int(rng.integers(1, 12)))))
                        if fastq:
                            st, r_, _ = ctx.push_fastq(text[p0:p0 + n], st)
                        else:
                            st, r_, _ = ctx.push_fasta(text[p0:p0 + n], st, last=(p0 + n >= len(text)))
                        nr += r_
                        p0 += n
                    res = ctx.finish()
                    assert nr == len(reads), "parsed read count %d != %d" % (nr, len(reads))
                    assert res.ninst == exp.ninst and np.array_equal(res.hist, exp.hist), "text parser: hist"
                    assert np.array_equal(res.table, exp.table), "text parser: table"
        except Exception as e:                                      # noqa: BLE001
            print("FAILED", dict(desc, seed=seed), repr(e), "; the host array of the reads is",
                  "not guarded" if g is None else ("intact" if g.intact() else "CHANGED since it was filled"), flush=True)
            os.makedirs("gpurun_out", exist_ok=True)
            np.save("gpurun_out/fuzz_fail_bases_%d_%d.npy" % (seed, it), np.array(bases))
            np.save("gpurun_out/fuzz_fail_boff_%d_%d.npy" % (seed, it), boff)
            raise
        if g is not None:
            assert g.intact(), "iteration %d of seed %d: the host array of the reads changed under the run" % (it, seed)
            g.release()
        ran += 1
        if it % 10 == 9 and not quiet:
            print("iteration %d ok (%.1f s)" % (it + 1, time.time() - t0), flush=True)
    return ran


def main():
    faulthandler.enable(all_threads=True)
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    only = int(os.environ["FUZZ_ONLY"]) if "FUZZ_ONLY" in os.environ else None
    first = int(os.environ["FUZZ_FROM"]) if "FUZZ_FROM" in os.environ else None      # (FUZZ_FROM=<m>: iterations m .. n)
    t0 = time.time()
    ran = run(iters, seed, only=only, first=first, guard=os.environ.get("FUZZ_GUARD", "1") != "0")
    print("all %d iterations equal to the oracle (seed %d, %d on the GPU, %.1f s)" % (iters, seed, ran, time.time() - t0))


if __name__ == "__main__":
    main()
