"""The N>1 path on CPU: world_size-2 gloo run of tests.shard_model.count_sharded with a checker
engine built from the oracle (tests may use the oracle; the product engine is HipEngine).
Verifies the exchange logic: bucket -> rank routing, all-to-all-v sizes, histogram all-reduce."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import orc
from tests import util


class OracleEngine:
    """split / count stages on CPU tensors using the oracle; bucket = f(minimizer of the super-mer)
    so that equal canonical k-mers always land in the same bucket (FastK.h:3-7)."""

    def __init__(self, kmer, world, cutoff, rounds=1):
        self.P = orc.params(kmer)
        self.world = world * rounds            # buckets: bucket r * world + d goes to rank d in round r
        self.cutoff = cutoff
        self.stride = self.P.smer_word
        self.pieces = None

    def _bucket(self, rec):
        P = self.P
        n = int(rec[P.smer_bytes]) + 1
        L = n - 1 + P.kmer
        codes = [(int(rec[i >> 2]) >> (6 - 2 * (i & 3))) & 3 for i in range(L)]
        best = None
        for j in range(L - 4):
            f = 0
            r = 0
            for t in range(5):
                f = f * 4 + codes[j + t]
                r = r * 4 + (3 - codes[j + 4 - t])
            v = min(f, r)
            best = v if best is None else min(best, v)
        return (best * 2654435761 >> 7) % self.world

    def split(self, reads):
        bases = reads.numpy()
        ends = np.nonzero(bases == 0)[0]
        boff = np.concatenate([[0], ends + 1]).astype(np.int64)
        recs, ninst = orc.distribute(self.P, bases, boff)
        b = np.array([self._bucket(r) for r in recs], dtype=np.int64)
        order = np.argsort(b, kind="stable")
        counts = np.bincount(b, minlength=self.world).tolist()
        out = torch.from_numpy(np.ascontiguousarray(recs[order]).reshape(-1))
        offs = [0]
        for c in counts[:-1]:
            offs.append(offs[-1] + c)
        return out, counts, offs, ninst

    # the rounds interface of the product engine: pieces are counted one by one and summed; pieces are
    # whole buckets, so their tables are disjoint
    def rounds_begin(self):
        self.pieces = []

    def rounds_add(self, recs, nsuper):
        self.pieces.append(self.count_supermers(recs.clone(), nsuper) if nsuper else None)

    def rounds_finish(self, fetch_table=False):
        P = self.P
        hist = np.zeros(0x8000, dtype=np.int64)
        mx = nw = nd = nt = 0
        tabs = []
        for p in self.pieces:
            if p is None:
                continue
            hist += p["hist"]; mx += p["max_inst"]; nw += p["nweighted"]; nd += p["ndistinct"]; nt += p["ntable"]
            tabs.append(p["result"].table)
        table = np.concatenate(tabs) if tabs else np.zeros((0, P.kmer_bytes + 2), dtype=np.uint8)
        table = table[np.lexsort(table[:, :P.kmer_bytes].T[::-1])] if len(table) else table

        class R:
            pass
        res = R()
        res.table = table
        return dict(hist=hist, max_inst=mx, nweighted=nw, ndistinct=nd, ntable=nt, result=res)

    # ---- checker versions of the stages of shard.profiles_exchanged: one record per valid k-mer
    #      (any cut of the reads into super-mers is a legal input of the counting stages)
    def split_with_positions(self, reads):
        P = self.P
        k = P.kmer
        bases = reads.numpy()
        code = orc._CODE[bases]
        recs, pos = [], []
        n = len(bases)
        valid_run = 0
        for i in range(n):
            valid_run = valid_run + 1 if code[i] < 4 else 0
            if valid_run >= k:
                st = i - k + 1
                rec = np.zeros(P.smer_word, dtype=np.uint8)
                c = code[st:st + k]
                pad = np.zeros(P.smer_bytes * 4, dtype=np.uint8)
                pad[:k] = c
                rec[:P.smer_bytes] = (pad.reshape(-1, 4) * np.array([64, 16, 4, 1], dtype=np.uint8)).sum(axis=1)
                rec[P.smer_bytes] = 0                                  # one k-mer
                recs.append(rec)
                pos.append(st << 1)
        recs = np.array(recs, dtype=np.uint8).reshape(-1, P.smer_word)
        b = np.array([self._bucket(r) for r in recs], dtype=np.int64)
        order = np.argsort(b, kind="stable")
        counts = np.bincount(b, minlength=self.world).tolist()
        offs = [0]
        for c in counts[:-1]:
            offs.append(offs[-1] + c)
        return (torch.from_numpy(np.ascontiguousarray(recs[order]).reshape(-1)), counts, offs, len(recs),
                torch.from_numpy(np.array(pos, dtype=np.int64)[order]))

    def kmers_per_record(self, recs, nsuper):
        return recs[: nsuper * self.stride].view(nsuper, self.stride)[:, self.P.smer_bytes].to(torch.int64) + 1

    def lookup_supermers(self, recs, nsuper):
        P = self.P
        kb = P.kmer_bytes
        tab = self.last_table
        keys = {bytes(r[:kb]): int(r[kb]) | (int(r[kb + 1]) << 8) for r in tab}
        a = recs.numpy().reshape(nsuper, P.smer_word)
        out = np.zeros(nsuper, dtype="<u2")
        comp = np.array([3, 2, 1, 0], dtype=np.uint8)
        wts = np.array([64, 16, 4, 1], dtype=np.uint8)
        for i, r in enumerate(a):
            codes = np.array([(int(r[j >> 2]) >> (6 - 2 * (j & 3))) & 3 for j in range(P.kmer)], dtype=np.uint8)
            rc = comp[codes[::-1]]
            def pack(c):
                pad = np.zeros(kb * 4, dtype=np.uint8)
                pad[:P.kmer] = c
                return bytes((pad.reshape(-1, 4) * wts).sum(axis=1).astype(np.uint8))
            out[i] = keys.get(min(pack(codes), pack(rc)), 0)
        return torch.from_numpy(out.view(np.uint8).copy())

    def scatter_and_encode(self, reads, recs, pos, nsuper, counts):
        bases = reads.numpy()
        c16 = np.zeros(len(bases), dtype=np.uint16)
        got = counts.numpy().view("<u2")
        c16[pos.numpy()[:nsuper] >> 1] = got[:nsuper]
        ends = np.nonzero(bases == 0)[0]
        blobs, start = [], 0
        for e in ends:
            npos = e - start - self.P.kmer + 1
            blobs.append(orc.profile_encode(c16[start:start + npos]) if npos > 0 else b"")
            start = e + 1
        offs = np.concatenate([[0], np.cumsum([len(b) for b in blobs])]).astype(np.int64)
        return np.frombuffer(b"".join(blobs), dtype=np.uint8), offs

    def set_table(self, records):
        self.dictionary = records[np.lexsort(records[:, :self.P.kmer_bytes].T[::-1])]

    def make_profiles(self, reads):
        bases = reads.numpy()
        ends = np.nonzero(bases == 0)[0]
        boff = np.concatenate([[0], ends + 1]).astype(np.int64)
        blobs = [orc.profile_encode(c) for c in orc.profile_counts(self.P.kmer, bases, boff, self.dictionary)]
        offs = np.concatenate([[0], np.cumsum([len(b) for b in blobs])]).astype(np.int64)
        return np.frombuffer(b"".join(blobs), dtype=np.uint8), offs

    def count_supermers(self, recs, nsuper, fetch_table=False):
        P = self.P
        a = recs.numpy().reshape(nsuper, P.smer_word)
        ss = orc.msd_sort(a, P.smer_word)
        kl, ovf, nd = orc.kmer_list(P, ss)
        ks = orc.msd_sort(kl, P.kmer_bytes)
        res = orc.count_sorted(P, ks, self.cutoff)
        self.last_table = res.table
        return dict(hist=res.hist, max_inst=res.max_inst + ovf, nweighted=len(kl),
                    ndistinct=res.ndistinct, ntable=res.ntable, result=res)


def _wfirst(case, bases, boff):
    """first-byte census of the weighted k-mers of the whole data set (what the all-reduced wfirst of
    the product run holds); the checker engine does not track it, so it is recomputed here"""
    P = orc.params(case["k"])
    recs, _ = orc.distribute(P, bases, boff)
    kl, _, _ = orc.kmer_list(P, orc.msd_sort(recs, P.smer_word))
    return np.bincount(kl[:, 0], minlength=256).astype(np.int64)


def _worker(rank, world, port, name, q, outdir, rounds=1):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests import shard_model as shard
    shard.MAX_PAIR_BYTES = 4096       # force the multi-round exchange
    case, bases, boff = util.load_case(name)
    nreads = len(boff) - 1
    lo, hi = rank * nreads // world, (rank + 1) * nreads // world
    mine = torch.from_numpy(bases[boff[lo]:boff[hi]].copy())
    eng = OracleEngine(case["k"], world, case["cutoff"], rounds)
    if rounds == 1:
        out = shard.count_sharded(eng, mine, verify=True)
    else:
        out = shard.count_sharded_rounds(eng, mine, rounds, verify=True)
    P = orc.params(case["k"])

    def cpu_sort(recs):
        return recs[np.lexsort(recs[:, :P.kmer_bytes].T[::-1])]

    merged = shard.gather_table(out["local"]["result"].table, P.kmer_bytes, cpu_sort)
    # the same table written without the gather: every rank writes the parts of its first-byte range
    shard.write_table_sharded(out["local"]["result"].table, _wfirst(case, bases, boff), out["ntable"], case["k"],
                              case["cutoff"], 2, outdir, "y", cpu_sort)
    if rank == 0:
        import fastk_amd
        fastk_amd.write_files(case["k"], case["cutoff"], case["T"], out["hist"], out["max_inst"],
                              merged, outdir, "x")
        q.put(dict(hist=out["hist"], max_inst=out["max_inst"], ninst=out["ninst"],
                   ntable=out["ntable"], ndistinct=out["ndistinct"], merged=merged))
    dist.destroy_process_group()


@pytest.mark.parametrize("rounds", [1, 3])
@pytest.mark.parametrize("name", ["synth_tiny_k40_t1_T2", "edge_k21_t2_T3"])
def test_two_rank_shard_matches_golden(name, rounds, tmp_path):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, q, str(tmp_path), rounds))
             for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    case, bases, boff = util.load_case(name)
    exp = orc.fastk(case["k"], bases, boff, cutoff=case["cutoff"])
    assert got["ninst"] == exp.ninst
    assert got["ndistinct"] == exp.ndistinct and got["ntable"] == exp.ntable
    # ranks hold disjoint k-mer sets; the gathered, re-ordered union is the reference table, and the
    # files rank 0 wrote with the library's writers carry the reference's bytes
    util.check_against_golden(case, got["hist"], got["max_inst"], got["merged"])
    import hashlib
    exp = case["expected"]
    assert hashlib.sha256(open(tmp_path / "x.hist", "rb").read()).hexdigest() == exp["hist_sha256"]
    t = orc.read_ktab(str(tmp_path / "x"))
    assert t["stream_sha256"] == exp["ktab"]["stream_sha256"]
    assert (t["nparts"], t["minval"], t["ibytes"], t["nels"]) == \
        (case["T"], case["cutoff"], exp["ktab"]["ibytes"], exp["ktab"]["nels"])
    # written rank by rank (2 ranks x 2 parts): the same canonical stream, and byte for byte the files
    # the one-process writer makes for 4 parts from the merged table
    y = orc.read_ktab(str(tmp_path / "y"))
    assert y["stream_sha256"] == exp["ktab"]["stream_sha256"] and y["nparts"] == 4
    import fastk_amd
    fastk_amd.write_files(case["k"], case["cutoff"], 4, got["hist"], got["max_inst"], got["merged"],
                          str(tmp_path), "z", wfirst=_wfirst(case, bases, boff))
    for f in ["%s.ktab"] + [".%%s.ktab.%d" % (i + 1) for i in range(4)]:
        assert open(tmp_path / (f % "y"), "rb").read() == open(tmp_path / (f % "z"), "rb").read(), f


def _prof_worker(rank, world, port, name, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests import shard_model as shard
    shard.MAX_PAIR_BYTES = 4096
    case, bases, boff = util.load_case(name)
    nreads = len(boff) - 1
    lo, hi = rank * nreads // world, (rank + 1) * nreads // world
    mine = torch.from_numpy(bases[boff[lo]:boff[hi]].copy())
    eng = OracleEngine(case["k"], world, 1)
    out = shard.count_sharded(eng, mine, verify=True)
    data, offs = shard.profiles_sharded(eng, mine, out["local"]["result"].table)
    q.put((rank, lo, hi, data.tobytes(), offs))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_profiles_match_whole_data_oracle():
    """Every rank's profiles (its own reads, union of all ranks' tables as dictionary) are the profiles
    of those reads in the whole data set."""
    name, world = "synth_tiny_k40_t1_T2", 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30500 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_prof_worker, args=(r, world, port, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=300) for _ in range(world)])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    case, bases, boff = util.load_case(name)
    k = case["k"]
    want = orc.profile_counts(k, bases, boff, orc.fastk(k, bases, boff, cutoff=1).table)
    allc = []
    for rank, lo, hi, raw, offs in got:
        assert len(offs) == hi - lo + 1
        for i in range(hi - lo):
            assert raw[offs[i]:offs[i + 1]] == orc.profile_encode(want[lo + i])
            allc.append(orc.profile_decode(raw[offs[i]:offs[i + 1]]))
    assert orc.profiles_digest(allc) == case["expected"]["prof"]["decoded_sha256"]   # the reference's


def _xprof_worker(rank, world, port, name, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests import shard_model as shard
    shard.MAX_PAIR_BYTES = 4096
    case, bases, boff = util.load_case(name)
    nreads = len(boff) - 1
    lo, hi = rank * nreads // world, (rank + 1) * nreads // world
    mine = torch.from_numpy(bases[boff[lo]:boff[hi]].copy())
    eng = OracleEngine(case["k"], world, 1)
    tot, data, offs = shard.profiles_exchanged(eng, mine)
    q.put((rank, lo, hi, data.tobytes(), offs, tot["hist"], tot["max_inst"], tot["ninst"]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_profiles_with_owner_side_lookups():
    """shard.profiles_exchanged over two ranks: records to their owners, counts back to the readers.
    Every rank's profiles are those of its reads in the whole data set; totals are the golden ones."""
    name, world = "synth_tiny_k40_t1_T2", 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_xprof_worker, args=(r, world, port, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=600) for _ in range(world)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    case, bases, boff = util.load_case(name)
    exp = orc.fastk(case["k"], bases, boff, cutoff=1)
    allc = []
    for rank, lo, hi, raw, offs, hist, mx, ninst in got:
        assert len(offs) == hi - lo + 1
        assert np.array_equal(hist, exp.hist) and mx == exp.max_inst and ninst == exp.ninst
        allc += [orc.profile_decode(raw[offs[i]:offs[i + 1]]) for i in range(hi - lo)]
    assert orc.profiles_digest(allc) == case["expected"]["prof"]["decoded_sha256"]
