"""TEST INFRASTRUCTURE: a model of the sharded run's protocol (C1 bucket exchange, C2 all-reduce, C3 final gather, the two
profile routes) over torch.distributed, behind a small engine interface -- so that the exchange logic runs on CPU with
gloo and a checker engine built from the oracle (tests/test_shard_gloo.py, world size 2), and on RCCL ranks sharing
one GPU with the HIP stages (tools/ranks_on_one_gpu.py).  The PRODUCT's implementation of C1-C3 is the C engine,
fastk_amd/csrc/fk_shard.hip (fk_shard_*: RCCL called from C) -- what `FastK_amd -G<n>` and `bench.py --gpus N` run; until
round 5 this module lived in the package (fastk_amd/shard.py) and bench.py's --config 1 ran through it.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  Every rank splits its
own stripe of reads into super-mers grouped by bucket; bucket b belongs to rank b (a bucket is a
function of the canonical minimizer, so equal k-mers always meet on one rank, FastK.h:3-7); one
all-to-all-v of SMER_WORD records replaces the reference's ".T" file shuffle (split.c:1263 <->
count.c:1347); each rank then sorts, expands, sorts and counts its records with no further
communication; the 0x8000-bin histogram and the scalar totals are all-reduced (count.c:1543-1553).

The compute stages sit behind a small engine interface so that the exchange logic can be
exercised on CPU (gloo) in tests with a checker engine; the product engine is HipEngine, which
calls libfastk_amd.so and has no CPU path.
"""
import numpy as np
import torch
import torch.distributed as dist

from fastk_amd.api import Context, HIST_BINS


class HipEngine:
    """Stages on torch CUDA(=HIP) tensors through the C-ABI."""

    def __init__(self, ctx: Context, device):
        self.ctx = ctx
        self.device = device
        self.stride = ctx.w.smer_stride
        # The library launches on the context's stream, torch and RCCL order their work against torch's
        # current stream: bind the two, so that an exchange (w.wait() only orders torch's current
        # stream) is complete before a stage kernel reads the inbox, and a stage is complete before
        # torch reuses its output.  Contract for callers: keep this device's current stream unchanged
        # for the life of the engine, or call bind_stream() again after switching.
        self.bind_stream()

    def bind_stream(self):
        if isinstance(self.device, torch.device) and self.device.type == "cuda":
            self.ctx.set_stream(torch.cuda.current_stream(self.device).cuda_stream)

    def train_buckets(self, reads, group=None, sample_bytes=2 << 20):
        """Balance the buckets (= ranks) on a sample of the reads: every rank takes a census of the
        head of its stripe, the censuses are all-reduced, and every rank derives the same minimizer ->
        bucket assignment from the sum (fk_set_bucket_weights)."""
        sample = reads[:sample_bytes].cpu().numpy()
        counts = torch.from_numpy(self.ctx.bucket_census(sample)).to(self.device)
        if dist.is_initialized():
            dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
        self.ctx.set_bucket_weights(counts.cpu().numpy())

    def split(self, reads):
        """reads: uint8 tensor in HBM.  Returns (records uint8 tensor, per-bucket counts, per-bucket
        record offsets into that tensor, ninst).  One emit pass into sampled, padded regions; the
        exact count-then-emit pair is the fallback when a region overflows."""
        n = reads.numel()
        cap, offs = self.ctx.split_plan(reads.data_ptr(), n)
        recs = torch.empty(max(cap, 1) * self.stride, dtype=torch.uint8, device=self.device)
        got = self.ctx.split_planned(reads.data_ptr(), n, recs.data_ptr(), cap, offs) if cap else ([0] * (len(offs) - 1), 0)
        if got is not None:
            counts, ni = got
            return recs, counts, offs[:-1], ni
        ns, ni, counts = self.ctx.split(reads.data_ptr(), n)
        recs = torch.empty(max(ns, 1) * self.stride, dtype=torch.uint8, device=self.device)
        if ns:
            self.ctx.split_emit(reads.data_ptr(), n, recs.data_ptr(), ns, counts)
        offs = [0]
        for c in counts[:-1]:
            offs.append(offs[-1] + c)
        return recs, counts, offs, ni

    def sort_table(self, recs):
        """(n, KMER_WORD) host records -> sorted on the k-mer bytes with the device radix engine."""
        n, kw = recs.shape
        if n == 0:
            return recs
        stride = self.ctx.w.kmer_stride
        dev = np.zeros((n, stride), dtype=np.uint8)
        dev[:, :self.ctx.w.kmer_bytes] = recs[:, :self.ctx.w.kmer_bytes]
        dev[:, stride - 2:] = recs[:, kw - 2:]
        a = self.ctx.alloc(dev.nbytes).upload(dev)
        b = self.ctx.alloc(dev.nbytes)
        p = self.ctx.msd_sort(a.ptr, b.ptr, n, stride, self.ctx.w.kmer_bytes)
        out = a.download(dev.nbytes, ptr=p).reshape(n, stride)
        a.free(); b.free()
        res = np.empty((n, kw), dtype=np.uint8)
        res[:, :kw - 2] = out[:, :kw - 2]
        res[:, kw - 2:] = out[:, stride - 2:]
        return res

    def set_table(self, records):
        """(n, KMER_WORD) host records, any order -> dictionary of make_profiles (fk_set_table)."""
        self.ctx.set_table(records)

    def make_profiles(self, reads):
        """reads: uint8 tensor in HBM (0-terminated reads) -> (codec bytes, nreads + 1 offsets)."""
        return self.ctx.make_profiles(reads.data_ptr(), reads.numel())

    # ---- profiles with the look-ups on the owning rank (profiles_exchanged) ----
    def split_with_positions(self, reads):
        """Exact count-then-emit split that also returns, per record, (position << 1) | flip (int64
        tensor in HBM, same order as the records).  Buckets are contiguous, no padding."""
        n = reads.numel()
        ns, ni, counts = self.ctx.split(reads.data_ptr(), n)
        recs = torch.empty(max(ns, 1) * self.stride, dtype=torch.uint8, device=self.device)
        pos = torch.empty(max(ns, 1), dtype=torch.int64, device=self.device)
        if ns:
            self.ctx.split_emit_pos(reads.data_ptr(), n, recs.data_ptr(), ns, counts, pos.data_ptr())
        offs = [0]
        for c in counts[:-1]:
            offs.append(offs[-1] + c)
        return recs, counts, offs, ni, pos

    def kmers_per_record(self, recs, nsuper):
        """int64 tensor: k-mers of each record (its length byte + 1)."""
        col = recs[: nsuper * self.stride].view(nsuper, self.stride)[:, self.ctx.w.smer_bytes]
        return col.to(torch.int64) + 1

    def lookup_supermers(self, recs, nsuper):
        """Counts (uint16, as a uint8 tensor of 2 bytes each) of the k-mers of the records, record after
        record, from the table this context holds after counting."""
        ninst = self.ctx.profile_lookup_supermers(recs.data_ptr() if nsuper else None, nsuper)
        out = torch.empty(max(ninst, 1) * 2, dtype=torch.uint8, device=self.device)
        if nsuper:
            self.ctx.profile_lookup_supermers(recs.data_ptr(), nsuper, out.data_ptr(), ninst)
        return out[: ninst * 2]

    def scatter_and_encode(self, reads, recs, pos, nsuper, counts):
        n = reads.numel()
        self.ctx.profile_scatter(recs.data_ptr() if nsuper else None, pos.data_ptr() if nsuper else None, nsuper,
                                 counts.data_ptr() if nsuper else None, n, reset=True)
        return self.ctx.profile_encode(reads.data_ptr(), n)

    def rounds_begin(self):
        self.ctx.rounds_begin()

    def rounds_add(self, recs, nsuper):
        self.ctx.rounds_add(recs.data_ptr() if nsuper else None, nsuper)

    def rounds_finish(self, fetch_table=False):
        res = self.ctx.rounds_finish(fetch_table=fetch_table)
        return dict(hist=res.hist, max_inst=res.max_inst, nweighted=res.nweighted,
                    ndistinct=res.ndistinct, ntable=res.ntable, wfirst=res.wfirst, result=res)

    def count_supermers(self, recs, nsuper, fetch_table=False):
        res = self.ctx.count_device_supermers(recs.data_ptr() if nsuper else None, nsuper,
                                              fetch_table=fetch_table)
        return dict(hist=res.hist, max_inst=res.max_inst, nweighted=res.nweighted,
                    ndistinct=res.ndistinct, ntable=res.ntable, wfirst=res.wfirst, result=res)


# One all_to_all_single call moves at most this many bytes between any pair of ranks: element counts
# beyond 2^31 are not safe in every layer underneath (observed: a 4.7 GB self-exchange lost records).
MAX_PAIR_BYTES = 1 << 30


def _exchange_records(recs, inbox, send_n, recv_n, stride, group, s_off=None):
    """all-to-all-v of fixed-width records in rounds of bounded size.  recs holds the outgoing records
    grouped by destination rank, inbox receives them grouped by source rank.  A rank's own bucket
    never enters the collective: it is a device-to-device copy (RCCL moves a self-exchange at a
    fraction of the copy rate)."""
    world = len(send_n)
    me = dist.get_rank(group)
    dev = recs.device
    if s_off is None:
        s_off = [0] * world
        for i in range(1, world):
            s_off[i] = s_off[i - 1] + send_n[i - 1]
    r_off = [0] * world
    for i in range(1, world):
        r_off[i] = r_off[i - 1] + recv_n[i - 1]
    assert send_n[me] == recv_n[me]
    if send_n[me]:
        inbox[r_off[me] * stride:(r_off[me] + recv_n[me]) * stride].copy_(
            recs[s_off[me] * stride:(s_off[me] + send_n[me]) * stride])
    if world == 1:
        return
    # every other pair: point-to-point sends and receives straight between the bucket regions and the
    # inbox (no staging copies), batched so that RCCL runs them as one grouped all-to-all; a pair
    # moves at most MAX_PAIR_BYTES per batch
    per = max(1, MAX_PAIR_BYTES // stride)                     # records per pair and round
    most = torch.tensor([max([send_n[d] for d in range(world) if d != me] + [0])], dtype=torch.int64,
                        device=dev)
    dist.all_reduce(most, op=dist.ReduceOp.MAX, group=group)
    rounds = max(1, -(-int(most.item()) // per))
    for r in range(rounds):
        ops = []
        for k in range(1, world):
            to, frm = (me + k) % world, (me - k) % world
            sl = max(0, min(per, send_n[to] - r * per))
            rl = max(0, min(per, recv_n[frm] - r * per))
            if sl:
                ops.append(dist.P2POp(dist.isend, recs[(s_off[to] + r * per) * stride:
                                                       (s_off[to] + r * per + sl) * stride],
                                      dist.get_global_rank(group, to) if group is not None else to, group))
            if rl:
                ops.append(dist.P2POp(dist.irecv, inbox[(r_off[frm] + r * per) * stride:
                                                        (r_off[frm] + r * per + rl) * stride],
                                      dist.get_global_rank(group, frm) if group is not None else frm, group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()


def count_sharded(engine, reads, group=None, verify=False, fetch_table=False):
    """Run the sharded path; every rank returns the same global totals.

    verify=True adds a checksum of the exchanged payload (used by bench.py's warm-up steps).

    Returns dict(hist int64[0x8000], max_inst, ninst, nsuper, nweighted, ndistinct, ntable,
    wfirst int64[256], local=<this rank's engine result>)."""
    world = dist.get_world_size(group)
    stride = engine.stride
    recs, counts, s_off, ninst = engine.split(reads)
    assert len(counts) == world, "context must be created with nbuckets == world size"
    dev = recs.device

    send = torch.tensor(counts, dtype=torch.int64, device=dev)
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send, group=group)
    send_n = [int(c) for c in counts]
    recv_n = [int(c) for c in recv.tolist()]
    nrecv = sum(recv_n)
    inbox = torch.empty(max(nrecv, 1) * stride, dtype=torch.uint8, device=dev)[: nrecv * stride]
    _exchange_records(recs, inbox, send_n, recv_n, stride, group, s_off)
    if verify:
        # payload check: the byte sum of everything sent equals the byte sum of everything received
        sent = sum(recs[o * stride:(o + c) * stride].sum(dtype=torch.int64)
                   for o, c in zip(s_off, send_n)) + torch.zeros((), dtype=torch.int64, device=dev)
        chk = torch.stack([sent, inbox.sum(dtype=torch.int64)])
        dist.all_reduce(chk, op=dist.ReduceOp.SUM, group=group)
        if int(chk[0].item()) != int(chk[1].item()):
            raise RuntimeError("super-mer exchange corrupted the payload (checksums differ)")
    del recs

    loc = engine.count_supermers(inbox, nrecv, fetch_table) if fetch_table \
        else engine.count_supermers(inbox, nrecv)

    tot = torch.zeros(HIST_BINS + 8 + 256, dtype=torch.int64, device=dev)
    tot[:HIST_BINS] = torch.from_numpy(np.asarray(loc["hist"], dtype=np.int64)).to(dev)
    extra = [loc["max_inst"], ninst, nrecv, loc["nweighted"], loc["ndistinct"], loc["ntable"],
             sum(send_n)]
    tot[HIST_BINS:HIST_BINS + 7] = torch.tensor(extra, dtype=torch.int64, device=dev)
    if loc.get("wfirst") is not None:       # first-byte census of the weighted k-mers (part cuts)
        tot[HIST_BINS + 8:] = torch.from_numpy(np.asarray(loc["wfirst"], dtype=np.int64)).to(dev)
    dist.all_reduce(tot, op=dist.ReduceOp.SUM, group=group)
    t = tot.cpu().numpy()
    if int(t[HIST_BINS + 2]) != int(t[HIST_BINS + 6]):
        raise RuntimeError("super-mer exchange lost records: %d sent, %d received"
                           % (int(t[HIST_BINS + 6]), int(t[HIST_BINS + 2])))
    return dict(hist=t[:HIST_BINS].copy(), max_inst=int(t[HIST_BINS]), ninst=int(t[HIST_BINS + 1]),
                nsuper=int(t[HIST_BINS + 2]), nweighted=int(t[HIST_BINS + 3]),
                ndistinct=int(t[HIST_BINS + 4]), ntable=int(t[HIST_BINS + 5]),
                wfirst=t[HIST_BINS + 8:].copy(), local=loc)


def gather_table(table, kmer_bytes, sort_fn, group=None, dst=0):
    """Final gather (the role of Merge_Tables, table.c:346): every rank holds a sorted table of the
    k-mers of ITS buckets ((n, KMER_WORD) uint8, disjoint between ranks); rank dst receives all of
    them and orders the union.  sort_fn(records ndarray) -> records sorted on the first kmer_bytes
    bytes (HipEngine.sort_table on the GPU; tests pass a CPU sorter).  Returns the merged table on
    dst, None elsewhere."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" \
        else torch.device("cpu")
    kw = table.shape[1]
    sizes = torch.zeros(world, dtype=torch.int64, device=dev)
    sizes[rank] = table.shape[0]
    dist.all_reduce(sizes, op=dist.ReduceOp.SUM, group=group)
    sizes = [int(x) for x in sizes.tolist()]
    mine = torch.from_numpy(np.ascontiguousarray(table).reshape(-1)).to(dev)
    per = max(1, MAX_PAIR_BYTES // kw) * kw
    gdst = dist.get_global_rank(group, dst) if group is not None else dst
    if rank != dst:
        for o in range(0, mine.numel(), per):
            dist.send(mine[o:o + per], gdst, group=group)
        return None
    parts = []
    for r in range(world):
        if r == dst:
            parts.append(mine)
            continue
        buf = torch.empty(sizes[r] * kw, dtype=torch.uint8, device=dev)
        gsrc = dist.get_global_rank(group, r) if group is not None else r
        for o in range(0, buf.numel(), per):
            dist.recv(buf[o:o + per], gsrc, group=group)
        parts.append(buf)
    merged = torch.cat(parts).cpu().numpy().reshape(-1, kw)
    return sort_fn(merged) if world > 1 else merged


def write_table_sharded(table, wfirst, ntable, kmer, cutoff, parts_per_rank, outdir, root, sort_fn,
                        group=None):
    """Final step without a gather (SURVEY 8e, C3 first option): a second all-to-all-v, keyed by the
    first k-mer byte, gives rank r the k-mers of a contiguous first-byte range -- the ranges of
    Table_Split over world x parts_per_rank parts, from the all-reduced weighted census wfirst -- and
    every rank writes the hidden part files of its range itself; rank 0 adds the stub from the summed
    prefix counts.  table: this rank's sorted (n, KMER_WORD) records (its buckets' k-mers); ntable: the
    global entry count (fixes the index width).  sort_fn orders the received runs (HipEngine.sort_table).
    The files are what fk_write_ktab writes from the merged table with nthreads = world x
    parts_per_rank.  outdir must be shared by the ranks."""
    from fastk_amd import api
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" \
        else torch.device("cpu")
    kw = table.shape[1]
    nparts = world * parts_per_rank
    split = api.ktab_split(wfirst, kmer, nparts)
    ib = api.ktab_idx_bytes(kmer, ntable)
    # destination r owns first bytes [split[r*m], split[(r+1)*m])
    first = table[:, 0] if table.shape[0] else np.zeros(0, dtype=np.uint8)
    cuts = [int(np.searchsorted(first, min(split[r * parts_per_rank], 256), side="left")) if r else 0
            for r in range(world)] + [table.shape[0]]
    send_n = [cuts[r + 1] - cuts[r] for r in range(world)]
    send = torch.tensor(send_n, dtype=torch.int64, device=dev)
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send, group=group)
    recv_n = [int(x) for x in recv.tolist()]
    out = torch.from_numpy(np.ascontiguousarray(table).reshape(-1)).to(dev)
    inbox = torch.empty(max(sum(recv_n), 1) * kw, dtype=torch.uint8, device=dev)[: sum(recv_n) * kw]
    _exchange_records(out, inbox, send_n, recv_n, kw, group)
    mine = inbox.cpu().numpy().reshape(-1, kw)
    if world > 1 and mine.shape[0]:
        mine = sort_fn(mine)                       # world sorted runs -> one
    cnt = api.write_ktab_range(mine, kmer, ib, split, rank * parts_per_rank, parts_per_rank, outdir, root)
    written = int(cnt.sum())                        # (before the reduce: on CPU `tot` shares cnt's memory)
    tot = torch.from_numpy(cnt).to(dev)
    dist.reduce(tot, dist.get_global_rank(group, 0) if group is not None else 0, op=dist.ReduceOp.SUM,
                group=group)
    # conservation: the prefix counts of what the ranks wrote must add up to the global entry count
    # (a collective that drops or duplicates entries must never produce a table that merely looks fine)
    chk = torch.tensor([written, table.shape[0]], dtype=torch.int64, device=dev)
    dist.all_reduce(chk, op=dist.ReduceOp.SUM, group=group)
    if int(chk[0]) != int(ntable) or int(chk[1]) != int(ntable):
        raise RuntimeError("table exchange lost entries: %d written, %d held by the ranks, %d counted"
                           % (int(chk[0]), int(chk[1]), int(ntable)))
    if rank == 0:
        api.write_ktab_stub(kmer, nparts, cutoff, ib, tot.cpu().numpy(), outdir, root)
    dist.barrier(group=group)
    return mine.shape[0]


def allgather_table(table, group=None):
    """Every rank's (n, KMER_WORD) table to every rank: the concatenation in rank order (disjoint k-mer
    sets, so it is the whole data set's table up to order).  One broadcast per source rank, at most
    MAX_PAIR_BYTES per operation."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" \
        else torch.device("cpu")
    kw = table.shape[1]
    sizes = torch.zeros(world, dtype=torch.int64, device=dev)
    sizes[rank] = table.shape[0]
    dist.all_reduce(sizes, op=dist.ReduceOp.SUM, group=group)
    sizes = [int(x) for x in sizes.tolist()]
    out = torch.empty(sum(sizes) * kw, dtype=torch.uint8, device=dev)
    per = max(1, MAX_PAIR_BYTES // kw) * kw
    o = 0
    for r in range(world):
        seg = out[o:o + sizes[r] * kw]
        if r == rank and sizes[r]:
            seg.copy_(torch.from_numpy(np.ascontiguousarray(table).reshape(-1)))
        src = dist.get_global_rank(group, r) if group is not None else r
        for c in range(0, seg.numel(), per):
            dist.broadcast(seg[c:c + per], src, group=group)
        o += sizes[r] * kw
    # conservation: every rank must now hold the bytes every source holds (per source segment the sum of
    # all bytes and of every third byte, compared across ranks by min/max reductions)
    sums = torch.zeros(2 * world, dtype=torch.int64, device=dev)
    o = 0
    for r in range(world):
        seg = out[o:o + sizes[r] * kw]
        sums[2 * r] = seg.sum(dtype=torch.int64)
        sums[2 * r + 1] = seg[::3].sum(dtype=torch.int64)
        o += sizes[r] * kw
    lo, hi = sums.clone(), sums.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    if not torch.equal(lo, hi):
        raise RuntimeError("table all-gather corrupted the payload (per-source checksums differ between ranks)")
    return out.cpu().numpy().reshape(-1, kw)


def profiles_sharded(engine, reads, local_table, group=None):
    """Profiles (FastK -p) of this rank's reads in a sharded run.  A read's k-mers were counted on the
    ranks owning their minimizer buckets, so every rank first receives all tables (cutoff 1; an
    all-gather of KMER_WORD records), installs the union as its dictionary and then looks its own reads
    up locally -- no per-read traffic.  Returns (codec bytes, offsets) for the reads of this rank, in
    their order; rank r's reads follow rank r-1's in the data set, so the .pidx/.prof part files can be
    written per rank."""
    union = allgather_table(local_table, group) if dist.is_initialized() and dist.get_world_size(group) > 1 \
        else local_table
    engine.set_table(union)
    return engine.make_profiles(reads)


def profiles_exchanged(engine, reads, group=None, fetch_table=False):
    """Counting + profiles of this rank's reads with no table replication (scales with the data set):
    every super-mer record goes to the rank owning its bucket as in count_sharded, the sender keeps the
    position it was cut from; after counting, the owner looks the k-mers of every record it received up
    in its own table and the counts travel back over the same pairs (2 bytes per k-mer instance); the
    sender scatters them to the positions and runs the codec.  The context must have table_cutoff 1 and
    nbuckets == world size.  Returns (count_sharded-style totals dict, codec bytes, offsets); the rank's
    table stays in HBM unless fetch_table."""
    world = dist.get_world_size(group)
    stride = engine.stride
    recs, counts, s_off, ninst, pos = engine.split_with_positions(reads)
    assert len(counts) == world, "context must be created with nbuckets == world size"
    dev = recs.device if hasattr(recs, "device") else torch.device("cpu")
    send_n = [int(c) for c in counts]
    nsent = sum(send_n)
    per_rec = engine.kmers_per_record(recs, nsent)                      # k-mers of every sent record
    inst_to = [int(per_rec[o:o + c].sum().item()) for o, c in zip(s_off, send_n)]
    meta = torch.tensor(send_n + inst_to, dtype=torch.int64, device=dev).view(2, world).t().contiguous().view(-1)
    got = torch.empty_like(meta)
    dist.all_to_all_single(got, meta, group=group)
    got = got.view(world, 2)
    recv_n = [int(x) for x in got[:, 0].tolist()]
    inst_from = [int(x) for x in got[:, 1].tolist()]
    nrecv = sum(recv_n)
    inbox = torch.empty(max(nrecv, 1) * stride, dtype=torch.uint8, device=dev)[: nrecv * stride]
    _exchange_records(recs, inbox, send_n, recv_n, stride, group, s_off)
    kept = inbox.clone()                                                # counting clobbers its input
    loc = engine.count_supermers(inbox, nrecv, fetch_table)
    del inbox
    back = engine.lookup_supermers(kept, nrecv)                         # uint8 view of uint16 counts
    assert back.numel() == 2 * sum(inst_from)
    del kept
    mine = torch.empty(max(sum(inst_to), 1) * 2, dtype=torch.uint8, device=dev)[: sum(inst_to) * 2]
    _exchange_records(back, mine, inst_from, inst_to, 2, group)
    data, offs = engine.scatter_and_encode(reads, recs, pos, nsent, mine)

    tot = torch.zeros(HIST_BINS + 8, dtype=torch.int64, device=dev)
    tot[:HIST_BINS] = torch.from_numpy(np.asarray(loc["hist"], dtype=np.int64)).to(dev)
    tot[HIST_BINS:HIST_BINS + 6] = torch.tensor([loc["max_inst"], ninst, nrecv, loc["nweighted"], loc["ndistinct"],
                                                 loc["ntable"]], dtype=torch.int64, device=dev)
    dist.all_reduce(tot, op=dist.ReduceOp.SUM, group=group)
    tot2 = torch.tensor([nsent, nrecv, sum(inst_to), sum(inst_from)], dtype=torch.int64, device=dev)
    dist.all_reduce(tot2, op=dist.ReduceOp.SUM, group=group)
    t = tot.cpu().numpy()
    totals = dict(hist=t[:HIST_BINS].copy(), max_inst=int(t[HIST_BINS]), ninst=int(t[HIST_BINS + 1]),
                  nsuper=int(t[HIST_BINS + 2]), nweighted=int(t[HIST_BINS + 3]), ndistinct=int(t[HIST_BINS + 4]),
                  ntable=int(t[HIST_BINS + 5]), local=loc)
    # conservation: records and per-k-mer counts sent == received over all ranks, every k-mer instance split
    # off the reads is in the histogram
    if int(tot2[0]) != int(tot2[1]) or int(tot2[2]) != int(tot2[3]):
        raise RuntimeError("profile exchange lost data: %d records sent, %d received; %d counts expected back, %d sent back"
                           % tuple(int(x) for x in tot2.tolist()))
    h = totals["hist"]
    if int((h[1:0x7fff] * np.arange(1, 0x7fff)).sum()) + totals["max_inst"] != totals["ninst"]:
        raise RuntimeError("profile exchange: the histogram does not hold the %d k-mer instances that were split"
                           % totals["ninst"])
    return totals, data, offs


def _post_round(recs, inbox, send_n, recv_n, s_off, stride, group):
    """Start the exchange of one round: own bucket copied locally, every other pair by point-to-point
    operations handed to the backend as one batch.  Returns the work handles (wait on them before the
    inbox is read).  A pair moves at most MAX_PAIR_BYTES per operation."""
    world = len(send_n)
    me = dist.get_rank(group)
    r_off = [0] * world
    for i in range(1, world):
        r_off[i] = r_off[i - 1] + recv_n[i - 1]
    if send_n[me]:
        inbox[r_off[me] * stride:(r_off[me] + recv_n[me]) * stride].copy_(
            recs[s_off[me] * stride:(s_off[me] + send_n[me]) * stride])
    per = max(1, MAX_PAIR_BYTES // stride)
    ops = []
    for k in range(1, world):
        to, frm = (me + k) % world, (me - k) % world
        gto = dist.get_global_rank(group, to) if group is not None else to
        gfrm = dist.get_global_rank(group, frm) if group is not None else frm
        for o in range(0, send_n[to], per):
            n = min(per, send_n[to] - o)
            ops.append(dist.P2POp(dist.isend, recs[(s_off[to] + o) * stride:(s_off[to] + o + n) * stride],
                                  gto, group))
        for o in range(0, recv_n[frm], per):
            n = min(per, recv_n[frm] - o)
            ops.append(dist.P2POp(dist.irecv, inbox[(r_off[frm] + o) * stride:(r_off[frm] + o + n) * stride],
                                  gfrm, group))
    return dist.batch_isend_irecv(ops) if ops else []


def count_sharded_rounds(engine, reads, rounds, group=None, verify=False, fetch_table=False):
    """The sharded path with the exchange cut into `rounds` pieces: the engine's context has
    world * rounds minimizer buckets, bucket r * world + d goes to rank d in round r, and the exchange
    of round r + 1 runs while round r is being counted (the counting never waits for more than the
    first piece).  Same result dictionary as count_sharded."""
    world = dist.get_world_size(group)
    me = dist.get_rank(group)
    stride = engine.stride
    recs, counts, s_off, ninst = engine.split(reads)
    assert len(counts) == world * rounds, "context must be created with nbuckets == world size * rounds"
    dev = recs.device
    # counts[r * world + d]: my records for rank d in round r; every rank learns what it receives
    send = torch.tensor(counts, dtype=torch.int64, device=dev).view(rounds, world).t().contiguous()
    recv = torch.empty_like(send)                                   # recv[frm][r]
    dist.all_to_all_single(recv.view(-1), send.view(-1), group=group)
    recv = recv.cpu().tolist()
    send_n = [[int(counts[r * world + d]) for d in range(world)] for r in range(rounds)]
    send_o = [[int(s_off[r * world + d]) for d in range(world)] for r in range(rounds)]
    recv_n = [[int(recv[f][r]) for f in range(world)] for r in range(rounds)]
    most = max(sum(x) for x in recv_n)
    inbox = [torch.empty(max(most, 1) * stride, dtype=torch.uint8, device=dev) for _ in range(min(2, rounds))]
    engine.rounds_begin()
    works = _post_round(recs, inbox[0], send_n[0], recv_n[0], send_o[0], stride, group)
    nrecv = 0
    chk_sent = torch.zeros((), dtype=torch.int64, device=dev)
    chk_recv = torch.zeros((), dtype=torch.int64, device=dev)
    for r in range(rounds):
        for w in works:
            w.wait()
        box = inbox[r % len(inbox)]
        n_r = sum(recv_n[r])
        if verify:
            chk_recv += box[:n_r * stride].sum(dtype=torch.int64)
            for d in range(world):
                chk_sent += recs[send_o[r][d] * stride:(send_o[r][d] + send_n[r][d]) * stride].sum(dtype=torch.int64)
        if r + 1 < rounds:                       # the next piece travels while this one is counted
            works = _post_round(recs, inbox[(r + 1) % len(inbox)], send_n[r + 1], recv_n[r + 1],
                                send_o[r + 1], stride, group)
        else:
            works = []
        engine.rounds_add(box[:n_r * stride], n_r)
        nrecv += n_r
    loc = engine.rounds_finish(fetch_table)
    if verify:
        chk = torch.stack([chk_sent, chk_recv])
        dist.all_reduce(chk, op=dist.ReduceOp.SUM, group=group)
        if int(chk[0].item()) != int(chk[1].item()):
            raise RuntimeError("super-mer exchange corrupted the payload (checksums differ)")
    del recs

    tot = torch.zeros(HIST_BINS + 8 + 256, dtype=torch.int64, device=dev)
    tot[:HIST_BINS] = torch.from_numpy(np.asarray(loc["hist"], dtype=np.int64)).to(dev)
    extra = [loc["max_inst"], ninst, nrecv, loc["nweighted"], loc["ndistinct"], loc["ntable"],
             sum(int(c) for c in counts)]
    tot[HIST_BINS:HIST_BINS + 7] = torch.tensor(extra, dtype=torch.int64, device=dev)
    if loc.get("wfirst") is not None:
        tot[HIST_BINS + 8:] = torch.from_numpy(np.asarray(loc["wfirst"], dtype=np.int64)).to(dev)
    dist.all_reduce(tot, op=dist.ReduceOp.SUM, group=group)
    t = tot.cpu().numpy()
    if int(t[HIST_BINS + 2]) != int(t[HIST_BINS + 6]):
        raise RuntimeError("super-mer exchange lost records: %d sent, %d received"
                           % (int(t[HIST_BINS + 6]), int(t[HIST_BINS + 2])))
    return dict(hist=t[:HIST_BINS].copy(), max_inst=int(t[HIST_BINS]), ninst=int(t[HIST_BINS + 1]),
                nsuper=int(t[HIST_BINS + 2]), nweighted=int(t[HIST_BINS + 3]),
                ndistinct=int(t[HIST_BINS + 4]), ntable=int(t[HIST_BINS + 5]),
                wfirst=t[HIST_BINS + 8:].copy(), local=loc)
