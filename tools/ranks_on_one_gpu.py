#!/usr/bin/env python3
"""Two (or more) RCCL ranks on ONE GPU: the sharded path with a real exchange between processes.

A box with a single MI355X cannot run `bench.py --gpus 2`: RCCL refuses two ranks on one device
when both report the same host.  Giving every rank its own NCCL_HOSTID makes RCCL treat them as
two hosts and move the payload over its socket transport (loopback) -- slow, but it is the same
torch.distributed code path (all_to_all_single of the counts, batched isend/irecv of the records,
all_reduce of the histogram) that the 2/4/8-GPU runs take over xGMI, with the HIP stages on both
sides.  Every rank counts its stripe of synthetic reads through shard.count_sharded and
shard.count_sharded_rounds; rank 0 also counts ALL reads in one plain context and compares
histogram, totals and the gathered table bit for bit.

Also checked: the .ktab files the ranks write themselves (shard.write_table_sharded) and each rank's
read profiles (shard.profiles_sharded, shard.profiles_exchanged) against the one-context run.

    python tools/ranks_on_one_gpu.py [ranks]    # parent: starts the ranks, exit code 0 = all equal
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORLD = int(os.environ.get("FK_RANKS", "2"))
GENOME = 3_000_000          # bases, all ranks together
READ_LEN = 150
COVER = 20
ERR_PPM = 2000
SEED = 77
ROUNDS = 3


def worker():
    rank = int(os.environ["RANK"])
    import numpy as np
    import torch
    import torch.distributed as dist
    import fastk_amd
    from tests import shard_model as shard

    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=rank, world_size=WORLD, device_id=dev)
    total = COVER * GENOME // READ_LEN // WORLD * WORLD
    per = total // WORLD
    nbytes = per * (READ_LEN + 1)

    def stripe(ctx, first, n):
        buf = torch.empty(n * (READ_LEN + 1) + 64, dtype=torch.uint8, device=dev)
        ctx._ck(ctx.L.fk_synth_reads(ctx.h, SEED, GENOME, READ_LEN, ERR_PPM, first, n, buf.data_ptr()))
        torch.cuda.synchronize()
        return buf

    shared = os.environ["FK_SHARED_DIR"]
    PARTS = 2                                   # hidden .ktab parts written per rank
    outs = {}
    for name, nb in (("one all-to-all", WORLD), ("%d overlapped rounds" % ROUNDS, WORLD * ROUNDS)):
        with fastk_amd.Context(kmer=40, table_cutoff=1, nthreads=4, device=0, nbuckets=nb) as ctx:
            ctx.set_stream(torch.cuda.current_stream().cuda_stream)
            reads = stripe(ctx, rank * per, per)
            eng = shard.HipEngine(ctx, dev)
            eng.train_buckets(reads[:nbytes])
            if nb == WORLD:
                out = shard.count_sharded(eng, reads[:nbytes], verify=True, fetch_table=True)
            else:
                out = shard.count_sharded_rounds(eng, reads[:nbytes], ROUNDS, verify=True, fetch_table=True)
            table = out["local"]["result"].table
            nloc = len(table)
            merged = shard.gather_table(table, ctx.w.kmer_bytes, eng.sort_table)
            outs[name] = (out, merged, nloc)
            if nb == WORLD:
                # every rank writes the .ktab parts of its first-byte range (second exchange) ...
                shard.write_table_sharded(table, out["wfirst"], out["ntable"], 40, 1, PARTS, shared, "y",
                                          eng.sort_table)
                # ... and profiles its own reads against the all-gathered table
                pdata, poffs = shard.profiles_sharded(eng, reads[:nbytes], table)
                np.save(os.path.join(shared, "prof_gathered_%d_data.npy" % rank), pdata)
                np.save(os.path.join(shared, "prof_gathered_%d_offs.npy" % rank), poffs)
    # profiles with the look-ups on the owning ranks: no table replication
    with fastk_amd.Context(kmer=40, table_cutoff=1, nthreads=4, device=0, nbuckets=WORLD) as ctx:
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        reads = stripe(ctx, rank * per, per)
        eng = shard.HipEngine(ctx, dev)
        eng.train_buckets(reads[:nbytes])
        totals, pdata, poffs = shard.profiles_exchanged(eng, reads[:nbytes])
        np.save(os.path.join(shared, "prof_exchanged_%d_data.npy" % rank), np.asarray(pdata))
        np.save(os.path.join(shared, "prof_exchanged_%d_offs.npy" % rank), np.asarray(poffs))
        outs["profiles exchanged"] = (totals, None, None)
    dist.barrier()

    ok = torch.ones(1, dtype=torch.int64, device=dev)
    if rank == 0:
        with fastk_amd.Context(kmer=40, table_cutoff=1, nthreads=4, device=0) as ctx:
            allr = stripe(ctx, 0, total)
            ref = ctx.count_device_reads(allr.data_ptr(), total * (READ_LEN + 1), fetch_table=True)
            fdata, foffs = ctx.make_profiles(allr.data_ptr(), total * (READ_LEN + 1))
            assert len(foffs) == total + 1
            T = WORLD * PARTS
            first = outs["one all-to-all"][0]
            fastk_amd.write_files(40, 1, T, ref.hist, ref.max_inst, ref.table, shared, "x", wfirst=first["wfirst"])
            try:
                for f in ["%s.ktab"] + [".%%s.ktab.%d" % (i + 1) for i in range(T)]:
                    a = open(os.path.join(shared, f % "y"), "rb").read()
                    b = open(os.path.join(shared, f % "x"), "rb").read()
                    assert a == b, ("table file written by the ranks differs", f)
                for kind in ("gathered", "exchanged"):
                    for r in range(WORLD):
                        d = np.load(os.path.join(shared, "prof_%s_%d_data.npy" % (kind, r)))
                        o = np.load(os.path.join(shared, "prof_%s_%d_offs.npy" % (kind, r)))
                        lo, hi = int(foffs[r * per]), int(foffs[(r + 1) * per])
                        assert len(o) == per + 1 and int(o[-1]) == hi - lo, ("profile sizes", kind, r)
                        assert np.array_equal(np.asarray(o, dtype=np.int64), foffs[r * per:(r + 1) * per + 1] - lo), \
                            ("profile offsets", kind, r)
                        assert np.array_equal(d[:hi - lo], fdata[lo:hi]), ("profile bytes", kind, r)
                print("%d ranks: .ktab stub + %d parts written by the ranks and the profiles of all %d reads "
                      "(table all-gathered / look-ups on the owners): equal to the one-context run"
                      % (WORLD, T, total), flush=True)
            except AssertionError as e:
                print("%d ranks: MISMATCH %r" % (WORLD, e.args), flush=True)
                ok[0] = 0
            for name, (out, merged, nloc) in outs.items():
                try:
                    assert out["ninst"] == ref.ninst, ("ninst", out["ninst"], ref.ninst)
                    assert np.array_equal(out["hist"][1:], ref.hist[1:]), "histogram"
                    assert out["max_inst"] == ref.max_inst, "max_inst"
                    assert out["ndistinct"] == ref.ndistinct, "distinct"
                    if merged is not None:
                        assert 0 < nloc < len(merged), "rank 0 must own a proper part of the table"
                        assert np.array_equal(merged, ref.table), "gathered table"
                    print("%d ranks, %s: %d k-mer instances, %d distinct: equal to the one-context run"
                          % (WORLD, name, out["ninst"], out["ndistinct"]), flush=True)
                except AssertionError as e:
                    print("%d ranks, %s: MISMATCH %r" % (WORLD, name, e.args), flush=True)
                    ok[0] = 0
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    dist.destroy_process_group()
    sys.exit(0 if int(ok.item()) == 1 else 1)


def parent(world=WORLD, port=29641, timeout=420):
    import tempfile
    shared = tempfile.mkdtemp(prefix="fk_ranks")
    try:
        return _run_ranks(world, port, timeout, shared)
    finally:
        subprocess.run(["rm", "-rf", shared])


def _run_ranks(WORLD, port, timeout, shared):
    procs = []
    for r in range(WORLD):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE=str(WORLD), FK_RANKS=str(WORLD),
                   FK_SHARED_DIR=shared,
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   NCCL_HOSTID="fk-rank-%d" % r, NCCL_SOCKET_IFNAME="lo", NCCL_IB_DISABLE="1",
                   HSA_ENABLE_IPC_MODE_LEGACY="0", FK_RANK_WORKER="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    rc, text = 0, []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
            o += "\n[timed out]"
            rc = rc or 124
        text.append(o)
        rc = rc or p.returncode
    return rc, "\n".join(text)


if __name__ == "__main__":
    if os.environ.get("FK_RANK_WORKER") == "1":
        worker()
    else:
        rc, text = parent(int(sys.argv[1]) if len(sys.argv) > 1 else WORLD)
        print(text)
        sys.exit(rc)
