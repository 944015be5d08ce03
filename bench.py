#!/usr/bin/env python3
"""bench.py -- canonical k-mers/s of the split -> sort -> expand -> sort -> count hot path on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W   (N>1 under torch.distributed.run)
prints ONE JSON line on rank 0.  A "step" is one pass of the whole hot path over one batch of
synthetic reads that is already resident in HBM when the timed region starts.

Workload = BASELINE.json configs[1]: 50x coverage, 150 bp reads, 0.1 % substitutions, of a
100 Mbp uniform random genome, k=40 -t1 (33.3 M reads, ~3.7 G k-mer instances) per GPU.
At N GPUs the genome is N x 100 Mbp (weak scaling: fixed work per GPU); rank r owns read stripe
r; super-mers are exchanged by minimizer bucket with one RCCL all-to-all-v, then each GPU sorts
and counts its buckets; histograms are all-reduced.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--kmer", type=int, default=40)
    ap.add_argument("--genome-mbp", type=float, default=100.0, help="genome size per GPU (Mbp)")
    ap.add_argument("--coverage", type=float, default=50.0)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--err-ppm", type=int, default=1000)
    ap.add_argument("--seed", type=int, default=20251001)
    ap.add_argument("--cpu-sample-mbp", type=float, default=20.0,
                    help="genome size of the bounded CPU-baseline sample (same coverage/shape)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-shard", action="store_true",
                    help="run the sharded (all-to-all) path even on one GPU: exercises RCCL + shard.py")
    ap.add_argument("--stream-buckets", type=int, default=1,
                    help="count the minimizer buckets one after the other (HBM-budgeted mode), N=1 only")
    ap.add_argument("--exchange-rounds", type=int, default=4,
                    help="N > 1: the super-mer exchange is cut into this many pieces; piece i+1 travels "
                         "while piece i is counted (1 = one all-to-all, then count)")
    ap.add_argument("--debug", action="append", default=[], metavar="KEY=VALUE",
                    help="fk_debug_set knob for ablation runs (results may be invalid)")
    return ap.parse_args()


def cpu_baseline(args):
    """Reference FastK (oracle/_ref/FastK, built from the reference sources) on this box's host
    cores, on a bounded sample of the same workload; falls back to the scalar port."""
    from oracle import orc
    cores = os.cpu_count() or 1
    glen = int(args.cpu_sample_mbp * 1e6)
    nreads = int(args.coverage * glen / args.read_len)
    sample = "%gx coverage of a %g Mbp genome, %d x %d bp reads, err %d ppm, k=%d -t1" % (
        args.coverage, args.cpu_sample_mbp, nreads, args.read_len, args.err_ppm, args.kmer)
    bases, boff = orc.synth_block(args.seed, glen, args.read_len, args.err_ppm, 0, nreads)
    inst = nreads * (args.read_len - args.kmer + 1)
    if orc.have_ref():
        d = tempfile.mkdtemp(prefix="fkbase")
        try:
            L = args.read_len
            mat = np.empty((nreads, 3 + L + 3 + L + 1), dtype=np.uint8)
            mat[:, 0:3] = np.frombuffer(b"@r\n", dtype=np.uint8)
            mat[:, 3:3 + L] = bases.reshape(nreads, L + 1)[:, :L]
            mat[:, 3 + L:6 + L] = np.frombuffer(b"\n+\n", dtype=np.uint8)
            mat[:, 6 + L:6 + 2 * L] = ord("I")
            mat[:, 6 + 2 * L] = ord("\n")
            path = os.path.join(d, "s.fastq")
            mat.tofile(path)
            del mat
            cmd = [os.path.join(orc.REF_DIR, "FastK"), "-k%d" % args.kmer, "-t1", "-T%d" % cores,
                   "-P" + d, path]
            t0 = time.perf_counter()
            subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                           cwd=d)
            dt = time.perf_counter() - t0
            return dict(value=inst / dt, unit="k-mers/s", cores=cores, kind="reference",
                        sample=sample + " (FASTQ file, reference FastK -T%d, %.1f s wall)" % (cores, dt))
        finally:
            subprocess.run(["rm", "-rf", d])
    t0 = time.perf_counter()
    res = orc.fastk(args.kmer, bases, boff, cutoff=1)
    dt = time.perf_counter() - t0
    assert res.ninst == inst
    return dict(value=inst / dt, unit="k-mers/s", cores=1, kind="port",
                sample=sample + " (scalar CPU restatement, %.1f s)" % dt)


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run --nproc-per-node %d"
                             % (args.gpus, args.gpus))
        args.gpus = world

    import torch
    import fastk_amd
    from fastk_amd import shard

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    if os.environ.get("FK_RANKS_SHARE_GPU") == "1":
        # test rig for boxes with one GPU (tests/test_gpu_parity.py): all ranks on device 0, RCCL told
        # that they sit on different hosts so that it accepts them (socket transport; not a measurement)
        local_rank = 0
        os.environ["NCCL_HOSTID"] = "fk-rank-%d" % rank
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        os.environ.setdefault("NCCL_IB_DISABLE", "1")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    sharded = world > 1 or args.force_shard
    if sharded:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:
            os.environ["MASTER_ADDR"] = "127.0.0.1"
            os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)

    L = args.read_len
    glen = int(args.genome_mbp * 1e6) * world
    total_reads = int(args.coverage * glen / L)
    per = total_reads // world
    first = rank * per
    nbytes = per * (L + 1)

    ctx = fastk_amd.Context(kmer=args.kmer, table_cutoff=1, nthreads=4, device=local_rank,
                            nbuckets=world * max(1, args.exchange_rounds) if sharded else max(1, args.stream_buckets))
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    for kv in args.debug:
        key, val = kv.split("=")
        ctx.debug_set(key, int(val))
    reads = torch.empty(nbytes + 64, dtype=torch.uint8, device=dev)
    ctx._ck(ctx.L.fk_synth_reads(ctx.h, args.seed, glen, L, args.err_ppm, first, per,
                                 reads.data_ptr()))
    torch.cuda.synchronize()
    engine = shard.HipEngine(ctx, dev)
    if sharded:
        engine.train_buckets(reads[:nbytes])          # scheme set-up, like Determine_Scheme: not a step

    def step(verify=False):
        if not sharded:
            return ctx.count_device_reads(reads.data_ptr(), nbytes, fetch_table=False)
        if args.exchange_rounds > 1:
            return shard.count_sharded_rounds(engine, reads[:nbytes], args.exchange_rounds, verify=verify)
        return shard.count_sharded(engine, reads[:nbytes], verify=verify)

    def barrier():
        torch.cuda.synchronize()
        if sharded:
            dist.barrier()
            torch.cuda.synchronize()

    last = None
    for _ in range(args.warmup):
        last = step(verify=True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = step()
    barrier()
    dt = time.perf_counter() - t0
    if sharded:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if not sharded:
        loc = last
        ninst, nsuper, nweighted, ndistinct = last.ninst, last.nsuper, last.nweighted, last.ndistinct
    else:
        loc = last["local"]["result"]
        ninst, nsuper = last["ninst"], last["nsuper"]
        nweighted, ndistinct = last["nweighted"], last["ndistinct"]
    expect = total_reads // world * world * (L - args.kmer + 1)
    assert ninst == expect, "k-mer instance count %d != %d" % (ninst, expect)

    ms_step = 1e3 * dt / args.steps
    value = ninst / (dt / args.steps)

    # roofline of the dominant kernel: k_rx_scatter on the W weighted k-mer records (the two hashed
    # digit passes that bring equal k-mers into one of 65,536 bins before they are summed in LDS),
    # one stable 8-bit digit pass per launch.  Algorithmic bytes per launch = 2 * n * R (records read once and written
    # once at the reference width R = KMER_WORD); duration = HIP event pair around every scatter
    # launch on the library's stream, averaged over the launches of the last step.  pass_total adds
    # the per-pass helper kernels (digit-stream histogram + two scans) that feed it.
    w = ctx.w
    n_pass = loc.nweighted
    npk = max(loc.passes_kmer, 1)
    avg_ms = loc.ms_scatter_kmer / npk
    avg_pass_ms = loc.ms_pass_kmer / npk
    gbs = lambda nrec, width, ms: round((2.0 * nrec * width) / (ms * 1e-3) / 1e9, 1) if ms > 0 else 0.0
    achieved = gbs(n_pass, w.kmer_word, avg_ms)
    nps = max(loc.passes_super, 1)
    # HBM traffic per launch from PMC counters cannot be collected by this process; it comes from
    # the committed rocprofv3 passes over this same command (profiles/r01_pmc_traffic.json) and is
    # only reported when the workload (records per launch) is the profiled one.
    traffic = None
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
        if abs(pm["weighted_kmers"] - n_pass) <= 5e-3 * n_pass and w.kmer_word == 12:
            cands = [v for k, v in pm["kernels"].items() if k.startswith("_Z12k_rx_scatterILi3ELi12ELb")]
            traffic = max(c["traffic_bytes"] for c in cands)       # the launches over all W records
    except Exception:
        traffic = None
    roofline = dict(bound="hbm", achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic,
                    traffic_unit="bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, profiles/r01_pmc_traffic.json)",
                    algorithmic_bytes=2.0 * n_pass * w.kmer_word,
                    kernel="k_rx_scatter<3,12> (weighted k-mer records, R=%d B)" % w.kmer_word,
                    records_per_launch=int(n_pass), launches_per_step=int(loc.passes_kmer),
                    avg_launch_ms=round(avg_ms, 4),
                    pass_total=dict(avg_ms=round(avg_pass_ms, 4),
                                    achieved=gbs(n_pass, w.kmer_word, avg_pass_ms),
                                    note="scatter + k_rx_tilehist + k_rx_chunkscan + k_rx_superscan"),
                    table_sort=dict(records=int(loc.ncollapsed), launches=int(loc.passes_final),
                                    achieved_pass_total=gbs(loc.ncollapsed, w.kmer_word,
                                                            loc.ms_pass_final / max(loc.passes_final, 1))),
                    supermer_pass=dict(
                        kernel="k_rx_scatter_w<5,4,hashed> (super-mer records, R=%d B)" % w.smer_word,
                        records=int(loc.nsuper), launches=int(loc.passes_super),
                        avg_launch_ms=round(loc.ms_scatter_super / nps, 4),
                        achieved=gbs(loc.nsuper, w.smer_word, loc.ms_scatter_super / nps),
                        pass_total_achieved=gbs(loc.nsuper, w.smer_word, loc.ms_pass_super / nps)))

    out = dict(metric="canonical k-mers/sec (k=40, whole hot path, reads resident in HBM)",
               value=value, unit="k-mers/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
               ms_per_step=ms_step, higher_is_better=True, scaling="weak", vs_baseline=None,
               dtype="u8", data="synthetic",
               config=dict(workload="%gx coverage, %d bp reads, err %d ppm, of a %g Mbp genome per GPU"
                                    ", k=%d -t1 (BASELINE.json configs[1])"
                                    % (args.coverage, L, args.err_ppm, args.genome_mbp, args.kmer),
                           reads_per_gpu=per, kmer_instances=int(ninst), supermers=int(nsuper),
                           weighted_kmers=int(nweighted), distinct_kmers=int(ndistinct),
                           parallelism="minimizer-bucket shard x%d" % world
                                       + (", exchange in %d overlapped rounds" % args.exchange_rounds
                                          if sharded and args.exchange_rounds > 1 else "")),
               roofline=roofline,
               stage_ms=dict((k, round(v, 3)) for k, v in loc.ms.items()))
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out))
    ctx.close()
    if sharded:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
