#!/usr/bin/env python3
"""bench.py -- canonical k-mers/s of the split -> sort -> expand -> sort -> count hot path on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W   (N>1 under torch.distributed.run)
prints ONE JSON line on rank 0.  A "step" is one pass of the whole hot path over one batch of
synthetic reads that is already resident in HBM when the timed region starts.

N = 1 (default --config 2) = BASELINE.json configs[2], the set the north-star target is quoted on:
50x coverage of a 3 Gbp uniform random genome in 15 kbp reads with 0.2 % substitutions, k=40 -t4
(10 M reads, 150 G bases, ~149.6 G k-mer instances).  The 150 GB of ASCII reads stay resident; the
8.6 G super-mer records (172 GB) do not fit beside them, so the reads are split in `split_passes`
passes, each keeping one group of minimizer buckets (the first pass finds the minimizers and records
4-byte entries for the rest, the later passes replay them), and the buckets are counted one after the
other (the role of FastK's NPARTS, split.c:617-766; count.c:1202 bucket loop).
    --config 1 = BASELINE.json configs[1]: 50x of 100 Mbp in 150 bp reads, k=40 -t1, all resident.
N > 1 (default --config 3) = BASELINE.json configs[3]: the SAME 3 Gbp HiFi-shaped set striped over the N GPUs
(strong scaling: the N = 1 line above is its first point), counted through the C engine's sharded entry
points: rank r owns read stripe r, super-mers travel by minimizer bucket with grouped ncclSend/ncclRecv in
rounds that overlap the counting, histograms are all-reduced (fk_shard_count_device, fastk_amd/csrc/
fk_shard.hip).  --config 1 with N > 1 is configs[1]'s workload PER GPU (weak scaling) through the same C engine
(round 5: the torch.distributed harness it used to run through is test infrastructure now, tests/shard_model.py).

The line also carries (N = 1):
  roofline      dominant radix kernel (k_rx_scatter over the weighted k-mer records): algorithmic
                2*n*R bytes per launch / launch duration by HIP events on the library's stream;
                copy_ceiling = a device-to-device hipMemcpy measured in this process; copy_kernel_ceiling = the
                library's own uint4 copy kernel (fk_copy_rate) on the same buffers, 10 % above it
  value_device  SURVEY 8(d) "device pipeline": first H2D of reads lying in pinned host memory ->
                sorted table in pinned host memory (fk_push_block ... fk_finish)
  value_e2e     SURVEY 8(d) "end to end": FASTA file -> .hist + .ktab files through bin/FastK_amd
                (file in /dev/shm; --e2e-scale < 1 shrinks the genome, stated in the record)
  cpu_baseline  reference FastK (oracle/_ref, built from the reference's sources) on this box's
                host cores on a bounded sample of the same shape
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

CONFIGS = {
    1: dict(genome_mbp=100.0, coverage=50.0, read_len=150, err_ppm=1000, cutoff=1, buckets=1,
            split_passes=1, cpu_sample_mbp=20.0,
            label="50x coverage, 150 bp reads, err 1000 ppm, of a %g Mbp genome%s, k=%d -t1 "
                  "(BASELINE.json configs[1])"),
    2: dict(genome_mbp=3000.0, coverage=50.0, read_len=15000, err_ppm=2000, cutoff=4, buckets=48,
            split_passes=3, cpu_sample_mbp=200.0,
            label="50x coverage, 15 kbp HiFi-shaped reads, err 2000 ppm, of a %g Mbp genome%s, k=%d -t4 "
                  "(BASELINE.json configs[2])"),
    3: dict(genome_mbp=3000.0, coverage=50.0, read_len=15000, err_ppm=2000, cutoff=4, buckets=1,
            split_passes=1, cpu_sample_mbp=200.0,
            label="50x coverage, 15 kbp HiFi-shaped reads, err 2000 ppm, of a %g Mbp genome%s, k=%d -t4, "
                  "sharded over the GPUs by minimizer bucket (BASELINE.json configs[3])"),
}


def parse():
    if any(a == "--debug" or a.startswith("--debug=") for a in sys.argv[1:]):
        os.environ["FASTK_AMD_TEST_KNOBS"] = "1"          # (fk_debug_set takes its knobs from measurement processes only)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=0, choices=[0, 1, 2, 3],
                    help="BASELINE.json configs[] index; 0 = 2 on one GPU, 3 on several (the same set sharded through "
                         "the C engine fk_shard_*, strong scaling); 1 = configs[1] (per GPU with N > 1: weak scaling)")
    ap.add_argument("--kmer", type=int, default=40)
    ap.add_argument("--genome-mbp", type=float, default=None, help="genome size per GPU (Mbp)")
    ap.add_argument("--scale", type=float, default=1.0,
                    help="development aid: shrink the genome by this factor (named in config.workload)")
    ap.add_argument("--coverage", type=float, default=None)
    ap.add_argument("--read-len", type=int, default=None)
    ap.add_argument("--err-ppm", type=int, default=None)
    ap.add_argument("--cutoff", type=int, default=None)
    ap.add_argument("--seed", type=int, default=20251001)
    ap.add_argument("--cpu-sample-mbp", type=float, default=None,
                    help="genome size of the bounded CPU-baseline sample (same coverage/shape)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-device-leg", action="store_true", help="skip value_device (pinned host -> host)")
    ap.add_argument("--no-e2e", action="store_true", help="skip value_e2e (file -> files)")
    ap.add_argument("--no-packed-leg", action="store_true", help="skip value_packed_resident (reads resident in two bits per base)")
    ap.add_argument("--e2e-pause", type=float, default=12.0,
                    help="seconds between the runs of the e2e leg (the driver wipes released device memory in the background)")
    ap.add_argument("--e2e-scale", type=float, default=1.0,
                    help="value_e2e runs on this fraction of the genome (1 = the full 150 GB FASTA)")
    ap.add_argument("--force-shard", action="store_true",
                    help="run the sharded path (fk_shard_*: split, exchange, count, gather) even on one GPU")
    ap.add_argument("--stream-buckets", type=int, default=None,
                    help="minimizer buckets counted one after the other, N=1 only")
    ap.add_argument("--split-passes", type=int, default=None,
                    help="split passes over the resident reads (each keeps 1/passes of the super-mers)")
    ap.add_argument("--exchange-rounds", type=int, default=4,
                    help="N > 1: the super-mer exchange is cut into this many pieces; piece i+1 travels "
                         "while piece i is counted (1 = one all-to-all, then count)")
    ap.add_argument("--ascii-stripes", action="store_true",
                    help="N > 1: keep the ranks' stripes as 0-terminated ASCII (default: two bits per base)")
    ap.add_argument("--debug", action="append", default=[], metavar="KEY=VALUE",
                    help="fk_debug_set knob for ablation runs (results may be invalid)")
    ap.add_argument("--dump-table", default=None, metavar="DIR",
                    help="N > 1 / --config 3: every rank writes its gathered range of the table to DIR/table.<rank> (tests)")
    ap.add_argument("--verbose", action="store_true")
    return ap.parse_args()


def hist_file_sha256(kmer, hist, max_inst):
    """sha256 of the bytes of <root>.hist for this histogram (count.c:1893-1910: int k, int 1, int 0x7fff, int64
    counts[1], int64 max_inst, int64 counts[1..0x7fff]; 262,164 bytes) -- what the golden cases pin"""
    import hashlib
    h = np.asarray(hist, dtype=np.int64)
    raw = np.array([kmer, 1, 0x7fff], dtype=np.int32).tobytes() + np.array([h[1], max_inst], dtype=np.int64).tobytes() \
        + h[1:0x8000].astype(np.int64).tobytes()
    assert len(raw) == 262164
    return hashlib.sha256(raw).hexdigest()


def log(args, *a):
    if args.verbose:
        print("[bench]", *a, file=sys.stderr, flush=True)


def write_fastx(path, bases, nreads, L, fastq):
    """reads as rows of L+1 bytes (0-terminated) -> FASTA (one line per read) or FASTQ file."""
    rows = bases.reshape(nreads, L + 1)[:, :L]
    if fastq:
        mat = np.empty((nreads, 3 + L + 3 + L + 1), dtype=np.uint8)
        mat[:, 0:3] = np.frombuffer(b"@r\n", dtype=np.uint8)
        mat[:, 3:3 + L] = rows
        mat[:, 3 + L:6 + L] = np.frombuffer(b"\n+\n", dtype=np.uint8)
        mat[:, 6 + L:6 + 2 * L] = ord("I")
        mat[:, 6 + 2 * L] = ord("\n")
    else:
        mat = np.empty((nreads, 3 + L + 1), dtype=np.uint8)
        mat[:, 0:3] = np.frombuffer(b">r\n", dtype=np.uint8)
        mat[:, 3:3 + L] = rows
        mat[:, 3 + L] = ord("\n")
    mat.tofile(path)


def _write_sample(path, orc, seed, glen, L, err_ppm, nreads, fastq):
    """the synthetic reads of a sample as a FASTA / FASTQ file, 20,000 reads at a time"""
    with open(path, "wb") as f:
        for r0 in range(0, nreads, 20000):
            n = min(20000, nreads - r0)
            b, _ = orc.synth_block(seed, glen, L, err_ppm, r0, n)
            rows = b.reshape(n, L + 1)[:, :L]
            if fastq:
                mat = np.empty((n, 3 + L + 3 + L + 1), dtype=np.uint8)
                mat[:, 0:3] = np.frombuffer(b"@r\n", dtype=np.uint8)
                mat[:, 3:3 + L] = rows
                mat[:, 3 + L:6 + L] = np.frombuffer(b"\n+\n", dtype=np.uint8)
                mat[:, 6 + L:6 + 2 * L] = ord("I")
                mat[:, 6 + 2 * L] = ord("\n")
            else:
                mat = np.empty((n, 3 + L + 1), dtype=np.uint8)
                mat[:, 0:3] = np.frombuffer(b">r\n", dtype=np.uint8)
                mat[:, 3:3 + L] = rows
                mat[:, 3 + L] = ord("\n")
            mat.tofile(f)


def cpu_baseline(args, cfg):
    """Reference FastK (oracle/_ref/FastK, built from the reference sources) on this box's host cores, on a
    bounded sample of the same workload: the thread count is chosen by one run each of -T32/64/128/256 on a
    small sample (the reference is fastest at 32-64 threads on a 256-thread box; -T<all cores> is 1.4x slower),
    then the sample proper runs twice with the best one.  Falls back to the scalar port without the reference."""
    from oracle import orc
    cores = os.cpu_count() or 1
    L = cfg["read_len"]
    fastq = L <= 1000

    def sample_of(mbp):
        glen = int(mbp * 1e6)
        nreads = int(cfg["coverage"] * glen / L)
        return glen, nreads, nreads * (L - args.kmer + 1)

    def text(mbp, nreads):
        return "%gx coverage of a %g Mbp genome, %d x %d bp reads, err %d ppm, k=%d -t%d" % (
            cfg["coverage"], mbp, nreads, L, cfg["err_ppm"], args.kmer, cfg["cutoff"])

    if not orc.have_ref():
        glen, nreads, inst = sample_of(min(cfg["cpu_sample_mbp"], 20.0))
        bases, boff = orc.synth_block(args.seed, glen, L, cfg["err_ppm"], 0, nreads)
        t0 = time.perf_counter()
        res = orc.fastk(args.kmer, bases, boff, cutoff=cfg["cutoff"])
        dt = time.perf_counter() - t0
        assert res.ninst == inst
        return dict(value=inst / dt, unit="k-mers/s", cores=1, kind="port",
                    sample=text(min(cfg["cpu_sample_mbp"], 20.0), nreads) + " (scalar CPU restatement, %.1f s)" % dt)

    def run(path, d, T):
        cmd = [os.path.join(orc.REF_DIR, "FastK"), "-k%d" % args.kmer, "-t%d" % cfg["cutoff"], "-T%d" % T, "-P" + d, path]
        t0 = time.perf_counter()
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=d)
        return time.perf_counter() - t0

    d = tempfile.mkdtemp(prefix="fkbase")
    try:
        # thread count: one run each on a small sample
        small_mbp = min(20.0, cfg["cpu_sample_mbp"])
        glen, nreads, inst = sample_of(small_mbp)
        small = os.path.join(d, "t.fastq" if fastq else "t.fasta")
        _write_sample(small, orc, args.seed, glen, L, cfg["err_ppm"], nreads, fastq)
        sweep = {}
        for T in (32, 64, 128, 256):
            if T <= cores:
                sweep[T] = round(inst / run(small, d, T) / 1e9, 4)
        if not sweep:
            sweep[cores] = round(inst / run(small, d, cores) / 1e9, 4)
        best_T = max(sweep, key=lambda t: sweep[t])
        os.remove(small)
        # the sample proper, twice
        mbp = cfg["cpu_sample_mbp"]
        glen, nreads, inst = sample_of(mbp)
        path = os.path.join(d, "s.fastq" if fastq else "s.fasta")
        _write_sample(path, orc, args.seed, glen, L, cfg["err_ppm"], nreads, fastq)
        times = [run(path, d, best_T) for _ in range(2)]
        dt = min(times)
        out = dict(value=inst / dt, unit="k-mers/s", cores=best_T, kind="reference", host_threads=cores,
                   thread_sweep_gkmers_per_s={"-T%d" % t: v for t, v in sweep.items()},
                   thread_sweep_sample=text(small_mbp, sample_of(small_mbp)[1]),
                   seconds=[round(t, 2) for t in times],
                   sample=text(mbp, nreads) + " (%s file, reference FastK -T%d = the fastest of the sweep, best of two "
                                              "runs %.1f s wall, parse and file writes included)"
                                              % ("FASTQ" if fastq else "FASTA", best_T, dt))
        # the run is also a parity anchor: this sample is the golden case hifi50x200M_k40_t4_T8
        g = os.path.join(ROOT, "tests", "golden", "hifi50x200M_k40_t4_T8.json")
        if os.path.exists(g):
            case = json.load(open(g))
            sy = case["synth"]
            if (sy["seed"], sy["genome_len"], sy["read_len"], sy["err_ppm"], sy["nreads"], case["k"], case["cutoff"]) == \
                    (args.seed, glen, L, cfg["err_ppm"], nreads, args.kmer, cfg["cutoff"]):
                import hashlib
                h = hashlib.sha256(open(os.path.join(d, "s.hist"), "rb").read()).hexdigest()
                out["hist_sha256_equals_golden"] = (h == case["expected"]["hist_sha256"])
        # the reference on the timed configuration itself was run once (tools/cpu_baseline_full.py, ~6 min of host time:
        # not something the default run repeats): its record, when committed, rides along
        full = reference_full_size_record()
        if full is not None:
            out["full_size_run"] = dict(full, measured_in_this_run=False)
        return out
    finally:
        subprocess.run(["rm", "-rf", d])


def copy_ceiling(torch, dev, ctx=None):
    """Device-to-device copy rate on this box (read + write bytes per second), 4 GiB buffers: hipMemcpy (what
    torch's copy_ issues) and, with a context, the library's own copy kernel."""
    n = 4 << 30
    a = torch.empty(n, dtype=torch.uint8, device=dev)
    b = torch.empty(n, dtype=torch.uint8, device=dev)
    a.fill_(1)
    b.copy_(a)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        b.copy_(a)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    kern = None
    if ctx is not None:                  # ... and of a hand-written uint4 copy kernel (fk_copy_rate): the truer ceiling
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        kern = round(ctx.copy_rate(b.data_ptr(), a.data_ptr(), n), 1)
    del a, b
    torch.cuda.empty_cache()
    return round(2.0 * n / (best * 1e-3) / 1e9, 1), kern


def device_leg(args, cfg, fastk_amd, ctx_gen, glen, nreads, L, local_rank):
    """SURVEY 8(d) device pipeline: reads in pinned host memory -> fk_push_block (H2D) -> fk_finish ->
    sorted table in pinned host memory; then the same with the reads in two bits per base (fk_push_packed: a quarter of
    the bytes over PCIe).  Returns the record for the JSON line."""
    import ctypes as C
    lib = ctx_gen.L
    per_block = max(4, min(nreads, (1 << 30) // (L + 1)) // 4 * 4)  # ~1 GB DATA_BLOCKs (int32 offsets); whole code bytes
    nbytes = nreads * (L + 1)
    cbytes = (nreads * L + 3) // 4
    host, hcodes = C.c_void_p(), C.c_void_p()
    t0 = time.perf_counter()
    if lib.fk_host_alloc(nbytes + 64, C.byref(host)) != 0 or lib.fk_host_alloc(cbytes + 64, C.byref(hcodes)) != 0:
        return dict(skipped="cannot pin %d bytes of host memory" % (nbytes + cbytes))
    t_pin = time.perf_counter() - t0
    piece = ctx_gen.alloc(per_block * (L + 1) + 64)
    cpiece = ctx_gen.alloc(per_block * L // 4 + 64)
    for first in range(0, nreads, per_block):
        n = min(per_block, nreads - first)
        ctx_gen.synth_reads(args.seed, glen, L, cfg["err_ppm"], first, n, buf=piece)
        ctx_gen._ck(lib.fk_copy_to_host(ctx_gen.h, host.value + first * (L + 1), piece.ptr, n * (L + 1)))
        ctx_gen._ck(lib.fk_pack_fixed_reads(ctx_gen.h, piece.ptr, n, L, cpiece.ptr))
        ctx_gen._ck(lib.fk_copy_to_host(ctx_gen.h, hcodes.value + first * L // 4, cpiece.ptr, (n * L + 3) // 4))
    piece.free()
    cpiece.free()
    boff = (np.arange(per_block + 1, dtype=np.int64) * (L + 1)).astype(np.int32)
    rlen = np.full(per_block, L, dtype=np.int32)
    budget = int(args.device_budget_gb * 1e9)
    out = {}
    with fastk_amd.Context(kmer=args.kmer, table_cutoff=cfg["cutoff"], nthreads=4, device=local_rank,
                           nbuckets=max(cfg["buckets"], 1) if budget else 1, hbm_budget=budget) as ctx:
        for kv in args.debug:
            key, val = kv.split("=")
            ctx.debug_set(key, int(val))
        if budget and cfg["buckets"] > 1:
            sample = np.ctypeslib.as_array(C.cast(host, C.POINTER(C.c_uint8)), shape=(min(nbytes, 8 << 20),))
            ctx.set_bucket_weights(ctx.bucket_census(sample))
        times, ref = {}, None
        for rep, form in enumerate(("ascii", "ascii", "packed", "packed")):   # a form's first run sizes its buffers
            ctx.reset()
            t0 = time.perf_counter()
            for first in range(0, nreads, per_block):
                n = min(per_block, nreads - first)
                if form == "ascii":
                    ctx._ck(lib.fk_push_block(ctx.h, host.value + first * (L + 1), boff.ctypes.data, n, 0, 0))
                else:
                    ctx._ck(lib.fk_push_packed(ctx.h, hcodes.value + first * L // 4, n * L, rlen.ctypes.data, n, None, 0, 0, 0))
            t_push = time.perf_counter() - t0
            res = fastk_amd.api.CResult()
            ctx._ck(lib.fk_finish(ctx.h, C.byref(res)))
            times.setdefault(form, []).append(time.perf_counter() - t0)
            log(args, "device leg run %d (%s): push %.3f s, finish %.3f s (device ms: split %.0f, super-mers %.0f, expand %.0f, "
                      "k-mers %.0f, count %.0f, table sort %.0f, total %.0f)" % (
                rep, form, t_push, times[form][-1] - t_push, res.ms_split, res.ms_sort_super, res.ms_expand, res.ms_sort_kmer,
                res.ms_count, res.ms_table_sort, res.ms_total))
            inst = int(res.ninst)
            h = np.ctypeslib.as_array(res.hist).astype(np.int64)
            conserved = int((h[1:0x7fff] * np.arange(1, 0x7fff)).sum()) + int(res.max_inst)
            assert conserved == inst == nreads * (L - args.kmer + 1), "device leg (%s): %d k-mer instances counted, %d " \
                "in the histogram, %d expected" % (form, inst, conserved, nreads * (L - args.kmer + 1))
            sig = (h.tobytes(), int(res.max_inst), int(res.ntable))
            assert ref is None or sig == ref, "device leg: the packed form gives another histogram"
            ref = sig
        out = dict(value=inst / times["ascii"][-1], unit="k-mers/s", seconds=round(times["ascii"][-1], 3),
                   first_run_seconds=round(times["ascii"][0], 3), pin_seconds=round(t_pin, 2),
                   h2d_bytes=int(nbytes), d2h_bytes=int(res.ntable) * ctx.w.kmer_word,
                   table_entries=int(res.ntable), spilled_bytes=int(res.spilled_bytes),
                   hbm_budget_gb=args.device_budget_gb,
                   definition="first H2D of pinned host reads (fk_push_block, ~1 GB blocks of ASCII) -> sorted table "
                              "in pinned host memory (fk_finish); second of two runs",
                   packed=dict(value=inst / times["packed"][-1], unit="k-mers/s", seconds=round(times["packed"][-1], 3),
                               first_run_seconds=round(times["packed"][0], 3), h2d_bytes=int(cbytes + 4 * nreads),
                               definition="the same with the reads in two bits per base in pinned host memory "
                                          "(fk_push_packed: codes + read lengths; the reads stay packed in HBM and are split in that form)"))
    lib.fk_host_free(host)
    lib.fk_host_free(hcodes)
    return out


def write_synth_file(ctx_gen, path, fastq, seed, glen, L, err_ppm, nreads):
    """The synthetic reads of include/fk_synth.h as a FASTQ (four lines a record) or FASTA (one line a read) file."""
    per = max(1, min(nreads, (2 << 30) // (L + 1)))
    piece = ctx_gen.alloc(per * (L + 1) + 64)
    with open(path, "wb") as f:
        for first in range(0, nreads, per):
            n = min(per, nreads - first)
            ctx_gen.synth_reads(seed, glen, L, err_ppm, first, n, buf=piece)
            rows = piece.download(n * (L + 1)).reshape(n, L + 1)[:, :L]
            if fastq:
                mat = np.empty((n, 3 + L + 3 + L + 1), dtype=np.uint8)
                mat[:, 0:3] = np.frombuffer(b"@r\n", dtype=np.uint8)
                mat[:, 3:3 + L] = rows
                mat[:, 3 + L:6 + L] = np.frombuffer(b"\n+\n", dtype=np.uint8)
                mat[:, 6 + L:6 + 2 * L] = ord("I")
                mat[:, 6 + 2 * L] = ord("\n")
            else:
                mat = np.empty((n, 3 + L + 1), dtype=np.uint8)
                mat[:, 0:3] = np.frombuffer(b">r\n", dtype=np.uint8)
                mat[:, 3:3 + L] = rows
                mat[:, 3 + L] = ord("\n")
            mat.tofile(f)
    piece.free()


def ktab_stream_sha256(d, root):
    """sha256 of a .ktab's canonical stream: the stub's prefix index (from byte 16) followed by the payloads of the hidden
    parts (from byte 12 of each), in part order -- the bytes that do not depend on how many parts the table was cut into
    (table.c:485-498; the digest the golden fixtures carry as ktab.stream_sha256).  Returns (digest, entries, parts)."""
    import hashlib
    import struct
    h = hashlib.sha256()
    with open(os.path.join(d, root + ".ktab"), "rb") as f:
        head = f.read(16)
        nparts = struct.unpack("<iiii", head)[1]
        while True:
            b = f.read(1 << 26)
            if not b:
                break
            h.update(b)
    nels = 0
    for t in range(1, nparts + 1):
        with open(os.path.join(d, ".%s.ktab.%d" % (root, t)), "rb") as f:
            nels += struct.unpack("<iq", f.read(12))[1]
            while True:
                b = f.read(1 << 26)
                if not b:
                    break
                h.update(b)
    return h.hexdigest(), nels, nparts


def reference_full_size_record():
    """the newest committed record of reference FastK on configs[2] itself (tools/cpu_baseline_full.py)"""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_cpu_baseline_configs2_full.json")), reverse=True):
        try:
            r = json.load(open(f))
            if "hist_file_sha256" in r:
                r["record"] = os.path.basename(f)
                return r
        except Exception:
            pass
    return None


def e2e_leg(args, cfg, fastk_amd, ctx_gen, L):
    """SURVEY 8(d) end to end: FASTA file (RAM disk) -> .hist + .ktab files through bin/FastK_amd."""
    glen = int(cfg["genome_mbp"] * 1e6 * args.scale * args.e2e_scale)
    nreads = int(cfg["coverage"] * glen / L)
    if nreads < 1:
        return dict(skipped="e2e sample empty")
    base = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    d = tempfile.mkdtemp(prefix="fke2e", dir=base)
    try:
        fastq = L <= 1000
        path = os.path.join(d, "reads.fastq" if fastq else "reads.fasta")
        write_synth_file(ctx_gen, path, fastq, args.seed, glen, L, cfg["err_ppm"], nreads)
        fbytes = os.path.getsize(path)
        exe = os.path.join(ROOT, "fastk_amd", "bin", "FastK_amd")
        cmd = [exe, "-v", "-k%d" % args.kmer, "-t%d" % cfg["cutoff"], "-T%d" % args.e2e_threads,
               "-M%d" % args.e2e_mem_gb, "-N" + os.path.join(d, "out"), path]
        # The driver wipes the device memory a process releases in the background, at ~34 GB/s, and a process that starts
        # meanwhile waits for it inside hipMalloc (tools/probe/malloc_probe.cpp: ~270 GB per run here = ~8 s of wiping).
        # The three timed runs therefore start on a quiet GPU, args.e2e_pause seconds after the process before them (the
        # first of them also reads a file that was only just written and is the slowest by far); a fourth run started
        # at once gives the back-to-back figure.
        times, phases = [], ""
        for rep in range(4):
            if rep < 3:
                time.sleep(args.e2e_pause)
            t0 = time.perf_counter()
            p = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
            if p.returncode != 0:
                return dict(failed="FastK_amd exit %d: %s" % (p.returncode, p.stderr[-600:]), input_bytes=fbytes,
                            scale=args.scale * args.e2e_scale)
            if rep == 3:
                back_to_back = time.perf_counter() - t0
                break
            times.append(time.perf_counter() - t0)
            if times[-1] == min(times):
                phases = " ".join(x.strip() for x in p.stderr.splitlines() if "Wall s" in x)
            if os.environ.get("FK_E2E_LOG"):
                with open(os.environ["FK_E2E_LOG"], "a") as lf:
                    lf.write("---- run: %.3f s\n%s\n" % (times[-1], p.stderr))
        inst = nreads * (L - args.kmer + 1)
        out_bytes = sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d)
                        if f.startswith("out") or f.startswith(".out"))
        # the files of the last run against reference FastK's on the same 150 G bases (a committed record of ONE reference
        # run at full size: .hist bytes and the canonical .ktab stream; count.c:1893-1910, table.c:485-498)
        parity = {}
        ref = reference_full_size_record()
        if ref is not None and ref.get("scale") == args.scale * args.e2e_scale and ref.get("kmer_instances") == inst:
            import hashlib
            t0 = time.perf_counter()
            hs = hashlib.sha256(open(os.path.join(d, "out.hist"), "rb").read()).hexdigest()
            parity["hist_equals_reference"] = (hs == ref["hist_file_sha256"])
            if "ktab_stream_sha256" in ref:
                dig, nels, _ = ktab_stream_sha256(d, "out")
                parity["ktab_stream_equals_reference"] = (dig == ref["ktab_stream_sha256"] and nels == ref["table_entries"])
                parity["ktab_stream_sha256"] = dig
            parity["reference_record"] = ref["record"]
            parity["digest_seconds"] = round(time.perf_counter() - t0, 1)
        return dict(value=inst / min(times), unit="k-mers/s", seconds=round(min(times), 3),
                    seconds_median=round(sorted(times)[len(times) // 2], 3), seconds_all=[round(x, 3) for x in times],
                    scale=args.scale * args.e2e_scale, kmer_instances=inst, input_bytes=fbytes, phases=phases, **parity,
                    output_bytes=out_bytes, seconds_back_to_back=round(back_to_back, 3), pause_seconds=args.e2e_pause,
                    command=" ".join(os.path.basename(c) if c == exe else
                                                             ("<file>" if c == path else c) for c in cmd[:-2]),
                    definition="process start -> exit of bin/FastK_amd on a %s file in %s: read + parse, "
                               "H2D, device pipeline, D2H, .hist + .ktab written; best of 3 runs on a quiet GPU "
                               "(seconds_back_to_back: started the moment the previous process exits, while the driver "
                               "still wipes the memory it released)"
                               % ("FASTQ" if fastq else "FASTA", base))
    finally:
        subprocess.run(["rm", "-rf", d])


def shim_leg(args, cfg):
    """The reference's own main() over the C-ABI shim (oracle/_ref/FastK_gpu) on the small sample
    the CPU reference's thread sweep runs on: fast (default) and FASTK_AMD_EXACT=1, process start to exit."""
    from oracle import orc
    exe = os.path.join(orc.REF_DIR, "FastK_gpu")
    if not os.path.exists(exe):
        return dict(skipped="oracle/_ref/FastK_gpu not built")
    L = cfg["read_len"]
    glen = int(20e6)
    nreads = int(cfg["coverage"] * glen / L)
    inst = nreads * (L - args.kmer + 1)
    fastq = L <= 1000
    base = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    d = tempfile.mkdtemp(prefix="fkshim", dir=base)
    try:
        path = os.path.join(d, "s.fastq" if fastq else "s.fasta")
        _write_sample(path, orc, args.seed, glen, L, cfg["err_ppm"], nreads, fastq)
        out = dict(sample="%gx coverage of a 20 Mbp genome, %d x %d bp reads" % (cfg["coverage"], nreads, L),
                   definition="process start -> exit of the reference's main() linked against libfastk_amd.so "
                              "(-T8, io.c's reader threads feed fk_push_block); best of 2")
        for mode in ("fast", "exact"):
            env = dict(os.environ)
            env.pop("FASTK_AMD_EXACT", None)
            if mode == "exact":
                env["FASTK_AMD_EXACT"] = "1"
            best = None
            for _ in range(2):
                t0 = time.perf_counter()
                p = subprocess.run([exe, "-k%d" % args.kmer, "-t%d" % cfg["cutoff"], "-T8", "-P" + d, path], cwd=d, env=env,
                                   stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
                if p.returncode != 0:
                    return dict(failed="FastK_gpu (%s) exit %d: %s" % (mode, p.returncode, p.stderr[-400:]))
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            out[mode] = dict(seconds=round(best, 3), value=inst / best, unit="k-mers/s")
        return out
    finally:
        subprocess.run(["rm", "-rf", d])


def main_config3(args, cfg, torch, fastk_amd, dist, rank, local_rank, world, dev, weak=False):
    """BASELINE configs[3]: the 50x HiFi-shaped 3 Gbp set of configs[2], striped over the GPUs (rank r generates
    reads r*n/world ..), counted through the C engine's sharded entry points (fk_shard_count_device: planned
    split, grouped ncclSend/ncclRecv of the records in overlapped rounds, per-rank counting, ncclAllReduce).
    Total work is fixed: strong scaling.  One process per GPU; torch.distributed only hands the RCCL id round
    and provides the barrier of the timing contract."""
    L = cfg["read_len"]
    glen = int(cfg["genome_mbp"] * 1e6 * args.scale) * (world if weak else 1)
    total_reads = int(cfg["coverage"] * glen / L) // world * world
    per = total_reads // world
    first = rank * per
    nbytes = per * (L + 1)
    rounds = max(4, 32 // world)
    if world * rounds > 256:
        raise SystemExit("bench.py: %d ranks x %d exchange rounds need %d minimizer buckets, a context takes 256"
                         % (world, rounds, world * rounds))
    ctx = fastk_amd.Context(kmer=args.kmer, table_cutoff=cfg["cutoff"], nthreads=4, device=local_rank,
                            nbuckets=world * rounds)
    idt = torch.zeros(128, dtype=torch.uint8, device=dev)
    if rank == 0:
        idt.copy_(torch.frombuffer(bytearray(fastk_amd.Shard.unique_id()), dtype=torch.uint8))
    if dist is not None:
        dist.broadcast(idt, 0)
    shard = fastk_amd.Shard(ctx, rank, world, bytes(idt.cpu().numpy().tobytes()))
    reads = torch.empty(nbytes + 64, dtype=torch.uint8, device=dev)
    ctx._ck(ctx.L.fk_synth_reads(ctx.h, args.seed, glen, L, cfg["err_ppm"], first, per, reads.data_ptr()))
    ctx._ck(ctx.L.fk_synchronize(ctx.h))
    # The stripe is resident in TWO BITS PER BASE (the north-star's form; a quarter of the bytes beside the records of the
    # exchange, one split pass): packed on the device before the timed region, the ASCII form released.  --ascii-stripes
    # keeps round 4's 0-terminated ASCII stripes.
    codes = roff = None
    if not args.ascii_stripes:
        nbases = per * L
        codes = torch.empty((nbases + 3) // 4 + 64, dtype=torch.uint8, device=dev)
        ctx._ck(ctx.L.fk_pack_fixed_reads(ctx.h, reads.data_ptr(), per, L, codes.data_ptr()))
        roff = torch.arange(per + 1, dtype=torch.int64, device=dev) * L
        ctx._ck(ctx.L.fk_synchronize(ctx.h))
        torch.cuda.synchronize()
        del reads
        reads = None
        torch.cuda.empty_cache()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # A step = C1 (split + exchange of the super-mers by minimizer bucket) + per-GPU sort and count + C2 (all-reduce of
    # histogram and census) + C3, the final gather: every rank receives its first-byte range of the whole table from
    # all ranks, orders it and brings it to pinned host memory (what the reference's Merge_Tables does before it writes;
    # file writing itself is outside, as at N = 1).  `value` is over the whole step; gather_ms is C3's share.
    nparts = max(8, world)
    nparts = (nparts + world - 1) // world * world

    def step():
        if codes is not None:
            res = shard.count_packed(codes.data_ptr(), per * L, roff.data_ptr(), per)
        else:
            res = shard.count(reads.data_ptr(), nbytes)
        t1 = time.perf_counter()
        shard.gather(res, nparts)
        return res, time.perf_counter() - t1

    last, gather_s = None, 0.0
    for _ in range(args.warmup):
        last, _g = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last, g = step()
        gather_s += g
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt, gather_s], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, gather_s = float(t[0].item()), float(t[1].item())
    expect = total_reads * (L - args.kmer + 1)
    assert last.ninst == expect, "k-mer instance count %d != %d" % (last.ninst, expect)
    gathered = shard.gather(last, nparts)                      # (outside the timed region: entries this rank holds)
    if dist is not None:
        t = torch.tensor([gathered], dtype=torch.int64, device=dev)
        dist.all_reduce(t)
        gathered = int(t.item())
    assert gathered == last.ntable, "the final gather holds %d of %d table entries" % (gathered, last.ntable)
    if args.dump_table:
        shard.gather(last, nparts, copy=True).tofile(os.path.join(args.dump_table, "table.%d" % rank))
    scale_note = "" if args.scale == 1.0 else " SCALED by %g (development run)" % args.scale
    # roofline of the graded kernel on rank 0's share: k_rx_scatter<3,12> over the weighted k-mers of the pieces
    # this rank counted (2 launches per piece), same accounting as at N = 1
    loc = shard.local_result()
    w = ctx.w
    nl = max(loc.launches_kmer, 1)
    passes_k = max(loc.passes_kmer, 1)
    algo = 2.0 * loc.nweighted * w.kmer_word * passes_k
    ach = round(algo / (loc.ms_scatter_kmer * 1e-3) / 1e9, 1) if loc.ms_scatter_kmer > 0 else 0.0
    roofline = dict(bound="hbm", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 4),
                    traffic=None, algorithmic_bytes=round(algo / nl, 1),
                    kernel="k_rx_scatter<3,12> (weighted k-mer records, R=%d B), rank 0's share" % w.kmer_word,
                    records_per_launch=int(loc.nweighted / (nl / passes_k)), launches_per_step=int(nl),
                    avg_launch_ms=round(loc.ms_scatter_kmer / nl, 4))
    if getattr(loc, "nrefs", 0) > 0:          # round 6: no pass over W; the table sort's pass on rank 0's share is graded
        roofline = roofline_record_refs(loc, w, 3, None, None)
        roofline["kernel"] += ", rank 0's share"
    # what the exchanges moved, rank by rank (fk_shard_get_stats): how many ranks RCCL's communicator really holds,
    # the super-mer bytes every rank sent / received / kept in C1, the device time of its sends and receives (they
    # overlap the counting), and C3 by phase
    mine = shard.stats()
    keys = ("comm_ranks", "rounds", "sent_bytes", "recv_bytes", "kept_bytes", "exchange_ms", "gather_sent_bytes",
            "gather_exchange_ms", "gather_sort_ms", "gather_d2h_ms")
    row = torch.tensor([float(mine[k]) for k in keys], dtype=torch.float64, device=dev)
    rows = [torch.zeros_like(row) for _ in range(world)]
    if dist is not None:
        dist.all_gather(rows, row)
    else:
        rows = [row]
    per_rank = [dict((k, (int(v) if k.endswith("bytes") or k in ("comm_ranks", "rounds") else round(float(v), 3)))
                     for k, v in zip(keys, r.tolist())) for r in rows]
    out = dict(metric="canonical k-mers/sec (k=%d, whole hot path incl. the final gather of the table to host memory, reads resident "
                      "in HBM %s)" % (args.kmer, "as 0-terminated ASCII" if codes is None else "in two bits per base"),
               value=last.ninst / (dt / args.steps), unit="k-mers/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
               ms_per_step=1e3 * dt / args.steps, higher_is_better=True, scaling="weak" if weak else "strong", vs_baseline=None,
               dtype="u8", data="synthetic",
               gather_ms=round(1e3 * gather_s / args.steps, 3), count_ms=round(1e3 * (dt - gather_s) / args.steps, 3),
               rccl_ranks=per_rank[0]["comm_ranks"], exchange=per_rank,
               hist_file_sha256=hist_file_sha256(args.kmer, last.hist, last.max_inst),
               value_without_gather=last.ninst / max((dt - gather_s) / args.steps, 1e-9),
               config=dict(workload=cfg["label"] % (cfg["genome_mbp"] * args.scale, " per GPU" if weak else "", args.kmer) + scale_note,
                           reads_per_gpu=per, bases_per_gpu=per * L, kmer_instances=int(last.ninst),
                           supermers=int(last.nsuper), weighted_kmers=int(last.nweighted),
                           distinct_kmers=int(last.ndistinct), table_entries=int(last.ntable), table_cutoff=cfg["cutoff"],
                           gathered_entries=int(gathered), table_parts=nparts,
                           parallelism="minimizer-bucket shard x%d through fk_shard_* (RCCL from C), %d exchange rounds "
                                       "overlapped with counting" % (world, ctx.params.nbuckets // world)),
               roofline=roofline,
               stage_ms_rank0=dict((k, round(v, 3)) for k, v in last.ms.items()))
    # (cpu_baseline rides on the N = 1 line only, as the measurement contract says: rank 0 of an N > 1 run would keep its
    #  peers waiting in the group's teardown for the minute the reference takes)
    if rank == 0:
        print(json.dumps(out))
    shard.close()
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


def roofline_record(loc, w, cfg_id, ceiling, ceiling_kernel):
    """roofline object of one resident step (loc = its fk_result): the dominant kernel is k_rx_scatter on the weighted
    k-mer records (the two hashed digit passes that bring equal k-mers into one of 65,536 bins before they are summed
    in LDS), one stable 8-bit digit pass per launch.  Algorithmic bytes per launch = 2 * n * R (records read once and
    written once at the reference width R = KMER_WORD); duration = HIP event pair around every scatter launch on the
    library's stream.  With bucket streaming a step has 2 launches per bucket: achieved = (sum of the launches'
    algorithmic bytes) / (sum of their durations), and the per-launch figures are the averages.  pass_total adds the
    per-pass helper kernels."""
    if getattr(loc, "nrefs", 0) > 0:
        return roofline_record_refs(loc, w, cfg_id, ceiling, ceiling_kernel)
    n_rec = loc.nweighted
    nl_k = max(loc.launches_kmer, 1)
    nl_s = max(loc.launches_super, 1)
    gbs = lambda nbytes_, ms: round(nbytes_ / (ms * 1e-3) / 1e9, 1) if ms > 0 else 0.0
    passes_k = max(loc.passes_kmer, 1)
    algo_k = 2.0 * n_rec * w.kmer_word * passes_k          # all scatter launches over W of one step
    achieved = gbs(algo_k, loc.ms_scatter_kmer)
    passes_s = max(loc.passes_super, 1)
    algo_s = 2.0 * loc.nsuper * w.smer_word * passes_s
    # HBM traffic per launch from PMC counters cannot be collected by this process; it comes from the committed
    # rocprofv3 passes over this same command and is only reported when (a) the workload (records per launch) is the
    # profiled one and (b) the kernel source has not changed since the profile was taken (the JSON carries the
    # sha256 of fastk_amd/csrc/fk_radix.hip it was collected with) -- otherwise null.
    traffic = None
    traffic_src = None
    import glob
    import hashlib
    names = sorted((os.path.basename(f) for f in glob.glob(os.path.join(ROOT, "profiles", "r0*_configs%d_pmc_traffic.json" % cfg_id))),
                   reverse=True)                              # the newest round's first
    try:
        radix_sha = hashlib.sha256(open(os.path.join(ROOT, "fastk_amd", "csrc", "fk_radix.hip"), "rb").read()).hexdigest()
    except OSError:
        radix_sha = None
    for name in names[:1]:                                    # only the newest profile counts
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", name)))
            per_launch = n_rec / (nl_k / passes_k)
            if abs(pm["records_per_launch"] - per_launch) <= 2e-2 * per_launch and w.kmer_word == 12 \
                    and pm.get("kernel_source_sha256") == radix_sha:
                traffic = pm["scatter_traffic_bytes_per_launch"]
                traffic_src = name
        except Exception:
            continue
    return dict(bound="hbm", achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s",
                frac=round(achieved / HBM_PEAK_GBS, 4), traffic=traffic,
                traffic_unit="bytes per launch (FETCH_SIZE + WRITE_SIZE with the guide's gfx950 corrections, "
                             "profiles/%s)" % traffic_src if traffic_src else None,
                copy_ceiling=ceiling, frac_of_copy_ceiling=round(achieved / ceiling, 4) if ceiling else None,
                copy_kernel_ceiling=ceiling_kernel,
                frac_of_copy_kernel_ceiling=round(achieved / ceiling_kernel, 4) if ceiling_kernel else None,
                algorithmic_bytes=round(algo_k / nl_k, 1),
                kernel="k_rx_scatter<3,12> (weighted k-mer records, R=%d B)" % w.kmer_word,
                records_per_launch=int(n_rec / (nl_k / passes_k)), launches_per_step=int(nl_k),
                avg_launch_ms=round(loc.ms_scatter_kmer / nl_k, 4),
                pass_total=dict(avg_ms=round(loc.ms_pass_kmer / nl_k, 4),
                                achieved=gbs(algo_k, loc.ms_pass_kmer),
                                note="scatter + k_rx_tilehist + k_rx_chunkscan + k_rx_superscan"),
                table_sort=dict(records=int(loc.ncollapsed), launches=int(loc.passes_final),
                                achieved_pass_total=gbs(2.0 * loc.ncollapsed * w.kmer_word * max(loc.passes_final, 1),
                                                        loc.ms_pass_final)),
                supermer_pass=dict(
                    kernel="k_rx_scatter_w<5,4,hashed> (super-mer records, R=%d B)" % w.smer_word,
                    records_per_launch=int(loc.nsuper / (nl_s / passes_s)), launches=int(nl_s),
                    avg_launch_ms=round(loc.ms_scatter_super / nl_s, 4),
                    achieved=gbs(algo_s, loc.ms_scatter_super),
                    pass_total_achieved=gbs(algo_s, loc.ms_pass_super)))


def roofline_record_refs(loc, w, cfg_id, ceiling, ceiling_kernel):
    """Round 6: no digit pass runs over the W weighted k-mer records any more (fk_recut.hip: 8-byte references to
    minimizer domains are sorted instead, the expansion writes the k-mers grouped).  The radix-sort pass over packed
    k-mer records that remains -- the same kernel, k_rx_scatter<3,12>, on the same 12-byte (k-mer, count) records -- is
    the LSD pass of the TABLE sort (3.0 G entries at configs[2], `passes_final` launches per step): that is the graded
    kernel now.  Algorithmic bytes per launch = 2 * n * R; duration = HIP event pair around every scatter launch on the
    library's stream.  `reference_sort` and `supermer_pass` are the other two instantiations of the pass."""
    gbs = lambda nbytes_, ms: round(nbytes_ / (ms * 1e-3) / 1e9, 1) if ms > 0 else 0.0
    nt = loc.ntable
    pf = max(loc.passes_final, 1)
    algo_t = 2.0 * nt * w.kmer_word * pf
    achieved = gbs(algo_t, loc.ms_scatter_final)
    nl_s = max(loc.launches_super, 1)
    passes_s = max(loc.passes_super, 1)
    algo_s = 2.0 * loc.nsuper * w.smer_word * passes_s
    nl_r = max(loc.launches_kmer, 1)
    passes_r = max(loc.passes_kmer, 1)
    algo_r = 2.0 * loc.nrefs * 8 * passes_r
    return dict(bound="hbm", achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s",
                frac=round(achieved / HBM_PEAK_GBS, 4), traffic=None, traffic_unit=None,
                copy_ceiling=ceiling, frac_of_copy_ceiling=round(achieved / ceiling, 4) if ceiling else None,
                copy_kernel_ceiling=ceiling_kernel,
                frac_of_copy_kernel_ceiling=round(achieved / ceiling_kernel, 4) if ceiling_kernel else None,
                algorithmic_bytes=round(algo_t / pf, 1),
                kernel="k_rx_scatter<3,12> (LSD pass of the table sort over the (k-mer, count) records, R=%d B)" % w.kmer_word,
                records_per_launch=int(nt), launches_per_step=int(pf),
                avg_launch_ms=round(loc.ms_scatter_final / pf, 4),
                pass_total=dict(avg_ms=round(loc.ms_pass_final / pf, 4), achieved=gbs(algo_t, loc.ms_pass_final),
                                note="scatter + k_rx_tilehist + k_rx_chunkscan + k_rx_superscan"),
                kmer_grouping=dict(
                    note="no pass over the W weighted k-mer records: references to minimizer domains are sorted instead",
                    weighted_kmers=int(loc.nweighted), references=int(loc.nrefs), launches_over_W=0),
                reference_sort=dict(
                    kernel="k_rx_scatter<2,*> (references, R=8 B)", records_per_launch=int(loc.nrefs / (nl_r / passes_r)),
                    launches=int(nl_r), avg_launch_ms=round(loc.ms_scatter_kmer / nl_r, 4),
                    achieved=gbs(algo_r, loc.ms_scatter_kmer), pass_total_achieved=gbs(algo_r, loc.ms_pass_kmer)),
                supermer_pass=dict(
                    kernel="k_rx_scatter_w<5,4,hashed> (super-mer records, R=%d B)" % w.smer_word,
                    records_per_launch=int(loc.nsuper / (nl_s / passes_s)), launches=int(nl_s),
                    avg_launch_ms=round(loc.ms_scatter_super / nl_s, 4),
                    achieved=gbs(algo_s, loc.ms_scatter_super),
                    pass_total_achieved=gbs(algo_s, loc.ms_pass_super)))


def train_buckets(ctx, reads, nbytes, sample_bytes=8 << 20):
    """Balance the minimizer buckets on a sample of the reads (scheme set-up, like Determine_Scheme: not a step)."""
    sample = reads[:min(nbytes, sample_bytes)].cpu().numpy()
    ctx.set_bucket_weights(ctx.bucket_census(sample))


def packed_resident_leg(args, cfg, fastk_amd, torch, dev, local_rank, reads, nbytes, per, L, ceiling, ceiling_kernel):
    """The same step with the reads resident in TWO BITS PER BASE (fk_count_device_packed): 37.5 GB instead of 150 GB,
    so that every super-mer record fits beside them -- one split pass, no replay -- and the splitter's tile loader
    converts nothing.  The caller's ASCII reads are packed on the device first (not timed: the form is the leg's
    input) and released; returns the record for the JSON line."""
    import hashlib
    nbases = per * L
    ctx = fastk_amd.Context(kmer=args.kmer, table_cutoff=cfg["cutoff"], nthreads=4, device=local_rank,
                            nbuckets=max(1, cfg["buckets"]), split_passes=1)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    for kv in args.debug:
        key, val = kv.split("=")
        ctx.debug_set(key, int(val))
    try:
        return _packed_resident_leg(args, cfg, ctx, torch, dev, reads, nbytes, per, L, ceiling, ceiling_kernel)
    finally:
        ctx.close()
        torch.cuda.empty_cache()


def _packed_resident_leg(args, cfg, ctx, torch, dev, reads, nbytes, per, L, ceiling, ceiling_kernel):
    import hashlib
    nbases = per * L
    if cfg["buckets"] > 1:
        train_buckets(ctx, reads, nbytes)
    codes = torch.empty((nbases + 3) // 4 + 64, dtype=torch.uint8, device=dev)
    ctx._ck(ctx.L.fk_pack_fixed_reads(ctx.h, reads.data_ptr(), per, L, codes.data_ptr()))
    roff = torch.arange(per + 1, dtype=torch.int64, device=dev) * L
    torch.cuda.synchronize()
    reads.set_(torch.empty(0, dtype=torch.uint8, device=dev).untyped_storage())    # the ASCII form goes
    torch.cuda.empty_cache()

    def step():
        return ctx.count_device_packed(codes.data_ptr(), nbases, roff.data_ptr(), per, None, 0, fetch_table=False)

    last = None
    for i in range(args.warmup):
        t0 = time.perf_counter()
        last = step()
        log(args, "packed-resident warm-up step %d: %.3f s" % (i, time.perf_counter() - t0))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    h = last.hist.astype(np.int64)
    conserved = int((h[1:0x7fff] * np.arange(1, 0x7fff)).sum()) + int(last.max_inst)
    assert conserved == last.ninst == per * (L - args.kmer + 1), \
        "packed-resident leg: %d instances, %d in the histogram, %d expected" % (last.ninst, conserved, per * (L - args.kmer + 1))
    log(args, "packed-resident: %.3f s per step" % dt, last.ms)
    out = dict(metric="canonical k-mers/sec (k=%d, whole hot path, reads resident in HBM in two bits per base)" % args.kmer,
               value=last.ninst / dt, unit="k-mers/s", ms_per_step=1e3 * dt, steps=args.steps,
               resident_bytes=int(codes.numel() + roff.numel() * 8),
               split_passes=int(last.split_passes), replay_passes=int(last.replay_passes),
               buckets=int(last.buckets_counted), supermers=int(last.nsuper), weighted_kmers=int(last.nweighted),
               table_entries=int(last.ntable), histogram_sha256=hashlib.sha256(h.tobytes()).hexdigest(),
               stage_ms=dict((k, round(v, 3)) for k, v in last.ms.items()),
               roofline=roofline_record(last, ctx.w, 2 if cfg["buckets"] > 1 else 1, ceiling, ceiling_kernel))
    out["stage_ms"]["table_sort"] = round(last.ms_table_sort, 3)
    return out


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

    cfg_id = args.config or (3 if world > 1 else (1 if sharded else 2))
    cfg = dict(CONFIGS[cfg_id])
    for key, val in (("genome_mbp", args.genome_mbp), ("coverage", args.coverage), ("read_len", args.read_len),
                     ("err_ppm", args.err_ppm), ("cutoff", args.cutoff), ("buckets", args.stream_buckets),
                     ("split_passes", args.split_passes), ("cpu_sample_mbp", args.cpu_sample_mbp)):
        if val is not None:
            cfg[key] = val
    custom = any(v is not None for v in (args.genome_mbp, args.coverage, args.read_len, args.err_ppm, args.cutoff))
    args.device_budget_gb = 280.0 if cfg_id == 2 else 0.0
    args.e2e_threads = 32 if cfg_id == 2 else 4
    args.e2e_mem_gb = 256 if cfg_id == 2 else 64

    if cfg_id == 3 or sharded:
        # every sharded run goes through the C engine (fk_shard_*): configs[3] -- the configs[2] set striped over the GPUs,
        # strong scaling -- or, with --config 1, configs[1]'s workload PER GPU (weak scaling: genome and reads grow with N)
        return main_config3(args, cfg, torch, fastk_amd, dist if sharded else None, rank, local_rank, world, dev,
                            weak=(cfg_id != 3))

    L = cfg["read_len"]
    glen = int(cfg["genome_mbp"] * 1e6 * args.scale) * world
    total_reads = int(cfg["coverage"] * glen / L)
    per = total_reads // world
    first = rank * per
    nbytes = per * (L + 1)

    nbuckets = world * max(1, args.exchange_rounds) if sharded else max(1, cfg["buckets"])
    passes = 1 if sharded else max(1, cfg["split_passes"])
    ctx = fastk_amd.Context(kmer=args.kmer, table_cutoff=cfg["cutoff"], nthreads=4, device=local_rank,
                            nbuckets=nbuckets, split_passes=passes)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    for kv in args.debug:
        key, val = kv.split("=")
        ctx.debug_set(key, int(val))
    ceiling, ceiling_kernel = copy_ceiling(torch, dev, ctx) if rank == 0 else (None, None)
    log(args, "copy ceiling", ceiling, "GB/s (hipMemcpy),", ceiling_kernel, "GB/s (copy kernel); generating", nbytes, "bytes of reads")
    reads = torch.empty(nbytes + 64, dtype=torch.uint8, device=dev)
    ctx._ck(ctx.L.fk_synth_reads(ctx.h, args.seed, glen, L, cfg["err_ppm"], first, per,
                                 reads.data_ptr()))
    torch.cuda.synchronize()
    assert not sharded                                 # (every sharded run went to main_config3 above: one rank, one context here)
    if nbuckets > 1:
        train_buckets(ctx, reads, nbytes)              # (scheme set-up on a sample, like Determine_Scheme: not a step)

    def step(verify=False):
        return ctx.count_device_reads(reads.data_ptr(), nbytes, fetch_table=False)

    def barrier():
        torch.cuda.synchronize()
        if sharded:
            dist.barrier()
            torch.cuda.synchronize()

    last = None
    for i in range(args.warmup):
        t0 = time.perf_counter()
        last = step(verify=True)
        log(args, "warm-up step %d: %.3f s" % (i, time.perf_counter() - t0))
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
        ntable = last.ntable
        h = last.hist.astype(np.int64)
        conserved = int((h[1:0x7fff] * np.arange(1, 0x7fff)).sum()) + int(last.max_inst)
        assert conserved == ninst, "conservation law violated: %d != %d" % (conserved, ninst)
    else:
        loc = last["local"]["result"]
        ninst, nsuper = last["ninst"], last["nsuper"]
        nweighted, ndistinct = last["nweighted"], last["ndistinct"]
        ntable = None
    expect = total_reads // world * world * (L - args.kmer + 1)
    assert ninst == expect, "k-mer instance count %d != %d" % (ninst, expect)

    ms_step = 1e3 * dt / args.steps
    value = ninst / (dt / args.steps)
    log(args, "timed: %.3f s per step" % (dt / args.steps), loc.ms)
    # the N > 1 lines end with the sorted table in pinned host memory (the final gather): the same boundary at N = 1
    # is the step + the D2H of its table -- one warm-up call (it pins the host buffer), then one timed
    on_host = None
    if not sharded and cfg["cutoff"] > 0:
        import ctypes as C
        r_h = fastk_amd.api.CResult()          # (the C call itself: the Python wrapper would copy the 36 GB table once more)
        ctx._ck(ctx.L.fk_count_device_reads(ctx.h, reads.data_ptr(), nbytes, 1, C.byref(r_h)))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx._ck(ctx.L.fk_count_device_reads(ctx.h, reads.data_ptr(), nbytes, 1, C.byref(r_h)))
        torch.cuda.synchronize()
        dt_h = time.perf_counter() - t0
        assert r_h.ntable == last.ntable and bool(r_h.table)
        on_host = dict(value=ninst / dt_h, unit="k-mers/s", ms_per_step=round(1e3 * dt_h, 3),
                       d2h_bytes=int(r_h.ntable) * ctx.w.kmer_word,
                       definition="the step + its sorted table brought to pinned host memory: the boundary of the N > 1 "
                                  "lines (their final gather), for a scaling curve computed against this N = 1 point")

    w = ctx.w
    roofline = roofline_record(loc, w, cfg_id, ceiling, ceiling_kernel)

    scale_note = "" if args.scale == 1.0 else " SCALED by %g (development run)" % args.scale
    workload = cfg["label"] % (cfg["genome_mbp"] * args.scale, " per GPU" if sharded else "", args.kmer) + scale_note
    if custom:
        workload = "custom: " + workload
    out = dict(metric="canonical k-mers/sec (k=%d, whole hot path, reads resident in HBM as 0-terminated ASCII)" % args.kmer,
               value=value, unit="k-mers/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
               ms_per_step=ms_step, higher_is_better=True,
               # N = 1 is the first point of the curve whose N > 1 points stripe the SAME read set over the GPUs
               # (configs[3]): total work fixed.  (--config 1 with N > 1 is a per-GPU workload: weak.)
               scaling="weak" if (sharded and cfg_id == 1) else "strong", vs_baseline=None,
               dtype="u8", data="synthetic",
               config=dict(workload=workload, reads_per_gpu=per, bases_per_gpu=per * L,
                           kmer_instances=int(ninst), supermers=int(nsuper),
                           weighted_kmers=int(nweighted), distinct_kmers=int(ndistinct),
                           table_entries=ntable, table_cutoff=cfg["cutoff"],
                           parallelism=("minimizer-bucket shard x%d" % world
                                        + (", exchange in %d overlapped rounds" % args.exchange_rounds
                                           if args.exchange_rounds > 1 else "")) if sharded else
                                       "1 GPU, %d minimizer buckets counted one after the other, %d split pass(es) "
                                       "over the resident reads" % (loc.buckets_counted, loc.split_passes)),
               roofline=roofline,
               stage_ms=dict((k, round(v, 3)) for k, v in loc.ms.items()))
    if on_host is not None:
        out["value_with_table_on_host"] = on_host
    if not sharded:
        out["stage_ms"]["table_sort"] = round(loc.ms_table_sort, 3)
        out["histogram_sha256"] = __import__("hashlib").sha256(h.tobytes()).hexdigest()
        out["hist_file_sha256"] = hist_file_sha256(args.kmer, last.hist, last.max_inst)    # what <root>.hist would hold
    if rank == 0 and world == 1 and not args.force_shard and not args.no_packed_leg:
        ctx.close()
        ctx = None
        try:
            out["value_packed_resident"] = packed_resident_leg(args, cfg, fastk_amd, torch, dev, local_rank, reads, nbytes,
                                                               per, L, ceiling, ceiling_kernel)
        except Exception as e:                                   # (a resource failure of the leg; a WRONG result is not one)
            out["value_packed_resident"] = dict(failed=repr(e)[-600:])
        # parity of the two input forms is not a leg that may fail quietly: another histogram ends the run without a line
        if "histogram_sha256" in out["value_packed_resident"]:
            assert out["value_packed_resident"]["histogram_sha256"] == out["histogram_sha256"], \
                "the packed-resident step gives another histogram than the ASCII-resident one"
        log(args, "packed-resident leg", out["value_packed_resident"])
    del reads
    torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not args.force_shard:
        if ctx is not None:
            ctx.close()
        ctx = None
        gen = fastk_amd.Context(kmer=args.kmer, device=local_rank)
        if not args.no_device_leg:
            t0 = time.perf_counter()
            try:
                out["value_device"] = device_leg(args, cfg, fastk_amd, gen, glen, per, L, local_rank)
            except Exception as e:                               # a leg that fails must not take the line with it
                out["value_device"] = dict(failed=repr(e)[-600:])
            log(args, "device leg %.1f s" % (time.perf_counter() - t0), out["value_device"])
        if not args.no_e2e:
            t0 = time.perf_counter()
            try:
                out["value_e2e"] = e2e_leg(args, cfg, fastk_amd, gen, L)
            except Exception as e:
                out["value_e2e"] = dict(failed=repr(e)[-600:])
            log(args, "e2e leg %.1f s" % (time.perf_counter() - t0), out["value_e2e"])
        gen.close()
        if not args.no_e2e:
            try:
                out["value_e2e_reference_main"] = shim_leg(args, cfg)
            except Exception as e:
                out["value_e2e_reference_main"] = dict(failed=repr(e)[-600:])
        if not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args, cfg)
            except Exception as e:
                out["cpu_baseline"] = dict(failed=repr(e)[-600:])
    if rank == 0:
        print(json.dumps(out))
    if ctx is not None:
        ctx.close()
    if sharded:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
