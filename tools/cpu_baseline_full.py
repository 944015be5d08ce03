#!/usr/bin/env python3
"""Reference FastK on BASELINE configs[2] ITSELF, once: the full 150 G-base FASTA (50x of a 3 Gbp genome in 15 kbp reads,
the reads of include/fk_synth.h written by the GPU) in /dev/shm, `FastK -k40 -t4 -T<n>` from oracle/_ref with its
temporary files on the same RAM disk, wall time from process start to exit.  Checks the .hist it writes (sum over the
histogram = instances) and records its sha256 -- the same bytes bench.py's hist_file_sha256 stands for, so that the
GPU path and the reference can be compared at the full size.  Needs ~400 GB of RAM (the file 150 GB, the reference's
bit-stuffed super-mer files and table parts ~100 GB, its sort memory 12 GB).

  python tools/cpu_baseline_full.py [--threads 32] [--scale 1.0] > profiles/r05_cpu_baseline_configs2_full.json
  python tools/cpu_baseline_full.py --golden tests/golden/configs2_k40_t4.json     (also writes the golden fixture that
        tests/test_gpu_parity.py::test_full_size_properties_configs2 asserts: .hist sha256, .ktab canonical-stream
        sha256, entries -- digests of what oracle/_ref/FastK wrote; the reads are regenerated from the synth spec)
  python tools/cpu_baseline_full.py --golden-from profiles/r05_cpu_baseline_configs2_full.json --golden <path>
        (the same fixture from the record of an earlier run of this script, without running the reference again)
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def write_golden(path, rec, source):
    """tests/golden/configs2_k40_t4.json: what reference FastK (oracle/_ref/FastK, -k40 -t4) left for BASELINE configs[2]
    at full size -- digests only; the 150 G bases are fk_synth_reads(seed, genome_len, read_len, err_ppm)."""
    assert rec["scale"] == 1.0 and rec["conserved"]
    L = 15000
    glen = 3000000000
    nreads = int(50 * glen / L)
    case = dict(name="configs2_k40_t4", kind="full", k=40, cutoff=4, T=rec["cores"], fmt="fasta",
                synth=dict(seed=20251001, genome_len=glen, read_len=L, err_ppm=2000, nreads=nreads),
                generated_by="tools/cpu_baseline_full.py --golden (reference FastK -k40 -t4 -T%d run on the GPU box's host, "
                             "%.1f s; record %s)" % (rec["cores"], rec["seconds"], source),
                expected=dict(hist_len=rec["hist_len"], hist_sha256=rec["hist_file_sha256"],
                              kmer_instances=rec["kmer_instances"],
                              ktab=dict(nels=rec["table_entries"], stream_sha256=rec["ktab_stream_sha256"], ibytes=3)))
    with open(path, "w") as f:
        json.dump(case, f, indent=1, sort_keys=True)
        f.write("\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=32)
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--kmer", type=int, default=40)
    ap.add_argument("--golden", default=None, help="write the golden fixture of this run to this path")
    ap.add_argument("--golden-from", default=None, help="make the fixture from an earlier run's record instead of running")
    args = ap.parse_args()
    if args.golden_from:
        write_golden(args.golden, json.load(open(args.golden_from)), os.path.basename(args.golden_from))
        return
    import bench
    import fastk_amd
    from oracle import orc
    mem_gb = os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_PHYS_PAGES") / 1e9
    need = 400 * args.scale
    if mem_gb < need:
        print(json.dumps(dict(skipped="the box has %.0f GB of RAM, the run needs ~%.0f" % (mem_gb, need))))
        return
    if not orc.have_ref():
        print(json.dumps(dict(skipped="oracle/_ref/FastK is not built")))
        return
    L, k = 15000, args.kmer
    glen = int(3000e6 * args.scale)
    nreads = int(50 * glen / L)
    inst = nreads * (L - k + 1)
    d = tempfile.mkdtemp(prefix="fkfull", dir="/dev/shm")
    try:
        path = os.path.join(d, "reads.fasta")
        ctx = fastk_amd.Context(kmer=k)
        t0 = time.perf_counter()
        bench.write_synth_file(ctx, path, False, 20251001, glen, L, 2000, nreads)
        ctx.close()
        t_gen = time.perf_counter() - t0
        cmd = [os.path.join(orc.REF_DIR, "FastK"), "-k%d" % k, "-t4", "-T%d" % args.threads, "-P" + d, path]
        t0 = time.perf_counter()
        p = subprocess.run(cmd, cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
        dt = time.perf_counter() - t0
        if p.returncode != 0:
            print(json.dumps(dict(failed=p.stderr[-500:], seconds=round(dt, 1))))
            return
        raw = open(os.path.join(d, "reads.hist"), "rb").read()
        kk, lo, hi = np.frombuffer(raw[:12], dtype=np.int32)
        ilow, ihigh = np.frombuffer(raw[12:28], dtype=np.int64)
        hist = np.frombuffer(raw[28:], dtype=np.int64)
        conserved = int((hist[:-1] * np.arange(lo, hi)).sum()) + int(ihigh)
        parts = sorted(f for f in os.listdir(d) if f.startswith(".reads.ktab."))
        nels = 0
        for f in parts:
            with open(os.path.join(d, f), "rb") as fh:
                fh.seek(4)
                nels += int(np.frombuffer(fh.read(8), dtype=np.int64)[0])
        t0 = time.perf_counter()
        stream, nels2, nparts = bench.ktab_stream_sha256(d, "reads")
        t_dig = time.perf_counter() - t0
        assert nels2 == nels and nparts == len(parts)
        rec = dict(
            ktab_stream_sha256=stream, ktab_parts=nparts, stream_digest_seconds=round(t_dig, 1),
            value=inst / dt, unit="k-mers/s", kind="reference", cores=args.threads, host_threads=os.cpu_count(),
            seconds=round(dt, 1), kmer_instances=inst, instances_in_histogram=conserved,
            conserved=(conserved == inst), hist_len=len(raw), hist_file_sha256=hashlib.sha256(raw).hexdigest(),
            table_entries=nels, scale=args.scale, input_bytes=os.path.getsize(path), file_written_in_s=round(t_gen, 1),
            sample="BASELINE configs[2] itself%s: 50x of a %g Mbp genome in %d reads of %d bp, err 2000 ppm, FASTA in /dev/shm; "
                   "reference FastK -k%d -t4 -T%d -P<same RAM disk>, process start to exit"
                   % ("" if args.scale == 1.0 else " SCALED by %g" % args.scale, glen / 1e6, nreads, L, k, args.threads))
        print(json.dumps(rec))
        if args.golden and args.scale == 1.0 and k == 40:
            write_golden(args.golden, rec, "this run")
    finally:
        subprocess.run(["rm", "-rf", d])


if __name__ == "__main__":
    main()
