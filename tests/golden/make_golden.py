#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by running the REFERENCE FastK
(oracle/_ref/FastK, built from /root/reference by oracle/Makefile) on deterministic inputs.

Run from the repo root in the build container (needs /root/reference):
    python tests/golden/make_golden.py

Each case <name> produces
    <name>.json      input description + expected results from the reference:
                     sparse .hist (nonzero bins, ilow, ihigh, sha256 of the 262,164 file bytes),
                     .ktab header fields, nels, part sizes, sha256 of the canonical stream
                     (stub index from byte 16 + part payloads from byte 12), sha256 of every file,
                     digest of the decoded -p profiles (orc.profiles_digest)
    <name>.fa.gz     the input reads, only for hand-built edge-case inputs (synthetic inputs are
                     regenerated from include/fk_synth.h parameters)
    <name>.table.gz  the full (k-mer bytes, count) table, only for small cases
Fixtures are data (inputs + expected outputs); no reference source text is stored.
"""
import gzip
import hashlib
import json
import os
import random
import shutil
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def edge_reads(seed, k):
    """Both strands, N's (single and runs), reads < k, == k, k+1, lowercase/uppercase,
    a read duplicated 40,000 times (weight clipping + saturation) and poly-A reads."""
    rnd = random.Random(seed)
    g = "".join(rnd.choice("acgt") for _ in range(5000))
    comp = {"a": "t", "c": "g", "g": "c", "t": "a", "n": "n"}
    reads = []
    for _ in range(3000):
        L = rnd.choice([30, k - 1, k, k + 1, 100, 151, 400])
        s = rnd.randrange(0, len(g) - L)
        r = g[s:s + L]
        if rnd.random() < 0.5:
            r = "".join(comp[c] for c in reversed(r))
        r = list(r)
        for j in range(len(r)):
            if rnd.random() < 0.002:
                r[j] = rnd.choice("acgt")
        if rnd.random() < 0.1:
            r[rnd.randrange(len(r))] = "N"
        if rnd.random() < 0.03:
            a = rnd.randrange(len(r))
            for j in range(a, min(len(r), a + rnd.randrange(1, 80))):
                r[j] = "n"
        r = "".join(r)
        if rnd.random() < 0.3:
            r = r.upper()
        reads.append(r)
    reads += [g[100:160]] * 40000
    reads += ["a" * 120] * 500
    return reads


CASES = [
    # name, kind, k, cutoff, threads, spec
    dict(name="edge_k40_t1_T4", kind="edge", k=40, cutoff=1, T=4, seed=47, fmt="fasta"),
    dict(name="edge_k51_t1_T4", kind="edge", k=51, cutoff=1, T=4, seed=58, fmt="fastq"),
    dict(name="edge_k40_t4_T1", kind="edge", k=40, cutoff=4, T=1, seed=47, fmt="fasta"),
    dict(name="edge_k21_t2_T3", kind="edge", k=21, cutoff=2, T=3, seed=28, fmt="fasta"),
    dict(name="synth_illumina_k40_t1_T4", kind="synth", k=40, cutoff=1, T=4, fmt="fastq",
         synth=dict(seed=11, genome_len=200000, read_len=150, err_ppm=1000, nreads=20000)),
    dict(name="synth_illumina_k51_t1_T4", kind="synth", k=51, cutoff=1, T=4, fmt="fasta",
         synth=dict(seed=12, genome_len=200000, read_len=150, err_ppm=1000, nreads=20000)),
    dict(name="synth_hifi_k40_t4_T8", kind="synth", k=40, cutoff=4, T=8, fmt="fasta",
         synth=dict(seed=13, genome_len=100000, read_len=15000, err_ppm=2000, nreads=333)),
    dict(name="synth_tiny_k40_t1_T2", kind="synth", k=40, cutoff=1, T=2, fmt="fasta",
         synth=dict(seed=14, genome_len=3000, read_len=100, err_ppm=5000, nreads=300)),
]


# Digest-only cases above fixture size (no table file, no profiles): BASELINE.json configs[0] exactly
# (1 M x 150 bp reads of a 10 Mbp genome, 0.1 % errors, FASTQ, -k40 -t1 -T4) and the two bounded
# samples bench.py times the reference on (50x of a 20 Mbp genome, Illumina- and HiFi-shaped).
# `python tests/golden/make_golden.py --large` regenerates only these (minutes of reference time).
LARGE_CASES = [
    dict(name="configs0_k40_t1_T4", kind="large", k=40, cutoff=1, T=4, fmt="fastq",
         synth=dict(seed=20251001, genome_len=10000000, read_len=150, err_ppm=1000, nreads=1000000)),
    dict(name="illumina50x20M_k40_t1_T4", kind="large", k=40, cutoff=1, T=4, fmt="fastq",
         synth=dict(seed=20251001, genome_len=20000000, read_len=150, err_ppm=1000, nreads=6666666)),
    dict(name="hifi50x20M_k40_t4_T4", kind="large", k=40, cutoff=4, T=4, fmt="fasta",
         synth=dict(seed=20251001, genome_len=20000000, read_len=15000, err_ppm=2000, nreads=66666)),
    # the reference with a small sort memory: NPARTS = 2 (unpadded trie, drand48 deal), 3, and 9 (the trie is padded)
    # buckets -- what the hidden part files then look like depends on its scheme (split.c:289-381,437-472,617-766)
    dict(name="configs0_k40_t1_T4_M1", kind="large", k=40, cutoff=1, T=4, fmt="fastq", ref_extra=["-M1"],
         synth=dict(seed=20251001, genome_len=10000000, read_len=150, err_ppm=1000, nreads=1000000)),
    dict(name="illumina50x20M_k40_t1_T4_M3", kind="large", k=40, cutoff=1, T=4, fmt="fastq", ref_extra=["-M3"],
         synth=dict(seed=20251001, genome_len=20000000, read_len=150, err_ppm=1000, nreads=6666666)),
    dict(name="illumina50x20M_k40_t1_T4_M1", kind="large", k=40, cutoff=1, T=4, fmt="fastq", ref_extra=["-M1"],
         synth=dict(seed=20251001, genome_len=20000000, read_len=150, err_ppm=1000, nreads=6666666)),
    # 10 M reads: the first block is two thirds of the input (ratio 1.5, io.c:528), 14 buckets, and seven heavy
    # minimizers are padded to 7 bases (PAD = 2, 1052 trie states)
    dict(name="illumina50x30M_k40_t1_T4_M1", kind="large", k=40, cutoff=1, T=4, fmt="fastq", ref_extra=["-M1"],
         synth=dict(seed=20251001, genome_len=30000000, read_len=150, err_ppm=1000, nreads=10000000)),
]

# Above 4 GiB (`--huge`, ~15 min of reference time, 10 GB input file): 50x of a 200 Mbp genome in
# 15 kbp reads -- 10 G bases, every device buffer of the bench's configuration beyond 2^32 bytes.
# The input is streamed to disk in pieces; the reads are regenerated on the device by
# fk_synth_reads, so only the digests travel.
HUGE_CASES = [
    dict(name="hifi50x200M_k40_t4_T8", kind="huge", k=40, cutoff=4, T=8, fmt="fasta",
         synth=dict(seed=20251001, genome_len=200000000, read_len=15000, err_ppm=2000, nreads=666666)),
    # BASELINE configs[1] at FULL size (50x of a 100 Mbp genome in 150 bp reads: 33.3 M reads, 3.70 G k-mer
    # instances; the reads test_full_size_properties_configs1 generates on the device)
    dict(name="configs1_k40_t1_T4", kind="huge", k=40, cutoff=1, T=4, fmt="fasta",
         synth=dict(seed=20251001, genome_len=100000000, read_len=150, err_ppm=1000, nreads=33333333)),
    # BASELINE configs[4] at the stated scale of 1/1000 (100x of a 32 Mbp genome, k=51 -t1 -p: 21.3 M reads;
    # the reads of test_configs4_scaled_slice_k51_profiles_with_spill), with the digest of the decoded profiles
    dict(name="configs4slice_k51_t1_T4_p", kind="huge", k=51, cutoff=1, T=4, fmt="fasta", prof=True,
         synth=dict(seed=4051, genome_len=32000000, read_len=150, err_ppm=1000, nreads=21333333)),
]


def sha(b):
    return hashlib.sha256(b).hexdigest()


def sha_file(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        while True:
            b = f.read(1 << 24)
            if not b:
                break
            h.update(b)
    return h.hexdigest()


def main():
    if not orc.have_ref():
        orc.build(ref=True)
    cases = LARGE_CASES if "--large" in sys.argv else (CASES + LARGE_CASES if "--all" in sys.argv else CASES)
    if "--huge" in sys.argv:
        cases = HUGE_CASES
    only = [a[7:] for a in sys.argv if a.startswith("--only=")]
    if only:
        cases = [c for c in CASES + LARGE_CASES + HUGE_CASES if c["name"] in only]
    for case in cases:
        name = case["name"]
        k = case["k"]
        if case["kind"] == "edge":
            reads = edge_reads(case["seed"], k)
            bases, boff = orc.block_from_reads(reads)
            with gzip.GzipFile(os.path.join(HERE, name + ".fa.gz"), "wb", mtime=0) as f:
                for i, r in enumerate(reads):
                    f.write(b">r%d\n%s\n" % (i, r.encode()))
        elif case["kind"] == "huge":
            bases = boff = None
        else:
            s = case["synth"]
            bases, boff = orc.synth_block(s["seed"], s["genome_len"], s["read_len"], s["err_ppm"],
                                          0, s["nreads"])
        d = tempfile.mkdtemp(prefix="fkgold", dir=os.environ.get("FK_GOLDEN_TMP"))
        path = os.path.join(d, "x." + case["fmt"])
        if case["kind"] == "huge":
            s = case["synth"]
            with open(path, "wb") as f:
                for r0 in range(0, s["nreads"], 20000):
                    nr = min(20000, s["nreads"] - r0)
                    b, _ = orc.synth_block(s["seed"], s["genome_len"], s["read_len"], s["err_ppm"],
                                           r0, nr)
                    L = s["read_len"]
                    mat = np.empty((nr, 3 + L + 1), dtype=np.uint8)
                    mat[:, 0:3] = np.frombuffer(b">r\n", dtype=np.uint8)
                    mat[:, 3:3 + L] = b.reshape(nr, L + 1)[:, :L]
                    mat[:, 3 + L] = ord("\n")
                    mat.tofile(f)
        elif case["fmt"] == "fasta":
            orc.write_fasta(path, bases, boff, width=0 if case["kind"] == "edge" else 100)
        else:
            orc.write_fastq(path, bases, boff)
        orc.run_ref_fastk(path, k, case["cutoff"], case["T"], d,
                          extra=tuple(case.get("ref_extra", ())) + (("-p",) if case.get("prof") else ()))
        h = orc.read_hist(os.path.join(d, "x.hist"))
        t = orc.read_ktab(os.path.join(d, "x"))
        nz = np.nonzero(h["hist"])[0]
        files = ["x.hist", "x.ktab"] + [".x.ktab.%d" % (i + 1) for i in range(case["T"])]
        exp = dict(
            hist_sha256=sha(h["raw"]), hist_len=len(h["raw"]), ilow=int(h["ilow"]),
            ihigh=int(h["ihigh"]),
            hist_nonzero=[[int(i) + h["low"], int(h["hist"][i])] for i in nz],
            ktab=dict(kmer=t["kmer"], nparts=t["nparts"], minval=t["minval"], ibytes=t["ibytes"],
                      nels=t["nels"], part_sizes=t["part_sizes"],
                      stream_sha256=t["stream_sha256"],
                      first=[t["table"][i].tobytes().hex() for i in range(min(8, t["nels"]))],
                      last=[t["table"][i].tobytes().hex()
                            for i in range(max(0, t["nels"] - 8), t["nels"])]),
            file_sha256={f: sha_file(os.path.join(d, f)) for f in files},
        )
        if case.get("prof"):          # a -p run above fixture size: the decoded profiles by the C decoder, part by part
            nr, nb, npos, dig = orc.profiles_digest_files(d, "x")
            exp["prof"] = dict(nreads=nr, ref_bytes=nb, kmer_positions=npos, decoded_sha256=dig)
        if case["kind"] in ("large", "huge"):
            meta = dict(case)
            meta["expected"] = exp
            meta["generated_by"] = "tests/golden/make_golden.py --large with oracle/_ref/FastK (reference build)"
            with open(os.path.join(HERE, name + ".json"), "w") as f:
                json.dump(meta, f, indent=1, sort_keys=True)
            shutil.rmtree(d)
            print(name, "nels", t["nels"], "parts", t["part_sizes"], "ihigh", h["ihigh"])
            continue
        # profiles: the reference's -p run (own directory, same input); what is pinned is the DECODED
        # count vectors (the reference's bytes depend on its internal super-mer cuts)
        pd = tempfile.mkdtemp(prefix="fkgoldp")
        ppath = os.path.join(pd, "x." + case["fmt"])
        shutil.copy(path, ppath)
        orc.run_ref_fastk(ppath, k, case["cutoff"], case["T"], pd, extra=("-p",))
        pk, enc = orc.read_profiles(pd, "x")
        pfiles = sorted(f for f in os.listdir(pd) if f == "x.prof" or f.startswith(".x.prof.") or f.startswith(".x.pidx."))
        exp["prof"] = dict(nreads=len(enc), ref_bytes=sum(len(e) for e in enc),
                           decoded_sha256=orc.profiles_digest([orc.profile_decode(e) for e in enc]),
                           first_ref_hex=[e.hex() for e in enc[:4]],
                           file_sha256={f: sha_file(os.path.join(pd, f)) for f in pfiles})
        shutil.rmtree(pd)
        meta = dict(case)
        meta["expected"] = exp
        meta["generated_by"] = "tests/golden/make_golden.py with oracle/_ref/FastK (reference build)"
        with open(os.path.join(HERE, name + ".json"), "w") as f:
            json.dump(meta, f, indent=1, sort_keys=True)
        if t["nels"] <= 40000:
            with gzip.GzipFile(os.path.join(HERE, name + ".table.gz"), "wb", mtime=0) as f:
                f.write(np.ascontiguousarray(t["table"]).tobytes())
        shutil.rmtree(d)
        print(name, "nels", t["nels"], "parts", t["part_sizes"], "ihigh", h["ihigh"])


if __name__ == "__main__":
    main()
