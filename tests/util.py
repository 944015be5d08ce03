"""Shared helpers for the tests: golden-case loading and result comparison."""
import glob
import gzip
import json
import os

import numpy as np

from oracle import orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def no_reference(what):
    """A test that checks against the reference itself found no reference build (oracle/_ref): that is a FAILURE
    wherever the builds are expected -- in this repository's container (they are made from /root/reference by
    oracle/Makefile) and on the GPU box (they travel with the snapshot) -- because a skipped reference test is an
    untested claim.  FK_REQUIRE_REF=0 turns it back into a skip (a clone on a box that never had the sources)."""
    import pytest
    if os.environ.get("FK_REQUIRE_REF", "1") == "1":
        pytest.fail(what + " -- build it with `make -C oracle ref` where /root/reference is mounted, or set FK_REQUIRE_REF=0")
    pytest.skip(what)


FIXTURE_KINDS = ("edge", "synth")        # cases small enough to be loaded, run through the oracle and parametrised over


def golden_names(kind=None):
    """kind None: every fixture-sized case.  The digest-only cases ("large", "huge", and "full" = BASELINE configs[2] at
    full size: 150 G bases) are asked for by name or kind -- never by default: a parametrised test that loads one
    synthesises its reads on the host.  (Round 6: the new "full" fixture slipped into this list, every golden test
    then tried to build 150 GB of reads in host memory, and two GPU boxes were lost to it; now an allow-list, and
    load_case refuses anything above 2 G bases.)"""
    out = []
    for p in sorted(glob.glob(os.path.join(GOLDEN, "*.json"))):
        c = json.load(open(p))
        if (kind is None and c["kind"] in FIXTURE_KINDS) or c["kind"] == kind:
            out.append(c["name"])
    return out


def load_case(name):
    """Return (case dict, bases, boff) for a golden fixture."""
    case = json.load(open(os.path.join(GOLDEN, name + ".json")))
    if case["kind"] == "edge":
        reads = []
        with gzip.open(os.path.join(GOLDEN, name + ".fa.gz"), "rb") as f:
            for line in f:
                if not line.startswith(b">"):
                    reads.append(line.rstrip(b"\n"))
        bases, boff = orc.block_from_reads(reads)
    else:
        s = case["synth"]
        if s["nreads"] * (s["read_len"] + 1) > (2 << 30):
            raise ValueError("%s: %d reads of %d bases are not a fixture to load on the host (digest-only case)"
                             % (name, s["nreads"], s["read_len"]))
        bases, boff = orc.synth_block(s["seed"], s["genome_len"], s["read_len"], s["err_ppm"], 0,
                                      s["nreads"])
    return case, bases, boff


def driver_exe(name):
    """fastk_amd/bin/<name> (FastK_amd, Fastmerge_amd) -- or, in a test process started with FASTK_AMD_EMU=1, the same C
    source linked against the CPU-emulated library (tests/csrc/build_emu_lib.py): test infrastructure, see tests/conftest.py"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if os.environ.get("FASTK_AMD_EMU") == "1":
        import sys
        sys.path.insert(0, os.path.join(root, "tests", "csrc"))
        import build_emu_lib
        return build_emu_lib.build_driver(name)
    return os.path.join(root, "fastk_amd", "bin", name)


def golden_table(name, kmer):
    p = os.path.join(GOLDEN, name + ".table.gz")
    if not os.path.exists(p):
        return None
    kw = ((2 * kmer + 7) >> 3) + 2
    return np.frombuffer(gzip.open(p, "rb").read(), dtype=np.uint8).reshape(-1, kw)


def expected_hist(case):
    h = np.zeros(0x8000, dtype=np.int64)
    for i, v in case["expected"]["hist_nonzero"]:
        h[i] = v
    return h


def check_against_golden(case, hist, max_inst, table, name=None):
    """hist: int64[0x8000] (index = count), table: (n, KMER_BYTES+2) uint8 sorted entries."""
    import hashlib
    exp = case["expected"]
    k = case["k"]
    eh = expected_hist(case)
    assert np.array_equal(np.asarray(hist)[1:], eh[1:]), "histogram differs from reference"
    assert int(max_inst) == exp["ihigh"], "ihighcnt differs from reference"
    raw = orc.hist_file_bytes(k, hist, max_inst)
    assert len(raw) == exp["hist_len"] == 262164
    assert hashlib.sha256(raw).hexdigest() == exp["hist_sha256"], ".hist bytes differ"
    kt = exp["ktab"]
    assert table.shape[0] == kt["nels"], "table entry count differs"
    assert orc.idx_bytes(k, table.shape[0]) == kt["ibytes"]
    assert orc.table_stream_sha256(k, table) == kt["stream_sha256"], ".ktab canonical stream differs"
    gt = golden_table(name or case["name"], k)
    if gt is not None:
        assert np.array_equal(table, gt), "table entries differ from reference"


def write_fastx(path, bases, boff, fastq):
    """Vectorised writer for equal-length reads (the synthetic cases): one line per read."""
    n = len(boff) - 1
    L = int(boff[1] - boff[0]) - 1
    rows = np.asarray(bases).reshape(n, L + 1)[:, :L]
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
    mat.tofile(path)


def sha_file(path):
    import hashlib
    h = hashlib.sha256()
    with open(path, "rb") as f:
        while True:
            b = f.read(1 << 24)
            if not b:
                break
            h.update(b)
    return h.hexdigest()
