"""FastK_amd -x -p deals the reads of FASTA / FASTQ input to the input threads the reference would use (input_threads,
io_nearest in fastk_amd/csrc/host/FastK_amd.c; io.c:409-490, 2340-2521), because the reference's .prof parts are those
threads' read ranges.  Here the host side alone -- tests/csrc/input_threads_print.c includes the driver's source, calls
input_threads and runs the host scanner with fk_push_block replaced by a counter -- against
the reference run on the same files: the (first read, reads) of every .pidx part it writes.  Random files with what
misleads a search for record starts: wrapped sequence lines, '>' and '@' inside headers, FASTQ quality lines that begin
with '@' or '+', several files of different sizes, compressed files, more and fewer threads than the data carries.
No GPU needed (the reference runs on the CPU; the driver's function touches no device)."""
import gzip
import os
import struct
import subprocess

import numpy as np
import pytest

from oracle import orc
from tests import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def _ref():
    if not orc.have_ref() and os.path.isdir(orc.REFERENCE_SRC):
        orc.build(ref=True)


@pytest.fixture(scope="module")
def printer(tmp_path_factory):
    lib = os.path.join(ROOT, "fastk_amd", "lib")
    assert os.path.exists(os.path.join(lib, "libfastk_amd.so")), "build fastk_amd/csrc first"
    exe = str(tmp_path_factory.mktemp("itp") / "input_threads_print")
    subprocess.run(["gcc", "-O1", "-Wall", "-Wno-unused-function", "-o", exe, os.path.join(ROOT, "tests", "csrc", "input_threads_print.c"),
                    os.path.join(ROOT, "fastk_amd", "csrc", "host", "input_formats.c"),
                    "-L" + lib, "-lfastk_amd", "-lz", "-lpthread", "-Wl,-rpath," + lib], check=True)
    return exe


def _write(path, rng, fastq, nreads, lens, width):
    """one file; returns the offset of every record's first byte"""
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    starts, out, at = [], [], 0
    for i in range(nreads):
        n = int(rng.choice(lens))
        seq = bytes(acgt[rng.integers(0, 4, size=n)])
        if fastq:
            qual = bytearray(rng.integers(33, 75, size=n, dtype=np.uint8).tobytes())
            if rng.random() < 0.2:
                qual[0] = ord("@")
            elif rng.random() < 0.1:
                qual[0] = ord("+")
            rec = b"@r%d x@y >z +w\n" % i + seq + b"\n+\n" + bytes(qual) + b"\n"
        else:
            rec = b">r%d a>b @c\n" % i
            if width:
                rec += b"".join(seq[j:j + width] + b"\n" for j in range(0, n, width))
            else:
                rec += seq + b"\n"
        starts.append(at)
        out.append(rec)
        at += len(rec)
    data = b"".join(out)
    if path.endswith(".gz"):
        with gzip.open(path, "wb", compresslevel=1) as f:
            f.write(data)
    else:
        with open(path, "wb") as f:
            f.write(data)
    return starts


CASES = [  # T, fastq, gz, (reads per file ...), read lengths, line width
    (4, False, False, (6000,), (100, 150, 250), 0), (4, False, False, (3000,), (100, 1500, 4000), 60),
    (3, True, False, (5000,), (100, 150, 151), 0), (8, True, False, (9000,), (40, 150, 400), 0),
    (2, False, False, (300,), (100, 150), 0),                                   # (too small for two threads)
    (6, False, False, (2500, 400, 3000), (100, 150, 250), 50), (5, True, False, (100, 4000, 100, 2000), (150,), 0),
    (4, False, False, (30, 20, 9000), (150, 300), 0),
    (3, True, True, (500, 800), (150,), 0), (2, False, True, (300, 200, 100, 400, 250), (150, 300), 70),
    (8, False, False, (1200,), (30, 5000, 60000), 100), (7, False, False, (2000, 2000), (33, 34, 35), 0)]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_input_threads_are_the_references(case, printer, tmp_path):
    if not orc.have_ref():
        util.no_reference("oracle/_ref/FastK not built")
    T, fastq, gz, per_file, lens, width = CASES[case]
    rng = np.random.default_rng(9100 + case)
    d = str(tmp_path)
    paths, starts, first = [], [], [0]
    for fi, n in enumerate(per_file):
        p = os.path.join(d, "xyzuvw"[fi] + (".fastq" if fastq else ".fasta") + (".gz" if gz else ""))
        starts.append(_write(p, rng, fastq, n, lens, width))
        first.append(first[-1] + n)
        paths.append(p)
    # the reference: one .pidx part per input thread, (first read, reads) in its header
    subprocess.run([os.path.join(orc.REF_DIR, "FastK"), "-k12", "-t1", "-T%d" % T, "-p", "-P" + d] + paths, check=True, cwd=d,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    kmer, nparts = struct.unpack("<ii", open(os.path.join(d, "x.prof"), "rb").read(8))
    ref = []
    for t in range(1, nparts + 1):
        b, n = struct.unpack("<qq", open(os.path.join(d, ".x.pidx.%d" % t), "rb").read(20)[4:])
        ref.append((b, n))
    # ours: where every thread begins -> the index of the read that begins there
    out = subprocess.run([printer, "starts", str(T), "1" if fastq else "0"] + paths, check=True, capture_output=True, text=True).stdout.split()
    begins = []
    for fi, off in zip(out[0::2], out[1::2]):
        fi, off = int(fi), int(off)
        assert off in starts[fi], ("not a record start", fi, off)
        begins.append(first[fi] + starts[fi].index(off))
    begins.append(first[-1])
    ours = [(begins[t], begins[t + 1] - begins[t]) for t in range(len(begins) - 1)]
    assert ours == ref
    # ... and the host scanner dealing the reads: the blocks it would push, counted by the thread they are pushed for
    dealt = subprocess.run([printer, "deal", str(T), "1" if fastq else "0"] + paths, check=True, capture_output=True, text=True).stdout.split()
    assert [int(x) for x in dealt] == [n for b, n in ref]
