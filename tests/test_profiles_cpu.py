"""Profiles (-p), CPU side: the oracle's per-read counts against the reference's own .prof files
(decoded), the codec round trip, and the library's profile writer read back by the reference's
Profex.  Needs oracle/_ref (built from /root/reference here; prebuilt on the GPU box)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import orc
from tests import util

pytestmark = pytest.mark.skipif(not orc.have_ref() and not os.path.isdir(orc.REFERENCE_SRC),
                                reason="reference build not available")


@pytest.fixture(scope="module", autouse=True)
def _ref():
    if not orc.have_ref() or not os.path.exists(os.path.join(orc.REF_DIR, "Profex")):
        orc.build(ref=True)


def ref_profiles(case, bases, boff, d, T=None):
    path = os.path.join(d, "r.fastq")
    orc.write_fastq(path, bases, boff)
    orc.run_ref_fastk(path, case["k"], case["cutoff"], T or case["T"], d, extra=("-p",))
    return orc.read_profiles(d, "r")


def profex_text(d, root):
    out = subprocess.run([os.path.join(orc.REF_DIR, "Profex"), os.path.join(d, root)] +
                         ["1-#"], check=True, capture_output=True, cwd=d)
    return out.stdout


def write_prof(data, offs, kmer, nparts, d, root):
    import fastk_amd
    from fastk_amd import api
    L = api.load_library()
    pr = api.CProfiles()
    data = np.ascontiguousarray(data, dtype=np.uint8)
    offs = np.ascontiguousarray(offs, dtype=np.int64)
    pr.nreads = len(offs) - 1
    pr.nbytes = len(data)
    pr.data = data.ctypes.data_as(C.POINTER(C.c_uint8))
    pr.offsets = offs.ctypes.data_as(C.POINTER(C.c_int64))
    assert L.fk_write_prof(C.byref(pr), kmer, nparts, d.encode(), root.encode()) == 0


@pytest.mark.parametrize("name", ["synth_tiny_k40_t1_T2", "edge_k40_t1_T4", "edge_k21_t2_T3", "edge_k51_t1_T4"])
def test_oracle_profiles_match_reference(name, tmp_path):
    case, bases, boff = util.load_case(name)
    k = case["k"]
    d = str(tmp_path)
    kk, enc = ref_profiles(case, bases, boff, d)
    assert kk == k and len(enc) == len(boff) - 1
    table = orc.fastk(k, bases, boff, cutoff=1).table
    exp = orc.profile_counts(k, bases, boff, table)
    ncanon = 0
    for i, (e, x) in enumerate(zip(enc, exp)):
        assert orc.profile_decode(e) == x.tolist(), "read %d" % i
        mine = orc.profile_encode(x)
        assert orc.profile_decode(mine) == x.tolist()
        assert len(mine) <= len(e)                      # canonical is never longer
        ncanon += (mine == e)
    assert ncanon > 0.5 * len(enc)                       # most reads are byte-identical anyway
    # the C decoder that digests the profiles of the cases above fixture size (orc.profiles_digest_files, batches of 777
    # reads) gives the digest the Python decoder gives, which is the golden one
    nr, nb, npos, dig = orc.profiles_digest_files(d, "r", batch=777)
    assert (nr, nb, npos) == (len(enc), sum(len(e) for e in enc), sum(len(x) for x in exp))
    assert dig == orc.profiles_digest(exp) == case["expected"]["prof"]["decoded_sha256"]

    # the library's writer, read by the reference's Profex: same listing as for the reference's files
    blobs = [orc.profile_encode(x) for x in exp]
    offs = np.concatenate([[0], np.cumsum([len(b) for b in blobs])]).astype(np.int64)
    data = np.frombuffer(b"".join(blobs), dtype=np.uint8)
    od = os.path.join(d, "o")
    os.mkdir(od)
    write_prof(data, offs, k, 3, od, "r")
    k2, back = orc.read_profiles(od, "r")
    assert k2 == k and back == blobs
    assert profex_text(od, "r") == profex_text(d, "r")


def test_codec_extremes():
    rng = np.random.default_rng(5)
    for _ in range(200):
        n = int(rng.integers(1, 400))
        mode = rng.integers(0, 4)
        if mode == 0:
            c = rng.integers(0, 32768, n)
        elif mode == 1:
            c = np.repeat(rng.integers(0, 32768, (n + 69) // 70), 70)[:n]
        elif mode == 2:
            c = np.clip(np.cumsum(rng.integers(-40, 41, n)) + 100, 0, 32767)
        else:
            c = rng.choice([0, 1, 127, 128, 32767], n)
        c = c.astype(np.uint16)
        assert orc.profile_decode(orc.profile_encode(c)) == c.tolist()
    assert orc.profile_encode([]) == b"" and orc.profile_decode(b"") == []
    assert orc.profile_encode([5] * 64) == bytes([5, 63])
    assert orc.profile_encode([5] * 65) == bytes([5, 63, 1])
    assert orc.profile_encode([200, 169, 168]) == bytes([0x80, 200, 0x40 | (-31 & 0x3f), 0x7f])
    assert orc.profile_encode([32767, 0]) == bytes([0xff, 0xff, 0x80, 0x01])
