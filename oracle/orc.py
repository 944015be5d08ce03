"""ctypes binding of the CPU restatement (oracle/fk_oracle.c) and helpers around the
reference build in oracle/_ref.

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Product code under fastk_amd/ must never import this module.
"""
import ctypes as C
import hashlib
import os
import struct
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_build", "libfkoracle.so")
REF_DIR = os.path.join(HERE, "_ref")
REFERENCE_SRC = "/root/reference"


def build(ref=True):
    """Compile the restatement and, when the reference sources are present, oracle/_ref."""
    subprocess.check_call(["make", "-s", "-C", HERE, "all"])
    if ref and os.path.isdir(REFERENCE_SRC):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])
        if os.path.exists(os.path.join(os.path.dirname(HERE), "fastk_amd", "lib", "libfastk_amd.so")):
            subprocess.check_call(["make", "-s", "-C", HERE, "ref_gpu"])


class OrcParams(C.Structure):
    _fields_ = [(n, C.c_int) for n in
                ("kmer", "min_len", "max_super", "smer", "slen_bits", "slen_bytes",
                 "kmer_bytes", "smer_bytes", "smer_word", "kmer_word")] + [("tran", C.c_int * 4)]


class OrcResult(C.Structure):
    _fields_ = [("hist", C.c_int64 * 0x8000), ("max_inst", C.c_int64), ("ninst", C.c_int64),
                ("nsuper", C.c_int64), ("ndistinct_super", C.c_int64),
                ("nweighted", C.c_int64), ("ndistinct", C.c_int64), ("ntable", C.c_int64),
                ("table", C.POINTER(C.c_uint8)), ("wfirst", C.c_int64 * 256)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build(ref=False)
        L = C.CDLL(LIB_PATH)
        L.orc_params_init.argtypes = [C.POINTER(OrcParams), C.c_int, C.c_int]
        L.orc_train_tran.argtypes = [C.POINTER(OrcParams), C.c_void_p, C.c_void_p, C.c_int64, C.c_int]
        L.orc_set_profile_mode.argtypes = [C.c_int]
        L.orc_set_profile_mode.restype = None
        L.orc_distribute_block.restype = C.c_int64
        L.orc_distribute_block.argtypes = [C.POINTER(OrcParams), C.c_void_p, C.c_void_p, C.c_int64,
                                           C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int64),
                                           C.POINTER(C.c_int64)]
        L.orc_msd_sort.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int]
        L.orc_lsd_sort.restype = C.c_void_p
        L.orc_lsd_sort.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.orc_kmer_list.restype = C.c_int64
        L.orc_kmer_list.argtypes = [C.POINTER(OrcParams), C.c_void_p, C.c_int64,
                                    C.POINTER(C.c_void_p), C.POINTER(C.c_int64),
                                    C.POINTER(C.c_int64)]
        L.orc_count_sorted.argtypes = [C.POINTER(OrcParams), C.c_void_p, C.c_int64, C.c_int,
                                       C.POINTER(OrcResult)]
        L.orc_fastk.argtypes = [C.POINTER(OrcParams), C.c_void_p, C.c_void_p, C.c_int64, C.c_int,
                                C.c_int, C.POINTER(OrcResult)]
        L.orc_brute.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int,
                                C.POINTER(OrcResult)]
        L.orc_scheme_train.argtypes = [C.POINTER(OrcParams), C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int,
                                       C.POINTER(OrcScheme)]
        L.orc_fastk_parts.argtypes = [C.POINTER(OrcParams), C.POINTER(OrcScheme), C.c_void_p, C.c_void_p, C.c_int64,
                                      C.c_int, C.c_int, C.POINTER(OrcResult)]
        L.orc_result_free.argtypes = [C.POINTER(OrcResult)]
        L.orc_hist_bytes.restype = C.c_int64
        L.orc_hist_bytes.argtypes = [C.c_int, C.POINTER(OrcResult), C.c_void_p]
        L.orc_table_split.argtypes = [C.POINTER(OrcParams), C.POINTER(OrcResult), C.c_int,
                                      C.POINTER(C.c_int)]
        L.orc_idx_bytes.restype = C.c_int
        L.orc_idx_bytes.argtypes = [C.c_int, C.c_int64]
        L.orc_write_outputs.argtypes = [C.POINTER(OrcParams), C.POINTER(OrcResult), C.c_int,
                                        C.c_int, C.POINTER(C.c_int), C.c_char_p, C.c_char_p]
        L.orc_synth_block.restype = C.c_void_p
        L.orc_synth_block.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64,
                                      C.c_int64, C.c_void_p]
        L.orc_load_fastx.restype = C.c_void_p
        L.orc_load_fastx.argtypes = [C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
        L.orc_profile_decode_stream.restype = C.c_int64
        L.orc_profile_decode_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]
        L.free = C.CDLL(None).free
        L.free.argtypes = [C.c_void_p]
        _lib = L
    return _lib


class OrcScheme(C.Structure):
    _fields_ = [("pad", C.c_int), ("states", C.c_int), ("nparts", C.c_int), ("part", C.POINTER(C.c_int))]


def params(kmer, pad=0):
    P = OrcParams()
    lib().orc_params_init(C.byref(P), kmer, pad)
    return P


# --------------------------------------------------------------------------- inputs

def block_from_reads(reads):
    """DATA_BLOCK-style (FastK.h:87-98) buffer from a list of str/bytes reads."""
    bs = [r.encode() if isinstance(r, str) else bytes(r) for r in reads]
    boff = np.zeros(len(bs) + 1, dtype=np.int64)
    np.cumsum([len(b) + 1 for b in bs], out=boff[1:])
    bases = np.frombuffer(b"".join(b + b"\0" for b in bs), dtype=np.uint8).copy()
    if bases.size == 0:
        bases = np.zeros(1, dtype=np.uint8)
    return bases, boff


def synth_block(seed, genome_len, read_len, err_ppm, first_read, nreads):
    L = lib()
    boff = np.zeros(nreads + 1, dtype=np.int64)
    p = L.orc_synth_block(seed, genome_len, read_len, err_ppm, first_read, nreads,
                          boff.ctypes.data)
    n = int(boff[nreads])
    bases = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n,)).copy()
    L.free(p)
    return bases, boff


def load_fastx(path):
    L = lib()
    bp = C.c_void_p()
    nr = C.c_int64()
    p = L.orc_load_fastx(path.encode(), C.byref(bp), C.byref(nr))
    if not p:
        raise IOError(path)
    boff = np.ctypeslib.as_array(C.cast(bp, C.POINTER(C.c_int64)), shape=(nr.value + 1,)).copy()
    n = int(boff[-1])
    bases = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(max(n, 1),)).copy()
    L.free(p)
    L.free(bp)
    return bases, boff


def write_fasta(path, bases, boff, width=0):
    with open(path, "wb") as f:
        for i in range(len(boff) - 1):
            s = bases[boff[i]:boff[i + 1] - 1].tobytes()
            f.write(b">r%d\n" % i)
            if width:
                for o in range(0, len(s), width):
                    f.write(s[o:o + width] + b"\n")
            else:
                f.write(s + b"\n")


def write_fastq(path, bases, boff):
    with open(path, "wb") as f:
        for i in range(len(boff) - 1):
            s = bases[boff[i]:boff[i + 1] - 1].tobytes()
            f.write(b"@r%d\n" % i + s + b"\n+\n" + b"I" * len(s) + b"\n")


# --------------------------------------------------------------------------- whole path

class Result:
    """hist / table of one run, in numpy form."""

    def __init__(self, kmer, R, kmer_word):
        self.kmer = kmer
        self.hist = np.ctypeslib.as_array(R.hist).copy()
        self.max_inst = int(R.max_inst)
        self.ninst = int(R.ninst)
        self.nsuper = int(R.nsuper)
        self.ndistinct_super = int(R.ndistinct_super)
        self.nweighted = int(R.nweighted)
        self.ndistinct = int(R.ndistinct)
        self.ntable = int(R.ntable)
        self.wfirst = np.ctypeslib.as_array(R.wfirst).copy()
        if R.ntable > 0:
            self.table = np.ctypeslib.as_array(R.table, shape=(R.ntable, kmer_word)).copy()
        else:
            self.table = np.zeros((0, kmer_word), dtype=np.uint8)

    def hist_bytes(self):
        return hist_file_bytes(self.kmer, self.hist, self.max_inst)


def hist_file_bytes(kmer, hist, max_inst):
    """.hist encoding (count.c:1893-1910): int k, int 1, int 0x7fff, i64 hist[1], i64 max_inst,
    i64 hist[1..0x7fff]."""
    h = np.asarray(hist, dtype=np.int64)
    return (struct.pack("<iii", kmer, 1, 0x7fff) + struct.pack("<qq", int(h[1]), int(max_inst))
            + h[1:0x8000].tobytes())


def first_block_reads(boff):
    """reads Get_First_Block(io, 1e9) hands to Determine_Scheme (io.c:2606-2630, END_SEQ io.c:547-556): up to 1e9/150
    reads, cut once the block is within DT_MINIM = 100,000 bytes of 1e9 + that many terminators"""
    nreads = len(boff) - 1
    maxrds = 1000000000 // 150
    omax = 1000000000 + maxrds
    ends = np.asarray(boff[1:], dtype=np.int64) - int(boff[0])
    over = np.nonzero(ends > omax - 100000)[0]
    train = nreads if len(over) == 0 else int(over[0]) + 1
    return min(train, maxrds, nreads)


def scheme(kmer, bases, boff, nparts, nthreads=4, bc_prefix=0):
    """Determine_Scheme restated: (pad, nparts, Min_Part as a list) for the training block of these reads"""
    L = lib()
    P = params(kmer)
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    boff = np.ascontiguousarray(boff, dtype=np.int64)
    train = first_block_reads(boff)
    L.orc_train_tran(C.byref(P), bases.ctypes.data, boff.ctypes.data, train, nthreads)
    S = OrcScheme()
    L.orc_scheme_train(C.byref(P), bases.ctypes.data, boff.ctypes.data, train, bc_prefix, nparts, C.byref(S))
    part = [S.part[i] for i in range(S.states)]
    return P, S, part


def fastk_parts(kmer, bases, boff, sort_memory, cutoff=1, bc_prefix=0, nthreads=4):
    """The whole path with the reference's buckets for a sort memory of `sort_memory` bytes (FastK.c:417-429):
    same histogram and table as fastk(); wfirst -- hence the part boundaries -- from bucket 0 alone."""
    L = lib()
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    boff = np.ascontiguousarray(boff, dtype=np.int64)
    nreads = len(boff) - 1
    train = first_block_reads(boff)
    totlen = int(boff[train] - boff[0]) - train
    ratio = 1.0 if train >= nreads else float(int(boff[nreads] - boff[0])) / float(totlen + train)
    kw = ((2 * kmer + 7) >> 3) + 2
    gsize = int(float(totlen - kmer * train) * ratio * kw)
    nparts = (gsize - 1) // sort_memory + 1
    if nparts <= 1:
        return fastk(kmer, bases, boff, cutoff=cutoff, bc_prefix=bc_prefix, nthreads=nthreads)
    P, S, _ = scheme(kmer, bases, boff, nparts, nthreads, bc_prefix)
    R = OrcResult()
    L.orc_fastk_parts(C.byref(P), C.byref(S), bases.ctypes.data, boff.ctypes.data, nreads, bc_prefix, cutoff,
                      C.byref(R))
    out = Result(kmer, R, P.kmer_word)
    out.params = P
    out.nparts = S.nparts
    L.orc_result_free(C.byref(R))
    return out


def fastk(kmer, bases, boff, cutoff=1, bc_prefix=0, train=True, pad=0, nthreads=4, profile=False):
    """profile: restate a run with -p -- the super-mers keep the read's strand (split.c:1245), which changes wfirst (the
    census that cuts the hidden .ktab parts) and nothing else of the result."""
    L = lib()
    P = params(kmer, pad)
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    boff = np.ascontiguousarray(boff, dtype=np.int64)
    if train:
        L.orc_train_tran(C.byref(P), bases.ctypes.data, boff.ctypes.data, len(boff) - 1, nthreads)
    R = OrcResult()
    L.orc_set_profile_mode(1 if profile else 0)
    try:
        L.orc_fastk(C.byref(P), bases.ctypes.data, boff.ctypes.data, len(boff) - 1, bc_prefix, cutoff,
                    C.byref(R))
    finally:
        L.orc_set_profile_mode(0)
    out = Result(kmer, R, P.kmer_word)
    out.params = P
    L.orc_result_free(C.byref(R))
    return out


def brute(kmer, bases, boff, cutoff=1, bc_prefix=0):
    L = lib()
    P = params(kmer)
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    boff = np.ascontiguousarray(boff, dtype=np.int64)
    R = OrcResult()
    L.orc_brute(kmer, bases.ctypes.data, boff.ctypes.data, len(boff) - 1, bc_prefix, cutoff,
                C.byref(R))
    out = Result(kmer, R, P.kmer_word)
    out.params = P
    L.orc_result_free(C.byref(R))
    return out


def table_split(P, res, nthreads):
    R = OrcResult()
    R.nweighted = res.nweighted
    for i in range(256):
        R.wfirst[i] = int(res.wfirst[i])
    sp = (C.c_int * nthreads)()
    lib().orc_table_split(C.byref(P), C.byref(R), nthreads, sp)
    return list(sp)


def idx_bytes(kmer, ntable):
    return lib().orc_idx_bytes(kmer, ntable)


def write_outputs(res, cutoff, nthreads, outdir, root, split=None):
    """Write .hist / .ktab files from a Result with the oracle's writers."""
    L = lib()
    P = res.params
    R = OrcResult()
    for i in range(0x8000):
        R.hist[i] = int(res.hist[i])
    R.max_inst = res.max_inst
    R.nweighted = res.nweighted
    R.ntable = res.ntable
    for i in range(256):
        R.wfirst[i] = int(res.wfirst[i])
    tab = np.ascontiguousarray(res.table)
    R.table = tab.ctypes.data_as(C.POINTER(C.c_uint8))
    sp = None
    if split is not None:
        sp = (C.c_int * nthreads)(*split)
    rc = L.orc_write_outputs(C.byref(P), C.byref(R), cutoff, nthreads, sp, outdir.encode(),
                             root.encode())
    if rc != 0:
        raise IOError("orc_write_outputs failed")


# --------------------------------------------------------------------------- stage helpers

def distribute(P, bases, boff, bc_prefix=0):
    L = lib()
    out = C.c_void_p()
    n = C.c_int64(0)
    cap = C.c_int64(0)
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    boff = np.ascontiguousarray(boff, dtype=np.int64)
    inst = L.orc_distribute_block(C.byref(P), bases.ctypes.data, boff.ctypes.data, len(boff) - 1,
                                  bc_prefix, C.byref(out), C.byref(n), C.byref(cap))
    if n.value:
        recs = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint8)),
                                     shape=(n.value, P.smer_word)).copy()
    else:
        recs = np.zeros((0, P.smer_word), dtype=np.uint8)
    if out.value:
        L.free(out)
    return recs, int(inst)


def msd_sort(recs, ksize):
    a = np.ascontiguousarray(recs).copy()
    if a.shape[0]:
        lib().orc_msd_sort(a.ctypes.data, a.shape[0], a.shape[1], ksize)
    return a


def lsd_sort(recs, byte_list):
    a = np.ascontiguousarray(recs).copy()
    t = np.empty_like(a)
    bl = (C.c_int * (len(byte_list) + 1))(*(list(byte_list) + [-1]))
    if a.shape[0] == 0:
        return a
    p = lib().orc_lsd_sort(a.shape[0], a.ctypes.data, t.ctypes.data, a.shape[1], bl)
    return a if p == a.ctypes.data else t


def kmer_list(P, sorted_smers):
    L = lib()
    out = C.c_void_p()
    ovf = C.c_int64()
    nd = C.c_int64()
    s = np.ascontiguousarray(sorted_smers)
    w = L.orc_kmer_list(C.byref(P), s.ctypes.data, s.shape[0], C.byref(out), C.byref(ovf),
                        C.byref(nd))
    if w:
        recs = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint8)),
                                     shape=(w, P.kmer_word)).copy()
    else:
        recs = np.zeros((0, P.kmer_word), dtype=np.uint8)
    L.free(out)
    return recs, int(ovf.value), int(nd.value)


def count_sorted(P, sorted_kmers, cutoff):
    L = lib()
    R = OrcResult()
    s = np.ascontiguousarray(sorted_kmers)
    L.orc_count_sorted(C.byref(P), s.ctypes.data, s.shape[0], cutoff, C.byref(R))
    out = Result(P.kmer, R, P.kmer_word)
    out.params = P
    L.orc_result_free(C.byref(R))
    return out


# --------------------------------------------------------------------------- .ktab / .hist readers

def read_hist(path):
    b = open(path, "rb").read()
    k, lo, hi = struct.unpack_from("<iii", b, 0)
    ilow, ihigh = struct.unpack_from("<qq", b, 12)
    h = np.frombuffer(b, dtype=np.int64, offset=28, count=hi - lo + 1)
    return dict(kmer=k, low=lo, high=hi, ilow=ilow, ihigh=ihigh, hist=h, raw=b)


def read_ktab(path_root):
    """Read <dir>/<root>.ktab + hidden parts.  Returns dict with header fields, per-part record
    arrays, the full (kmer bytes + count) table and the canonical stream of SURVEY.md section 4
    (stub index from byte 16 + part payloads from byte 12)."""
    d, root = os.path.split(path_root)
    d = d or "."
    stub = open(os.path.join(d, root + ".ktab"), "rb").read()
    k, nparts, minval, ib = struct.unpack_from("<iiii", stub, 0)
    idx = np.frombuffer(stub, dtype=np.int64, offset=16, count=1 << (8 * ib))
    kb = (2 * k + 7) >> 3
    pw = kb + 2 - ib
    parts = []
    payload = []
    for t in range(1, nparts + 1):
        pb = open(os.path.join(d, ".%s.ktab.%d" % (root, t)), "rb").read()
        pk, n = struct.unpack_from("<iq", pb, 0)
        assert pk == k and len(pb) == 12 + n * pw, (pk, k, len(pb), n, pw)
        parts.append(np.frombuffer(pb, dtype=np.uint8, offset=12).reshape(n, pw))
        payload.append(pb[12:])
    suffix = np.concatenate(parts) if parts else np.zeros((0, pw), dtype=np.uint8)
    nels = suffix.shape[0]
    # rebuild prefixes from the cumulative index
    counts = np.diff(np.concatenate([[0], idx]))
    pre = np.repeat(np.arange(len(idx), dtype=np.int64), counts)
    full = np.zeros((nels, kb + 2), dtype=np.uint8)
    for b in range(ib):
        full[:, b] = (pre >> (8 * (ib - 1 - b))) & 0xff
    full[:, ib:] = suffix
    stream = stub[16:] + b"".join(payload)
    return dict(kmer=k, nparts=nparts, minval=minval, ibytes=ib, index=idx, parts=parts,
                table=full, nels=nels, stream_sha256=hashlib.sha256(stream).hexdigest(),
                part_sizes=[p.shape[0] for p in parts])


def table_stream_sha256(kmer, table, ib=None):
    """Canonical stream digest computed from a full table (n, KMER_BYTES+2) array."""
    n = table.shape[0]
    if ib is None:
        ib = idx_bytes(kmer, n)
    pre = np.zeros(n, dtype=np.int64)
    for b in range(ib):
        pre = (pre << 8) | table[:, b].astype(np.int64)
    idx = np.cumsum(np.bincount(pre, minlength=1 << (8 * ib))).astype(np.int64)
    stream = idx.tobytes() + np.ascontiguousarray(table[:, ib:]).tobytes()
    return hashlib.sha256(stream).hexdigest()


# --------------------------------------------------------------------------- reference binary

_fkref = None


def ref_lsd_sort(recs, byte_list, nthreads=4):
    """The REFERENCE's own LSD_Sort (LSDsort.c:115, compiled where it lies into oracle/_ref/libfkref.so)
    on an (n, rsize) uint8 array; byte_list least significant first.  Returns the sorted copy."""
    global _fkref
    if _fkref is None:
        _fkref = C.CDLL(os.path.join(REF_DIR, "libfkref.so"), mode=os.RTLD_LAZY)   # CRAM symbols stay unresolved
        _fkref.LSD_Sort.restype = C.c_void_p
        _fkref.LSD_Sort.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    C.c_int.in_dll(_fkref, "NTHREADS").value = nthreads
    n, rsize = recs.shape
    src = np.ascontiguousarray(recs).copy()
    trg = np.empty_like(src)
    bl = (C.c_int * (len(byte_list) + 1))(*(list(byte_list) + [-1]))
    if n == 0:
        return src
    out = _fkref.LSD_Sort(n, src.ctypes.data, trg.ctypes.data, rsize, bl)
    return (src if out == src.ctypes.data else trg).copy()


class _RefRange(C.Structure):                  # FastK.h:135-143
    _fields_ = [("beg", C.c_int), ("end", C.c_int), ("off", C.c_int64), ("khist", C.c_int64 * 256),
                ("count", C.c_int64 * 0x8000), ("max_inst", C.c_int64), ("byte1", C.c_int)]


def ref_weighted_kmer_sort(recs, kmer, nthreads=4):
    """The REFERENCE's own Weighted_Kmer_Sort (MSDsort.c:536-544 -> msd_sort :308-390 -> radix_sort / shell_sort, with
    hist_kmers :491-509 called on every run of equal k-mers) on an (n, KMER_BYTES + 2) uint8 array of weighted k-mers.
    The engine wants its input dealt on the first key byte already (count.c:1520-1539 builds the list that way), so the
    records are first brought into first-byte order (stable).  Returns (sorted array, histogram int64[0x8000],
    max_inst): byte 0 of a run's first record comes back as the flag 1 and that record's count as the run's sum
    (capped at 0x7fff), exactly as the engine leaves them."""
    global _fkref
    if _fkref is None:
        ref_lsd_sort(np.zeros((0, 4), dtype=np.uint8), [0])          # loads the library
    L = _fkref
    n, rsize = recs.shape
    kb = (kmer + 3) // 4
    assert rsize == kb + 2
    L.Weighted_Kmer_Sort.restype = None
    L.Weighted_Kmer_Sort.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_int64), C.c_int,
                                     C.POINTER(_RefRange)]
    for name, val in (("KMER", kmer), ("KMER_BYTES", kb), ("KMER_WORD", rsize), ("DO_PROFILE", 0), ("NTHREADS", nthreads)):
        C.c_int.in_dll(L, name).value = val
    order = np.argsort(recs[:, 0], kind="stable")
    arr = np.zeros((n + 4, rsize), dtype=np.uint8)                  # (slack behind the list)
    arr[:n] = recs[order]
    part = (C.c_int64 * 256)(*[int(c) * rsize for c in np.bincount(recs[:, 0], minlength=256)])
    panels = (_RefRange * nthreads)()
    if n > 0:
        L.Weighted_Kmer_Sort(arr.ctypes.data, n, rsize, kb, part, nthreads, panels)
    hist = np.zeros(0x8000, dtype=np.int64)
    max_inst = 0
    for t in range(nthreads):
        hist += np.ctypeslib.as_array(panels[t].count)
        max_inst += int(panels[t].max_inst)
    return arr[:n].copy(), hist, max_inst


def ref_supermer_sort(recs, kmer, nthreads=4):
    """The REFERENCE's own Supermer_Sort (MSDsort.c:458-489 -> msd_sort :306-376 -> radix_sort / shell_sort, count_smers
    :381-456 on every run of equal super-mers) on an (n, SMER_WORD) uint8 array of super-mer records
    [SMER_BYTES 2-bit bases][SLEN_BYTES n-1] -- called as count.c:1458 calls it: key = the whole record, the list dealt
    on its first byte already (count.c:226-251 builds it that way; stable here).  Returns the sorted array; byte 0 of a
    run's first record comes back as the engine leaves it (its first-byte value), so callers compare bytes 1.. and
    restore byte 0 from the first-byte order."""
    global _fkref
    if _fkref is None:
        ref_lsd_sort(np.zeros((0, 4), dtype=np.uint8), [0])          # loads the library
    L = _fkref
    n, rsize = recs.shape
    min_len = 5
    max_super = kmer - min_len + 1                                   # FastK.c:456-468 with PAD_LEN = 5
    smer = max_super + kmer - 1
    smer_bytes = (smer + 3) // 4
    slen_bits = max(1, int(max_super - 1).bit_length())
    slen_bytes = (slen_bits + 7) // 8
    assert rsize == smer_bytes + slen_bytes, (rsize, smer_bytes, slen_bytes)
    L.Supermer_Sort.restype = None
    L.Supermer_Sort.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_int64), C.c_int,
                                C.POINTER(_RefRange)]
    for name, val in (("KMER", kmer), ("KMER_BYTES", (kmer + 3) // 4), ("SMER_BYTES", smer_bytes), ("SLEN_BYTES", slen_bytes),
                      ("SMER_WORD", rsize), ("MAX_SUPER", max_super), ("DO_PROFILE", 0), ("NTHREADS", nthreads)):
        C.c_int.in_dll(L, name).value = val
    order = np.argsort(recs[:, 0], kind="stable")
    buf = np.zeros((n + 8, rsize), dtype=np.uint8)                  # array[-1] = 0 and slack behind (count.c:1333, MSDsort.c:375)
    arr = buf[4:4 + n]
    arr[:] = recs[order]
    part = (C.c_int64 * 256)(*[int(c) * rsize for c in np.bincount(recs[:, 0], minlength=256)])
    panels = (_RefRange * nthreads)()
    if n > 0:
        L.Supermer_Sort(arr.ctypes.data, n, rsize, rsize, part, nthreads, panels)
    return arr.copy()


def have_fkref():
    return os.path.exists(os.path.join(REF_DIR, "libfkref.so"))


def have_ref():
    return os.path.exists(os.path.join(REF_DIR, "FastK"))


def run_ref_fastk(fastx_path, kmer, cutoff, nthreads, workdir, extra=()):
    """Run the reference FastK (oracle/_ref/FastK) on a file; outputs land next to the input."""
    cmd = [os.path.join(REF_DIR, "FastK"), "-k%d" % kmer, "-T%d" % nthreads, "-P" + workdir]
    if cutoff > 0:
        cmd.append("-t%d" % cutoff)
    cmd += list(extra) + [fastx_path]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                   cwd=workdir)


# --------------------------------------------------------------------------- profiles (-p)
# Restatement of what a profile IS (README.md:1010-1069): per read, the count of every k-mer in
# read order, 0 where the k-mer holds a non-acgt base, counts capped at 32767 -- computed here from
# the oracle's cutoff-1 table.  The reference assembles the same numbers from per-super-mer
# fragments (count.c:868-947) that merge.c stitches into read order; its byte stream flushes the
# pending zero run at every panel of 1024*NPARTS fragments (merge.c:65,711-716) and writes a junction
# d = -31 in two bytes (merge.c:456,590), so parity with the
# reference is on the DECODED counts (libfastk.c:1657 Fetch_Profile is the decoder restated below).

_CODE = np.full(256, 4, dtype=np.uint8)
for _i, _c in enumerate("acgt"):
    _CODE[ord(_c)] = _i
    _CODE[ord(_c.upper())] = _i


def profile_counts(kmer, bases, boff, table):
    """table: (n, KMER_BYTES+2) uint8 sorted, cutoff 1.  Returns a list of uint16 arrays, one per read."""
    kb = (kmer + 3) // 4
    keys = np.ascontiguousarray(table[:, :kb]).view("S%d" % kb).reshape(-1) if len(table) else \
        np.zeros(0, dtype="S%d" % kb)
    cnts = (table[:, kb].astype(np.uint16) | (table[:, kb + 1].astype(np.uint16) << 8)) if len(table) \
        else np.zeros(0, dtype=np.uint16)
    wts = np.array([64, 16, 4, 1], dtype=np.uint8)
    out = []
    for r in range(len(boff) - 1):
        s = bases[boff[r]:boff[r + 1] - 1]
        n = len(s) - kmer + 1
        if n <= 0:
            out.append(np.zeros(0, dtype=np.uint16))
            continue
        code = _CODE[s]
        win = np.lib.stride_tricks.sliding_window_view(code, kmer)            # (n, k)
        ok = (win < 4).all(axis=1)
        f = np.zeros((n, kb * 4), dtype=np.uint8)
        f[:, :kmer] = win & 3
        g = np.zeros((n, kb * 4), dtype=np.uint8)
        g[:, :kmer] = 3 - (win[:, ::-1] & 3)
        fb = (f.reshape(n, kb, 4) * wts).sum(axis=2).astype(np.uint8)
        gb = (g.reshape(n, kb, 4) * wts).sum(axis=2).astype(np.uint8)
        diff = fb != gb
        first = diff.argmax(axis=1)
        rows = np.arange(n)
        use_rc = diff.any(axis=1) & (gb[rows, first] < fb[rows, first])
        canon = np.where(use_rc[:, None], gb, fb)
        q = np.ascontiguousarray(canon).view("S%d" % kb).reshape(-1)
        pos = np.searchsorted(keys, q)
        pos = np.minimum(pos, max(len(keys) - 1, 0))
        hit = (keys[pos] == q) if len(keys) else np.zeros(n, dtype=bool)
        c = np.where(hit & ok, cnts[pos] if len(keys) else 0, 0).astype(np.uint16)
        out.append(c)
    return out


def profile_encode(counts):
    """Canonical stream of README.md:1029-1069: one-byte forms whenever possible, runs up to 63."""
    if len(counts) == 0:
        return b""
    o = bytearray()
    p = int(counts[0])
    if p < 128:
        o.append(p)
    else:
        o += bytes([0x80 | (p >> 8), p & 0xff])
    run = 0
    for c in counts[1:]:
        c = int(c)
        if c == p:
            run += 1
            if run == 63:
                o.append(63)
                run = 0
            continue
        if run:
            o.append(run)
            run = 0
        d = c - p
        if -32 < d < 32:
            o.append(0x40 | (d & 0x3f))
        else:
            d &= 0x7fff
            o += bytes([0x80 | (d >> 8), d & 0xff])
        p = c
    if run:
        o.append(run)
    return bytes(o)


def profile_decode(b):
    """Fetch_Profile, libfastk.c:1657-1780."""
    if len(b) == 0:
        return []
    if b[0] & 0x80:
        c = ((b[0] & 0x7f) << 8) | b[1]
        i = 2
    else:
        c = b[0]
        i = 1
    res = [c]
    while i < len(b):
        x = b[i]
        if x & 0x80:
            d = ((x & 0x7f) << 8) | b[i + 1]
            i += 2
            c = (c + d) & 0x7fff
            res.append(c)
        elif x & 0x40:
            d = x & 0x3f
            if d & 0x20:
                d -= 0x40
            c = (c + d) & 0xffff
            res.append(c)
            i += 1
        else:
            res.extend([c] * x)
            i += 1
    return res


def read_profiles(outdir, root):
    """<root>.prof stub + hidden parts -> (kmer, list of per-read encoded byte strings)."""
    kmer, nparts = struct.unpack("<ii", open(os.path.join(outdir, root + ".prof"), "rb").read(8))
    out = []
    for t in range(1, nparts + 1):
        px = open(os.path.join(outdir, ".%s.pidx.%d" % (root, t)), "rb").read()
        k2, = struct.unpack("<i", px[:4])
        b, n = struct.unpack("<qq", px[4:20])
        assert k2 == kmer and b == len(out)
        offs = np.frombuffer(px[20:20 + 8 * n], dtype=np.int64)
        data = open(os.path.join(outdir, ".%s.prof.%d" % (root, t)), "rb").read()
        assert n == 0 or offs[-1] == len(data)
        prev = 0
        for o in offs:
            out.append(data[prev:int(o)])
            prev = int(o)
    return kmer, out


def profiles_digest(count_lists):
    """sha256 over, per read, int32 length + uint16 counts (little-endian): the golden fixtures pin the
    reference's decoded profiles with it."""
    h = hashlib.sha256()
    for c in count_lists:
        a = np.asarray(c, dtype="<u2")
        h.update(struct.pack("<i", len(a)))
        h.update(a.tobytes())
    return h.hexdigest()


def profiles_digest_files(outdir, root, batch=1 << 20):
    """profiles_digest of a whole <root>.prof (stub + hidden parts) without a Python loop per count: the parts
    are decoded by the C restatement of Fetch_Profile, a batch of reads at a time.  Returns (reads, encoded
    bytes, k-mer positions, sha256) -- for data sets whose decoded profiles do not fit a Python list."""
    L = lib()
    kmer, nparts = struct.unpack("<ii", open(os.path.join(outdir, root + ".prof"), "rb").read(8))
    h = hashlib.sha256()
    nreads = nbytes = npos = 0
    for t in range(1, nparts + 1):
        px = open(os.path.join(outdir, ".%s.pidx.%d" % (root, t)), "rb").read()
        k2, = struct.unpack("<i", px[:4])
        b, n = struct.unpack("<qq", px[4:20])
        assert k2 == kmer and b == nreads, (k2, kmer, b, nreads)
        offs = np.frombuffer(px[20:20 + 8 * n], dtype=np.int64)
        data = np.fromfile(os.path.join(outdir, ".%s.prof.%d" % (root, t)), dtype=np.uint8)
        assert n == 0 or offs[-1] == len(data)
        for r0 in range(0, n, batch):
            ends = offs[r0:r0 + batch]
            lo = int(offs[r0 - 1]) if r0 else 0
            rel = np.ascontiguousarray(ends - lo)
            piece = np.ascontiguousarray(data[lo:int(ends[-1])])
            need = L.orc_profile_decode_stream(piece.ctypes.data, rel.ctypes.data, len(rel), None, 0)
            assert need >= 0, "a profile of part %d ends inside a two-byte code" % t
            out = np.empty(need, dtype=np.uint8)
            got = L.orc_profile_decode_stream(piece.ctypes.data, rel.ctypes.data, len(rel), out.ctypes.data, need)
            assert got == need
            h.update(out)
            npos += (need - 4 * len(rel)) // 2
        nreads += n
        nbytes += len(data)
    return nreads, nbytes, npos, h.hexdigest()


# --------------------------------------------------------------------------- SAM / BAM test inputs

def write_sam(path, reads, flags=None):
    """reads: list of str; flags[i] (default 4 = unmapped).  A minimal valid SAM with one aux tag per line
    (the reference insists on one, io.c:1481)."""
    with open(path, "wb") as f:
        f.write(b"@HD\tVN:1.6\tSO:unknown\n@RG\tID:x\n")
        for i, r in enumerate(reads):
            fl = 4 if flags is None else flags[i]
            q = "I" * len(r)
            f.write(("r%d\t%d\t*\t0\t0\t*\t*\t0\t0\t%s\t%s\tRG:Z:x\n" % (i, fl, r, q)).encode())


def _bgzf_block(data):
    import zlib
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    body = c.compress(data) + c.flush()
    bsize = 12 + 6 + len(body) + 8 - 1
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize) + body +
            struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data)))


def write_bam(path, reads, flags=None):
    """Unaligned BAM in BGZF blocks (SAM spec 4.2) with the reads as records; letters outside
    =ACMGRSVTWYHKDBN become N."""
    code = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}
    text = b"@HD\tVN:1.6\tSO:unknown\n"
    out = bytearray(b"BAM\x01" + struct.pack("<i", len(text)) + text + struct.pack("<i", 0))
    for i, r in enumerate(reads):
        name = ("r%d" % i).encode() + b"\x00"
        fl = 4 if flags is None else flags[i]
        L = len(r)
        nib = [code.get(ch.upper(), 15) for ch in r] + [0]
        seq = bytes((nib[j] << 4) | nib[j + 1] for j in range(0, L + (L & 1), 2)) if L else b""
        qual = b"\xff" * L
        rec = struct.pack("<iiBBHHHiiii", -1, -1, len(name), 0, 4680, 0, fl, L, -1, -1, 0) + name + seq + qual
        out += struct.pack("<i", len(rec)) + rec
    with open(path, "wb") as f:
        data = bytes(out)
        for o in range(0, len(data), 60000):
            f.write(_bgzf_block(data[o:o + 60000]))
        f.write(_bgzf_block(b""))


def sam_bases(read):
    """What the reference makes of a SAM SEQ field (IUPAC_2_DNA, io.c:1394-1404): everything is a base."""
    m = {"c": "c", "b": "c", "s": "c", "y": "c", "1": "c", "g": "g", "k": "g", "2": "g", "t": "t", "3": "t"}
    return "".join(m.get(ch.lower(), "a") for ch in read)
