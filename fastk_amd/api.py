"""ctypes host mirror of the C-ABI in include/fastk_amd.h.

Thin by design: it loads fastk_amd/lib/libfastk_amd.so (hand-written HIP for gfx950) and exposes
the same entry points with numpy-friendly arguments.  There is no CPU path: if the library is
missing or no MI355X is visible, calls raise FastKError.
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libfastk_amd.so")
HIST_BINS = 0x8000

EXPORTS = [
    "fk_get_widths", "fk_default_params", "fk_create", "fk_destroy", "fk_last_error",
    "fk_set_stream", "fk_synchronize", "fk_push_block", "fk_push_device", "fk_finish",
    "fk_write_hist", "fk_write_ktab", "fk_split_supermers", "fk_lsd_sort_records",
    "fk_msd_sort_records", "fk_expand_kmers", "fk_count_kmers", "fk_synth_reads",
    "fk_device_alloc", "fk_device_free", "fk_copy_to_device", "fk_copy_to_host",
    "fk_get_sort_stats", "fk_version", "fk_count_device_reads", "fk_count_device_supermers", "fk_debug_set", "fk_group_records",
    "fk_count_presorted_kmers", "fk_split_supermers_emit", "fk_split_plan", "fk_split_planned",
    "fk_train_block", "fk_count_unsorted_kmers", "fk_debug_get", "fk_push_fastq", "fk_host_alloc",
    "fk_host_free", "fk_bucket_census", "fk_set_bucket_weights", "fk_push_fasta", "fk_merge_tables",
    "fk_write_ktab_ex", "fk_rounds_begin", "fk_rounds_add", "fk_rounds_finish",
    "fk_make_profiles", "fk_write_prof", "fk_set_table", "fk_ktab_idx_bytes", "fk_ktab_split",
    "fk_write_ktab_range", "fk_write_ktab_stub", "fk_split_supermers_emit_pos", "fk_profile_lookup_supermers",
    "fk_profile_scatter", "fk_profile_encode", "fk_reset", "fk_shard_unique_id", "fk_shard_create",
    "fk_release_device", "fk_set_sort_memory", "fk_finish_device", "fk_write_ktab_device", "fk_push_packed", "fk_pack_fixed_reads", "fk_copy_rate", "fk_shard_count", "fk_shard_gather", "fk_shard_write", "fk_shard_destroy", "fk_shard_count_device", "fk_shard_local_result",
    "fk_count_device_packed", "fk_shard_get_stats", "fk_shard_set_write_cutoff", "fk_shard_count_device_packed", "fk_shard_profiles", "fk_shard_write_prof", "fk_write_prof_range", "fk_shard_sum_i64",
]


class FastKError(RuntimeError):
    pass


class Widths(C.Structure):
    _fields_ = [(n, C.c_int) for n in
                ("kmer", "min_len", "max_super", "smer_bytes", "slen_bytes", "smer_word",
                 "kmer_bytes", "kmer_word", "smer_stride", "kmer_stride")]


class Params(C.Structure):
    _fields_ = [("kmer", C.c_int), ("table_cutoff", C.c_int), ("nthreads", C.c_int),
                ("bc_prefix", C.c_int), ("device", C.c_int), ("nbuckets", C.c_int),
                ("hbm_budget", C.c_int64), ("exact_parts", C.c_int), ("split_passes", C.c_int)]


class CResult(C.Structure):
    _fields_ = [("hist", C.c_int64 * HIST_BINS), ("max_inst", C.c_int64), ("ninst", C.c_int64),
                ("nsuper", C.c_int64), ("ndistinct_super", C.c_int64), ("nweighted", C.c_int64),
                ("ndistinct", C.c_int64), ("ntable", C.c_int64),
                ("table", C.POINTER(C.c_uint8)), ("wfirst", C.c_int64 * 256),
                ("ms_split", C.c_double), ("ms_sort_super", C.c_double), ("ms_expand", C.c_double),
                ("ms_sort_kmer", C.c_double), ("ms_count", C.c_double), ("ms_total", C.c_double),
                ("passes_super", C.c_int), ("passes_kmer", C.c_int),
                ("ms_pass_super", C.c_double), ("ms_pass_kmer", C.c_double),
                ("ms_scatter_super", C.c_double), ("ms_scatter_kmer", C.c_double),
                ("ncollapsed", C.c_int64), ("passes_final", C.c_int), ("ms_pass_final", C.c_double),
                ("launches_super", C.c_int64), ("launches_kmer", C.c_int64), ("split_passes", C.c_int),
                ("replay_passes", C.c_int), ("buckets_counted", C.c_int), ("spilled_bytes", C.c_int64), ("ms_table_sort", C.c_double),
                ("nrefs", C.c_int64), ("ms_scatter_final", C.c_double)]


class CProfiles(C.Structure):
    _fields_ = [("nreads", C.c_int64), ("nbytes", C.c_int64), ("data", C.POINTER(C.c_uint8)),
                ("offsets", C.POINTER(C.c_int64)), ("nsplit", C.c_int), ("split", C.POINTER(C.c_int64))]


class ShardStats(C.Structure):
    _fields_ = [("comm_ranks", C.c_int), ("rounds", C.c_int), ("sent_bytes", C.c_int64), ("recv_bytes", C.c_int64),
                ("kept_bytes", C.c_int64), ("exchange_ms", C.c_double), ("gather_sent_bytes", C.c_int64),
                ("gather_exchange_ms", C.c_double), ("gather_sort_ms", C.c_double), ("gather_d2h_ms", C.c_double),
                ("table_entries_written", C.c_int64)]


class SortStats(C.Structure):
    _fields_ = [("passes", C.c_int), ("nelem", C.c_int64), ("rsize", C.c_int),
                ("pass_ms_total", C.c_double), ("scatter_ms_total", C.c_double),
                ("hist_ms", C.c_double)]


_lib = None


def load_library():
    """Load libfastk_amd.so; fail loudly when the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FastKError("%s is missing: build it with `make -C fastk_amd/csrc` "
                         "(or __graft_entry__.build()); there is no CPU fallback" % LIB_PATH)
    # PyTorch-ROCm ships its own copy of the HIP runtime.  A process that loads this library first
    # (binding /opt/rocm's runtime) and imports torch afterwards ends up with two runtimes, and the second
    # one finds no usable GPU ("No HIP GPUs are available").  Importing torch first makes the dynamic
    # linker resolve both to one runtime, whatever order the caller uses later.
    if "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = C.CDLL(LIB_PATH)
    vp, i64, ci = C.c_void_p, C.c_int64, C.c_int
    L.fk_get_widths.argtypes = [ci, C.POINTER(Widths)]
    L.fk_default_params.argtypes = [C.POINTER(Params)]
    L.fk_default_params.restype = None
    L.fk_create.argtypes = [C.POINTER(Params), C.POINTER(vp)]
    L.fk_destroy.argtypes = [vp]
    L.fk_destroy.restype = None
    L.fk_last_error.argtypes = [vp]
    L.fk_last_error.restype = C.c_char_p
    L.fk_set_stream.argtypes = [vp, vp]
    L.fk_synchronize.argtypes = [vp]
    L.fk_push_block.argtypes = [vp, vp, vp, ci, ci, ci]
    L.fk_push_packed.argtypes = [vp, vp, i64, vp, ci, vp, ci, ci, ci]
    L.fk_set_sort_memory.argtypes = [vp, i64, C.c_double]
    L.fk_pack_fixed_reads.argtypes = [vp, vp, i64, C.c_uint32, vp]
    L.fk_copy_rate.argtypes = [vp, vp, vp, i64, C.c_int, C.POINTER(C.c_double)]
    L.fk_train_block.argtypes = [vp, vp, vp, ci]
    L.fk_push_device.argtypes = [vp, vp, i64]
    L.fk_finish.argtypes = [vp, C.POINTER(CResult)]
    L.fk_reset.argtypes = [vp]
    L.fk_shard_unique_id.argtypes = [C.c_char_p]
    L.fk_shard_create.argtypes = [vp, ci, ci, C.c_char_p, C.POINTER(vp)]
    L.fk_shard_count.argtypes = [vp, C.POINTER(CResult)]
    L.fk_shard_count_device.argtypes = [vp, vp, i64, C.POINTER(CResult)]
    L.fk_shard_count_device_packed.argtypes = [vp, vp, i64, vp, i64, vp, i64, C.POINTER(CResult)]
    L.fk_shard_local_result.argtypes = [vp, C.POINTER(CResult)]
    L.fk_shard_write.argtypes = [vp, C.POINTER(CResult), ci, C.c_char_p, C.c_char_p]
    L.fk_shard_gather.argtypes = [vp, C.POINTER(CResult), ci, C.POINTER(vp), C.POINTER(i64)]
    L.fk_shard_get_stats.argtypes = [vp, C.POINTER(ShardStats)]
    L.fk_shard_set_write_cutoff.argtypes = [vp, ci]
    L.fk_shard_destroy.argtypes = [vp]
    L.fk_shard_destroy.restype = None
    L.fk_count_device_reads.argtypes = [vp, vp, i64, ci, C.POINTER(CResult)]
    L.fk_count_device_packed.argtypes = [vp, vp, i64, vp, i64, vp, i64, ci, C.POINTER(CResult)]
    L.fk_count_device_supermers.argtypes = [vp, vp, i64, ci, C.POINTER(CResult)]
    L.fk_write_hist.argtypes = [C.POINTER(CResult), ci, C.c_char_p]
    L.fk_write_ktab.argtypes = [C.POINTER(CResult), ci, ci, ci, C.c_char_p, C.c_char_p]
    L.fk_split_supermers.argtypes = [vp, vp, i64, vp, i64, C.POINTER(i64), C.POINTER(i64), vp]
    L.fk_split_supermers_emit.argtypes = [vp, vp, i64, vp, i64, vp]
    L.fk_split_plan.argtypes = [vp, vp, i64, C.POINTER(i64), vp]
    L.fk_split_planned.argtypes = [vp, vp, i64, vp, i64, vp, vp, C.POINTER(i64)]
    L.fk_lsd_sort_records.argtypes = [vp, i64, vp, vp, ci, C.POINTER(ci), C.POINTER(vp)]
    L.fk_group_records.argtypes = [vp, i64, vp, vp, ci, C.POINTER(vp)]
    L.fk_msd_sort_records.argtypes = [vp, vp, vp, i64, ci, ci, C.POINTER(vp)]
    L.fk_expand_kmers.argtypes = [vp, vp, i64, vp, i64, C.POINTER(i64), C.POINTER(i64),
                                  C.POINTER(i64)]
    L.fk_count_kmers.argtypes = [vp, vp, i64, ci, vp, C.POINTER(i64), C.POINTER(i64), vp, i64,
                                 C.POINTER(i64)]
    L.fk_count_presorted_kmers.argtypes = [vp, vp, i64, ci, ci, vp, C.POINTER(i64), C.POINTER(i64), vp,
                                           i64, C.POINTER(i64)]
    L.fk_synth_reads.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64,
                                 i64, vp]
    L.fk_device_alloc.argtypes = [vp, i64, C.POINTER(vp)]
    L.fk_device_free.argtypes = [vp, vp]
    L.fk_copy_to_device.argtypes = [vp, vp, vp, i64]
    L.fk_copy_to_host.argtypes = [vp, vp, vp, i64]
    L.fk_get_sort_stats.argtypes = [vp, C.POINTER(SortStats)]
    L.fk_debug_set.argtypes = [vp, C.c_char_p, i64]
    L.fk_debug_get.argtypes = [vp, C.c_char_p, C.POINTER(i64)]
    L.fk_push_fastq.argtypes = [vp, vp, i64, C.c_int, C.POINTER(C.c_int), C.POINTER(i64), C.POINTER(i64)]
    L.fk_push_fasta.argtypes = [vp, vp, i64, C.c_int, C.POINTER(C.c_int), C.POINTER(i64), C.POINTER(i64)]
    L.fk_merge_tables.argtypes = [vp, vp, i64, i64, C.POINTER(CResult)]
    L.fk_make_profiles.argtypes = [vp, vp, i64, C.POINTER(CProfiles)]
    L.fk_set_table.argtypes = [vp, vp, i64]
    L.fk_write_prof.argtypes = [C.POINTER(CProfiles), ci, ci, C.c_char_p, C.c_char_p]
    L.fk_ktab_idx_bytes.argtypes = [ci, i64]
    L.fk_ktab_split.argtypes = [C.POINTER(i64), ci, ci, C.POINTER(ci)]
    L.fk_write_ktab_range.argtypes = [vp, i64, ci, ci, C.POINTER(ci), ci, ci, C.c_char_p, C.c_char_p, vp]
    L.fk_write_ktab_stub.argtypes = [ci, ci, ci, ci, vp, C.c_char_p, C.c_char_p]
    L.fk_split_supermers_emit_pos.argtypes = [vp, vp, i64, vp, i64, C.POINTER(i64), vp]
    L.fk_profile_lookup_supermers.argtypes = [vp, vp, i64, vp, i64, C.POINTER(i64)]
    L.fk_profile_scatter.argtypes = [vp, vp, vp, i64, vp, i64, ci]
    L.fk_profile_encode.argtypes = [vp, vp, i64, C.POINTER(CProfiles)]
    L.fk_rounds_begin.argtypes = [vp]
    L.fk_rounds_add.argtypes = [vp, vp, i64]
    L.fk_rounds_finish.argtypes = [vp, C.c_int, C.POINTER(CResult)]
    L.fk_host_alloc.argtypes = [i64, C.POINTER(vp)]
    L.fk_host_free.argtypes = [vp]
    L.fk_bucket_census.argtypes = [vp, vp, i64, C.POINTER(i64)]
    L.fk_set_bucket_weights.argtypes = [vp, C.POINTER(i64)]
    L.fk_count_unsorted_kmers.argtypes = [vp, vp, vp, i64, C.c_int, C.POINTER(i64), C.POINTER(i64),
                                          C.POINTER(i64), C.POINTER(vp), C.POINTER(i64)]
    L.fk_version.restype = C.c_char_p
    _lib = L
    return L


def widths(kmer):
    w = Widths()
    if load_library().fk_get_widths(kmer, C.byref(w)) != 0:
        raise FastKError("unsupported k = %d" % kmer)
    return w


class Result:
    """fk_result copied into numpy arrays."""

    def __init__(self, cres, kmer_word):
        self.hist = np.ctypeslib.as_array(cres.hist).copy()
        for f in ("max_inst", "ninst", "nsuper", "ndistinct_super", "nweighted", "ndistinct",
                  "ntable"):
            setattr(self, f, int(getattr(cres, f)))
        self.wfirst = np.ctypeslib.as_array(cres.wfirst).copy()
        self.ms = {k: float(getattr(cres, "ms_" + k))
                   for k in ("split", "sort_super", "expand", "sort_kmer", "count", "total")}
        self.passes_super, self.passes_kmer = int(cres.passes_super), int(cres.passes_kmer)
        self.ms_pass_super, self.ms_pass_kmer = float(cres.ms_pass_super), float(cres.ms_pass_kmer)
        self.ms_scatter_super = float(cres.ms_scatter_super)
        self.ms_scatter_kmer = float(cres.ms_scatter_kmer)
        self.ncollapsed = int(cres.ncollapsed)
        self.passes_final, self.ms_pass_final = int(cres.passes_final), float(cres.ms_pass_final)
        self.launches_super, self.launches_kmer = int(cres.launches_super), int(cres.launches_kmer)
        self.split_passes, self.buckets_counted = int(cres.split_passes), int(cres.buckets_counted)
        self.replay_passes = int(cres.replay_passes)
        self.spilled_bytes, self.ms_table_sort = int(cres.spilled_bytes), float(cres.ms_table_sort)
        self.nrefs = int(cres.nrefs)
        self.ms_scatter_final = float(cres.ms_scatter_final)
        if self.ntable > 0 and cres.table:
            self.table = np.ctypeslib.as_array(cres.table, shape=(self.ntable, kmer_word)).copy()
        else:
            self.table = np.zeros((0, kmer_word), dtype=np.uint8)
        self._c = cres


class DeviceBuffer:
    """HBM allocation owned by a Context."""

    def __init__(self, ctx, nbytes):
        self.ctx = ctx
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        ctx._ck(ctx.L.fk_device_alloc(ctx.h, self.nbytes, C.byref(p)))
        self.ptr = p.value

    def upload(self, arr):
        a = np.ascontiguousarray(arr)
        assert a.nbytes <= self.nbytes
        self.ctx._ck(self.ctx.L.fk_copy_to_device(self.ctx.h, self.ptr, a.ctypes.data, a.nbytes))
        return self

    def download(self, nbytes=None, dtype=np.uint8, ptr=None):
        n = self.nbytes if nbytes is None else int(nbytes)
        out = np.empty(n, dtype=np.uint8)
        if n:
            self.ctx._ck(self.ctx.L.fk_copy_to_host(self.ctx.h, out.ctypes.data,
                                                    self.ptr if ptr is None else ptr, n))
        return out.view(dtype)

    def free(self):
        if self.ptr:
            self.ctx.L.fk_device_free(self.ctx.h, self.ptr)
            self.ptr = None


class Context:
    """fk_ctx wrapper: one per process / GPU."""

    def __init__(self, kmer=40, table_cutoff=0, nthreads=4, bc_prefix=0, device=0, nbuckets=1,
                 exact_parts=False, hbm_budget=0, split_passes=0):
        self.L = load_library()
        p = Params()
        self.L.fk_default_params(C.byref(p))
        p.kmer, p.table_cutoff, p.nthreads = kmer, table_cutoff, nthreads
        p.bc_prefix, p.device, p.nbuckets = bc_prefix, device, nbuckets
        p.exact_parts = int(exact_parts)        # (True / 1; 2: a run that will make profiles, see fk_params.exact_parts)
        p.hbm_budget = int(hbm_budget)
        p.split_passes = int(split_passes)
        self.params = p
        self.h = C.c_void_p()
        rc = self.L.fk_create(C.byref(p), C.byref(self.h))
        if rc != 0:
            raise FastKError("fk_create failed (%d): %s" %
                             (rc, self.L.fk_last_error(None).decode()))
        self.w = widths(kmer)
        self.kmer = kmer

    def _ck(self, rc):
        if rc != 0:
            raise FastKError("libfastk_amd error %d: %s" %
                             (rc, self.L.fk_last_error(self.h).decode()))

    def close(self):
        if self.h:
            self.L.fk_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_stream(self, stream_ptr):
        self._ck(self.L.fk_set_stream(self.h, stream_ptr))

    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    # ---- whole path -------------------------------------------------------------------
    def push_block(self, bases, boff, rem=0, tid=0):
        """bases: uint8 array of 0-terminated reads, boff: int32 offsets (DATA_BLOCK, FastK.h:87)."""
        b = np.ascontiguousarray(bases, dtype=np.uint8)
        o = np.ascontiguousarray(boff, dtype=np.int32)
        self._ck(self.L.fk_push_block(self.h, b.ctypes.data, o.ctypes.data, len(o) - 1, rem, tid))

    def push_packed(self, codes, nbases, rlen, inv=None, rem=0, tid=0):
        """codes: uint8 array, 4 bases per byte (first base in the high bits), reads back to back; rlen: int32 lengths;
        inv: int64 (n, 2) array of (first base, length) stretches without acgt."""
        c = np.ascontiguousarray(codes, dtype=np.uint8)
        r = np.ascontiguousarray(rlen, dtype=np.int32)
        v = np.ascontiguousarray(inv if inv is not None else np.zeros((0, 2)), dtype=np.int64).reshape(-1, 2)
        self._ck(self.L.fk_push_packed(self.h, c.ctypes.data, int(nbases), r.ctypes.data, len(r),
                                       v.ctypes.data if len(v) else None, len(v), rem, tid))

    def bucket_census(self, sample):
        """sample: uint8 array of reads (host).  Returns the int64[16384] work census per minimizer rank."""
        a = np.ascontiguousarray(sample, dtype=np.uint8)
        counts = np.zeros(16384, dtype=np.int64)
        self._ck(self.L.fk_bucket_census(self.h, a.ctypes.data, a.nbytes,
                                         counts.ctypes.data_as(C.POINTER(C.c_int64))))
        return counts

    def set_bucket_weights(self, counts):
        c = np.ascontiguousarray(counts, dtype=np.int64)
        self._ck(self.L.fk_set_bucket_weights(self.h, c.ctypes.data_as(C.POINTER(C.c_int64))))

    def push_fastq(self, raw, phase=0, hoco=False):
        """raw: bytes / uint8 array holding any piece of a FASTQ file; returns (phase, reads, bases)."""
        a = np.frombuffer(raw, dtype=np.uint8) if isinstance(raw, (bytes, bytearray)) else \
            np.ascontiguousarray(raw, dtype=np.uint8)
        ph, nr, nb = C.c_int(phase), C.c_int64(0), C.c_int64(0)
        self._ck(self.L.fk_push_fastq(self.h, a.ctypes.data, a.nbytes, 1 if hoco else 0, C.byref(ph), C.byref(nr),
                                      C.byref(nb)))
        return ph.value, nr.value, nb.value

    def push_fasta(self, raw, state=2, last=False):
        """raw: any piece of a FASTA file; state 2 at file start; returns (state, records, bases)."""
        a = np.frombuffer(raw, dtype=np.uint8) if isinstance(raw, (bytes, bytearray)) else \
            np.ascontiguousarray(raw, dtype=np.uint8)
        st, nr, nb = C.c_int(state), C.c_int64(0), C.c_int64(0)
        self._ck(self.L.fk_push_fasta(self.h, a.ctypes.data if a.nbytes else None, a.nbytes,
                                      1 if last else 0, C.byref(st), C.byref(nr), C.byref(nb)))
        return st.value, nr.value, nb.value

    def rounds_begin(self):
        self._ck(self.L.fk_rounds_begin(self.h))

    def rounds_add(self, ptr, nsuper):
        self._ck(self.L.fk_rounds_add(self.h, ptr, nsuper))

    def rounds_finish(self, fetch_table=False):
        r = CResult()
        self._ck(self.L.fk_rounds_finish(self.h, 1 if fetch_table else 0, C.byref(r)))
        return Result(r, self.w.kmer_word)

    def split_emit_pos(self, reads_ptr, nbytes, out_ptr, cap, counts, pos_ptr):
        bc = (C.c_int64 * 256)(*[int(c) for c in counts])
        self._ck(self.L.fk_split_supermers_emit_pos(self.h, reads_ptr, nbytes, out_ptr, cap, bc, pos_ptr))

    def profile_lookup_supermers(self, smers_ptr, nsuper, counts_ptr=None, cap=0):
        ni = C.c_int64()
        self._ck(self.L.fk_profile_lookup_supermers(self.h, smers_ptr, nsuper, counts_ptr, cap, C.byref(ni)))
        return ni.value

    def profile_scatter(self, smers_ptr, pos_ptr, nsuper, counts_ptr, nbytes, reset=True):
        self._ck(self.L.fk_profile_scatter(self.h, smers_ptr, pos_ptr, nsuper, counts_ptr, nbytes, 1 if reset else 0))

    def profile_encode(self, reads_ptr, nbytes):
        pr = CProfiles()
        self._ck(self.L.fk_profile_encode(self.h, reads_ptr, nbytes, C.byref(pr)))
        n = pr.nreads
        offs = np.ctypeslib.as_array(pr.offsets, shape=(n + 1,)).copy() if pr.offsets else np.zeros(1, dtype=np.int64)
        data = np.ctypeslib.as_array(pr.data, shape=(pr.nbytes,)).copy() if pr.nbytes > 0 \
            else np.zeros(0, dtype=np.uint8)
        return data, offs

    def set_table(self, records):
        """(n, KMER_WORD) uint8 entries, any order: the dictionary of the next make_profiles."""
        a = np.ascontiguousarray(records, dtype=np.uint8)
        self._ck(self.L.fk_set_table(self.h, a.ctypes.data if a.shape[0] else None, a.shape[0]))

    def make_profiles(self, ptr=None, nbytes=0, outdir=None, root=None, nparts=1):
        """Profiles of the reads just counted (cutoff 1, resident run): returns (data, offsets) as
        numpy copies; with outdir/root also writes <root>.prof + hidden parts."""
        pr = CProfiles()
        self._ck(self.L.fk_make_profiles(self.h, ptr, nbytes, C.byref(pr)))
        n = pr.nreads
        offs = np.ctypeslib.as_array(pr.offsets, shape=(n + 1,)).copy() if n >= 0 and pr.offsets \
            else np.zeros(1, dtype=np.int64)
        data = np.ctypeslib.as_array(pr.data, shape=(pr.nbytes,)).copy() if pr.nbytes > 0 \
            else np.zeros(0, dtype=np.uint8)
        if outdir is not None:
            self._ck(self.L.fk_write_prof(C.byref(pr), self.w.kmer, nparts, outdir.encode(), root.encode()))
        return data, offs

    def merge_tables(self, records, max_inst_in=0):
        """records: (n, KMER_WORD) uint8 entries of all input tables; returns the merged Result."""
        a = np.ascontiguousarray(records, dtype=np.uint8)
        r = CResult()
        self._ck(self.L.fk_merge_tables(self.h, a.ctypes.data, a.shape[0], int(max_inst_in), C.byref(r)))
        return Result(r, self.w.kmer_word)

    def reset(self):
        self._ck(self.L.fk_reset(self.h))

    def push_device(self, ptr, nbytes):
        self._ck(self.L.fk_push_device(self.h, ptr, nbytes))

    def finish(self):
        r = CResult()
        self._ck(self.L.fk_finish(self.h, C.byref(r)))
        return Result(r, self.w.kmer_word)

    def count_device_reads(self, ptr, nbytes, fetch_table=False):
        r = CResult()
        self._ck(self.L.fk_count_device_reads(self.h, ptr, nbytes, 1 if fetch_table else 0,
                                              C.byref(r)))
        return Result(r, self.w.kmer_word)

    def count_device_packed(self, codes_ptr, nbases, roff_ptr, nreads, inv_ptr=None, ninv=0, fetch_table=False):
        """reads resident in two bits per base (device pointers: codes, int64 roff[nreads + 1], int64 inv pairs)"""
        r = CResult()
        self._ck(self.L.fk_count_device_packed(self.h, codes_ptr, nbases, roff_ptr, nreads, inv_ptr, ninv,
                                               1 if fetch_table else 0, C.byref(r)))
        return Result(r, self.w.kmer_word)

    def count_device_supermers(self, ptr, nsuper, fetch_table=False):
        r = CResult()
        self._ck(self.L.fk_count_device_supermers(self.h, ptr, nsuper, 1 if fetch_table else 0,
                                                  C.byref(r)))
        return Result(r, self.w.kmer_word)

    def finish_device(self):
        """fk_finish without the host copy of the table (Result.table is None)."""
        r = CResult()
        self._ck(self.L.fk_finish_device(self.h, C.byref(r)))
        return Result(r, self.w.kmer_word)

    def write_ktab_device(self, res, outdir, root, nthreads=None):
        self._ck(self.L.fk_write_ktab_device(self.h, C.byref(res._c), nthreads or self.params.nthreads,
                                             outdir.encode(), root.encode()))

    def write_hist(self, res, path):
        self._ck(self.L.fk_write_hist(C.byref(res._c), self.kmer, path.encode()))

    def write_ktab(self, res, outdir, root, nthreads=None):
        # res._c.table points into ctx-owned memory that is valid until the next finish
        self._ck(self.L.fk_write_ktab(C.byref(res._c), self.kmer, self.params.table_cutoff,
                                      nthreads or self.params.nthreads, outdir.encode(),
                                      root.encode()))

    # ---- stages -----------------------------------------------------------------------
    def synth_reads(self, seed, genome_len, read_len, err_ppm, first_read, nreads, buf=None):
        n = nreads * (read_len + 1)
        buf = buf or self.alloc(n + 64)
        self._ck(self.L.fk_synth_reads(self.h, seed, genome_len, read_len, err_ppm, first_read,
                                       nreads, buf.ptr))
        self._ck(self.L.fk_synchronize(self.h))
        return buf, n

    def copy_rate(self, dst_ptr, src_ptr, nbytes, reps=5):
        """(read + write) GB/s of a plain uint4 copy kernel between two device buffers (measurement helper)."""
        g = C.c_double()
        self._ck(self.L.fk_copy_rate(self.h, dst_ptr, src_ptr, nbytes, reps, C.byref(g)))
        return g.value

    def split(self, reads_ptr, nbytes, out_ptr=None, cap=0):
        ns, ni = C.c_int64(), C.c_int64()
        bc = (C.c_int64 * 256)()
        self._ck(self.L.fk_split_supermers(self.h, reads_ptr, nbytes, out_ptr, cap, C.byref(ns),
                                           C.byref(ni), bc))
        return ns.value, ni.value, list(bc)[:self.params.nbuckets]

    def split_plan(self, reads_ptr, nbytes):
        cap = C.c_int64()
        offs = (C.c_int64 * 257)()
        self._ck(self.L.fk_split_plan(self.h, reads_ptr, nbytes, C.byref(cap), offs))
        return cap.value, list(offs)[:self.params.nbuckets + 1]

    def split_planned(self, reads_ptr, nbytes, out_ptr, cap, offsets):
        """Returns (counts, ninst) or None when a planned region overflowed (FK_ESTATE)."""
        offs = (C.c_int64 * 257)(*offsets)
        cnt = (C.c_int64 * 256)()
        ni = C.c_int64()
        rc = self.L.fk_split_planned(self.h, reads_ptr, nbytes, out_ptr, cap, offs, cnt, C.byref(ni))
        if rc == -6:
            return None
        self._ck(rc)
        return list(cnt)[:self.params.nbuckets], ni.value

    def split_emit(self, reads_ptr, nbytes, out_ptr, cap, counts):
        bc = (C.c_int64 * 256)(*counts)
        self._ck(self.L.fk_split_supermers_emit(self.h, reads_ptr, nbytes, out_ptr, cap, bc))

    def lsd_sort(self, src_ptr, trg_ptr, nelem, rsize, byte_list):
        bl = (C.c_int * (len(byte_list) + 1))(*(list(byte_list) + [-1]))
        res = C.c_void_p()
        self._ck(self.L.fk_lsd_sort_records(self.h, nelem, src_ptr, trg_ptr, rsize, bl,
                                            C.byref(res)))
        return res.value

    def group(self, src_ptr, trg_ptr, nelem, rsize):
        res = C.c_void_p()
        self._ck(self.L.fk_group_records(self.h, nelem, src_ptr, trg_ptr, rsize, C.byref(res)))
        return res.value

    def msd_sort(self, arr_ptr, tmp_ptr, nelem, rsize, ksize):
        res = C.c_void_p()
        self._ck(self.L.fk_msd_sort_records(self.h, arr_ptr, tmp_ptr, nelem, rsize, ksize,
                                            C.byref(res)))
        return res.value

    def sort_stats(self):
        st = SortStats()
        self._ck(self.L.fk_get_sort_stats(self.h, C.byref(st)))
        return dict(passes=st.passes, nelem=st.nelem, rsize=st.rsize,
                    pass_ms_total=st.pass_ms_total, scatter_ms_total=st.scatter_ms_total,
                    hist_ms=st.hist_ms)

    def expand(self, smers_ptr, nsuper, out_ptr=None, cap=0):
        nw, nd, ov = C.c_int64(), C.c_int64(), C.c_int64()
        self._ck(self.L.fk_expand_kmers(self.h, smers_ptr, nsuper, out_ptr, cap, C.byref(nw),
                                        C.byref(nd), C.byref(ov)))
        return nw.value, nd.value, ov.value

    def debug_set(self, key, value):
        self._ck(self.L.fk_debug_set(self.h, key.encode(), int(value)))

    def debug_get(self, key):
        v = C.c_int64()
        self._ck(self.L.fk_debug_get(self.h, key.encode(), C.byref(v)))
        return v.value

    def count_unsorted(self, kmers_ptr, tmp_ptr, nweighted, cutoff):
        """fk_count_unsorted_kmers: returns (hist, max_inst, ndistinct, ntable, table device ptr)."""
        hist = np.zeros(HIST_BINS, dtype=np.int64)
        mi, nd, nt = C.c_int64(0), C.c_int64(), C.c_int64()
        tp = C.c_void_p()
        self._ck(self.L.fk_count_unsorted_kmers(self.h, kmers_ptr, tmp_ptr, nweighted, cutoff,
                                                hist.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(mi),
                                                C.byref(nd), C.byref(tp), C.byref(nt)))
        return hist, mi.value, nd.value, nt.value, tp.value

    def count(self, kmers_ptr, nweighted, cutoff, table_ptr=None, cap=0, sorted_bytes=None):
        hist = np.zeros(HIST_BINS, dtype=np.int64)
        mi, nd, nt = C.c_int64(0), C.c_int64(), C.c_int64()
        if sorted_bytes is None:
            rc = self.L.fk_count_kmers(self.h, kmers_ptr, nweighted, cutoff, hist.ctypes.data,
                                       C.byref(mi), C.byref(nd), table_ptr, cap, C.byref(nt))
        else:
            rc = self.L.fk_count_presorted_kmers(self.h, kmers_ptr, nweighted, cutoff, sorted_bytes,
                                                 hist.ctypes.data, C.byref(mi), C.byref(nd),
                                                 table_ptr, cap, C.byref(nt))
            if rc == -6:
                return None
        self._ck(rc)
        return hist, mi.value, nd.value, nt.value


class Shard:
    """fk_shard wrapper: this process is one rank of a sharded run (include/fastk_amd.h).  The context must have
    been created with nbuckets = world * rounds."""

    def __init__(self, ctx, rank, world, unique_id):
        self.ctx, self.rank, self.world = ctx, rank, world
        self.h = C.c_void_p()
        ctx._ck(ctx.L.fk_shard_create(ctx.h, rank, world, bytes(unique_id), C.byref(self.h)))

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        L = load_library()
        if L.fk_shard_unique_id(buf) != 0:
            raise FastKError("fk_shard_unique_id failed: %s" % L.fk_last_error(None).decode())
        return buf.raw

    def count(self, ptr=None, nbytes=0):
        """Exchange + count + all-reduce over the pushed reads, or over resident reads at ptr.  Returns the
        GLOBAL Result (table stays sharded in HBM)."""
        r = CResult()
        if ptr is None:
            self.ctx._ck(self.ctx.L.fk_shard_count(self.h, C.byref(r)))
        else:
            self.ctx._ck(self.ctx.L.fk_shard_count_device(self.h, ptr, nbytes, C.byref(r)))
        return Result(r, self.ctx.w.kmer_word)

    def count_packed(self, codes_ptr, nbases, roff_ptr, nreads, inv_ptr=None, ninv=0):
        """The same over a stripe resident in two bits per base (the arguments of Context.count_device_packed)."""
        r = CResult()
        self.ctx._ck(self.ctx.L.fk_shard_count_device_packed(self.h, codes_ptr, nbases, roff_ptr, nreads, inv_ptr, ninv,
                                                             C.byref(r)))
        return Result(r, self.ctx.w.kmer_word)

    def local_result(self):
        r = CResult()
        self.ctx._ck(self.ctx.L.fk_shard_local_result(self.h, C.byref(r)))
        return Result(r, self.ctx.w.kmer_word)

    def gather(self, res, nparts, copy=False):
        """C3 alone: this rank's first-byte range of the whole table (parts rank*m+1 .. rank*m+m of an nparts-part
        .ktab), ordered, in pinned host memory owned by the shard.  Returns the entry count, or with copy=True the
        entries as an (n, KMER_BYTES + 2) uint8 array."""
        p, n = C.c_void_p(), C.c_int64()
        self.ctx._ck(self.ctx.L.fk_shard_gather(self.h, C.byref(res._c), nparts, C.byref(p), C.byref(n)))
        if not copy:
            return n.value
        kw = self.ctx.w.kmer_word
        if n.value == 0:
            return np.zeros((0, kw), dtype=np.uint8)
        return np.ctypeslib.as_array(C.cast(p.value, C.POINTER(C.c_uint8)), shape=(n.value * kw,)).reshape(n.value, kw).copy()

    def write(self, res, nparts, outdir, root):
        self.ctx._ck(self.ctx.L.fk_shard_write(self.h, C.byref(res._c), nparts, outdir.encode(), root.encode()))

    def stats(self):
        """what this rank's last count (C1) and gather (C3) moved, over how many RCCL ranks"""
        st = ShardStats()
        self.ctx._ck(self.ctx.L.fk_shard_get_stats(self.h, C.byref(st)))
        return dict((n, getattr(st, n)) for n, _ in ShardStats._fields_)

    def close(self):
        if self.h:
            self.ctx.L.fk_shard_destroy(self.h)
            self.h = None


def write_files(kmer, table_cutoff, nthreads, hist, max_inst, table, outdir, root, wfirst=None):
    """Write <root>.hist and, with table_cutoff > 0, <root>.ktab + hidden parts from host arrays with
    the library's writers (host-only C code: works without a GPU).  table: (n, KMER_WORD) uint8,
    sorted.  Part boundaries follow wfirst (first-byte census of the weighted k-mers, split.c's
    Table_Split input) when given, else the first-byte census of the table itself."""
    L = load_library()
    r = CResult()
    h = np.asarray(hist, dtype=np.int64)
    for i in range(1, HIST_BINS):
        r.hist[i] = int(h[i])
    r.max_inst = int(max_inst)
    t = np.ascontiguousarray(table, dtype=np.uint8)
    r.ntable = t.shape[0]
    r.table = t.ctypes.data_as(C.POINTER(C.c_uint8))
    if wfirst is not None and int(np.sum(wfirst)) > 0:
        census = np.asarray(wfirst, dtype=np.int64)
    else:
        census = np.bincount(t[:, 0], minlength=256) if t.shape[0] else np.zeros(256, dtype=np.int64)
    for x in range(256):
        r.wfirst[x] = int(census[x])
    r.nweighted = int(census.sum())
    if L.fk_write_hist(C.byref(r), kmer, os.path.join(outdir, root + ".hist").encode()) != 0:
        raise FastKError(L.fk_last_error(None).decode())
    if table_cutoff > 0 and L.fk_write_ktab(C.byref(r), kmer, table_cutoff, nthreads, outdir.encode(),
                                            root.encode()) != 0:
        raise FastKError(L.fk_last_error(None).decode())


# ---- one table written by several ranks (host-only C code behind these) ------------------------

def ktab_idx_bytes(kmer, ntable):
    return int(load_library().fk_ktab_idx_bytes(kmer, int(ntable)))


def ktab_split(wfirst, kmer, nparts):
    """First-byte boundaries of nparts parts (Table_Split) from the summed weighted k-mer census."""
    w = (C.c_int64 * 256)(*[int(x) for x in wfirst])
    out = (C.c_int * (nparts + 1))()
    if load_library().fk_ktab_split(w, kmer, nparts, out) != 0:
        raise FastKError("fk_ktab_split failed")
    return list(out)


def write_ktab_range(records, kmer, idx_bytes, split, part0, nhere, outdir, root):
    """Writes parts part0 .. part0+nhere-1 from sorted (n, KMER_WORD) records; returns the per-prefix
    entry counts (int64[256**idx_bytes]) of what was written."""
    L = load_library()
    a = np.ascontiguousarray(records, dtype=np.uint8)
    sp = (C.c_int * len(split))(*split)
    cnt = np.zeros(256 ** idx_bytes, dtype=np.int64)
    if L.fk_write_ktab_range(a.ctypes.data if a.shape[0] else None, a.shape[0], kmer, idx_bytes, sp, part0,
                             nhere, outdir.encode(), root.encode(), cnt.ctypes.data) != 0:
        raise FastKError(L.fk_last_error(None).decode())
    return cnt


def write_ktab_stub(kmer, nparts, table_cutoff, idx_bytes, prefix_counts, outdir, root):
    L = load_library()
    c = np.ascontiguousarray(prefix_counts, dtype=np.int64)
    if L.fk_write_ktab_stub(kmer, nparts, table_cutoff, idx_bytes, c.ctypes.data, outdir.encode(),
                            root.encode()) != 0:
        raise FastKError(L.fk_last_error(None).decode())
