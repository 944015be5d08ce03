// fk_common.h -- shared declarations for the gfx950 k-mer counting library (internal).
#pragma once

#ifdef FK_HOST_EMU                     // tests/csrc/hip_emu.h: a kernel's source run on the CPU by the CPU tests (not shipped)
#include "hip_emu.h"
#else
#include <hip/hip_runtime.h>
#define FK_DYN_LDS(type, name) extern __shared__ type name[]
#define FK_DYN_LDS_ALIGNED(type, name, al) extern __shared__ __attribute__((aligned(al))) type name[]
#define FK_BALLOT_ACTIVE(p) __ballot(p)                // a ballot inside a loop that the lanes of a wave leave one by one
#define FK_WAVE_UNIFORM(x) (x)                         // a value every lane of a wave reads in the same instruction
#define FK_EMU_WAVE_SYNC() ((void) 0)                  // (tests/csrc/hip_emu.h: where a kernel counts on a wave's lanes moving together)
#define FK_OPAQUE(x) asm volatile("" : "+v"(x))       // a value the optimiser cannot see through
#define FK_KEEP(x)   asm volatile("" :: "v"(x))        // a value that must be computed
#endif
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/fastk_amd.h"

#define FK_WAVE 64

typedef unsigned long long u64;
typedef unsigned int       u32;

#define FK_NSLOTS 56

// Reads in two bits per base (fk_push_packed, fk_count_device_packed): the splitter's tile loader reads this form
// directly.  codes = 16 bases per little-endian dword, the first base in the two high bits of the first byte (a c g t =
// 0 1 2 3, the .ktab encoding) -- no terminators: read r is the positions roff[r] .. roff[r+1]-1, and a k-mer is valid
// when it lies inside one read and touches none of the stretches inv[2j] .. inv[2j] + inv[2j+1] - 1 (sorted, disjoint:
// N runs, the padding behind a pushed block).  Passed beside (d_bases = codes, nbytes = positions) to the fkx_split*
// entry points; NULL there means 0-terminated ASCII reads.
struct fk_pkview
{ const int64_t *roff;      // [nreads + 1], device
  int64_t        nreads;
  const int64_t *inv;       // [2 * ninv] (first position, length), device
  int64_t        ninv;
};

// the packed reads a context was pushed (one store per read buffer: a chunk fills while the previous one is split)
struct fk_pkstore
{ int64_t *roff;            // device
  int64_t  roff_cap;        // entries
  int64_t  nreads;
  int64_t *inv;             // device, pairs
  int64_t  inv_cap;         // pairs
  int64_t  ninv;
  int64_t  npos;            // positions used in the read buffer (a multiple of 16 after every push)
};

// Minimizers are canonical 7-mers ordered by a bijective mix of their 14-bit code (odd multiplies and
// xor-shifts are invertible mod 2^14, so distinct 7-mers never tie): FK_NRANKS rank values, of which
// the 8192 images of canonical codes occur.
#define FK_EXACT_MAXPARTS 32                // buckets an exact_parts run follows the reference's scheme to (fk_scheme.hip)
#define FK_REGION_SLACK 32768              // records of room a streamed bucket region needs beyond its exact count (ragged stream ends)
#define FKX_TABLE_FULL 100                   // fkx_aggregate: the table buffer took fewer records than qualified (internal)
#define FK_CBASE_EXTRA ((2 << 20) + 64)      // partition sums of the chunk scan behind the chunk bases (fk_split.hip)
#define FK_CURSOR_STRIDE 512
#define FK_MIN_LEN 7
#define FK_NRANKS  16384

#ifdef __HIPCC__
__host__ __device__
#endif
static inline uint32_t fk_mrank14(uint32_t c)
{ // 14 product bits of (c ^ a) * b, both factors below 2^14: three instructions per position on the device (the
  // splitter is bound by VALU issue).  Not a bijection -- 7,905 of the 8,192 canonical codes keep a rank of their
  // own -- and it need not be: the bucket is a function of the RANK, so tied 7-mers share it.  The xor keeps
  // AAAAAAA/TTTTTTT away from rank 0: as the smallest rank it won 8 % of all windows on an AT-rich sequence with
  // poly-A tracts (largest bucket of 48 after the weighted deal: 4.1 x the mean; with this order 1.0 x).
  return ((((c ^ 0x1B3u) * 0x2F65u) >> 9) & 0x3fffu);
}

struct fk_chunk
{ void    *run[256];        // records of bucket b: cnt[b] of them at run[b] (HBM slab or pinned host buffer)
  int64_t  cnt[256];
  int64_t  total;
  int      on_host;         // records live in pinned host memory (spilled), not in HBM
  int      spill_slot;      // on_host: index into fk_ctx.spill_buf
};

// HBM store of the chunks' records: large slabs filled by bump allocation, kept until fk_destroy and
// filled again from the start by the next run (allocating and freeing a buffer per chunk instead
// stalled for seconds inside hipMalloc once the address space had been through one run)
struct fk_slab
{ char    *ptr;
  int64_t  cap, used;
};

struct fk_spill_buf         // pinned host buffer for one spilled chunk, kept for the next run
{ void    *ptr;
  int64_t  cap;
  int      in_use;
};

struct fk_block       // one fk_push_block call: which input thread, how many reads
{ int      tid;
  int      rem;      // > 0: the block's last read continues in the thread's next block
  int64_t  nreads;
  int64_t  pk_pos, pk_nbases, pk_read0, pk_inv0, pk_ninv;   // fk_push_packed: first position, bases, first read, first stretch
                                                            // and stretches of the block in the store
};

struct fk_ctx
{ fk_params  prm;
  fk_widths  wid;
  int        device;
  hipStream_t stream;
  bool       own_stream;
  char       err[512];

  // minimizer scheme (device tables)
  uint8_t   *d_mbucket;   // [FK_NRANKS] bucket of a minimizer RANK
  uint8_t   *d_mbucket_pass; // [FK_NRANKS] the same for one group pass of a multi-pass split (0xFF = not now)
  uint8_t    h_mbucket[FK_NRANKS];
  uint8_t   *h_mbucket_pass; // pinned staging of d_mbucket_pass
  u64       *d_plan;         // [32][256] spread bucket counters of the sampled split plan
  u64       *d_cursors;      // [(256 * 8 streams + 64) * FK_CURSOR_STRIDE] write cursors of the streamed split emit, 4 KB apart,
                             // then the 64 cursors of the entry sub-regions
  int64_t    ent_cap;        // multi-pass split with replay: room (entries) in FK_SLOT_ENT
  int64_t    ent_ntiles;     // tiles the recorded entries belong to
  bool       ent_valid;      // the last recording pass fitted
  int64_t    ent_totals[256];// exact super-mers per bucket, from the recording pass's count table

  // small device scratch (counters, histograms)
  u64       *d_scratch;   // 64 KB
  u64       *h_scratch;   // pinned mirror

  // radix sort workspace
  u64       *d_digit_hist; // [32*256]
  fk_sort_stats sort_stats;
  u64        rx_attr_done; // wide scatter instantiations whose dynamic-LDS attribute is set on this context's device
  int        rx_top_pbytes;// fkx_lsd_sort_top: leading bytes the records are in order on afterwards

  // streaming interface state
  char      *d_reads;      // pushed reads (HBM)
  int64_t    reads_len, reads_cap;
  // chunked ingest: host-to-device copies run on copy_stream into one of two read buffers while the
  // other one is split into super-mers by a helper thread on `stream`
  hipStream_t copy_stream;
  hipEvent_t  reads_ev;     // all copies into the buffer handed to the helper have been issued before it
  // exact_parts with several buckets: the reference's scheme (fk_scheme.hip)
  int64_t    sort_memory;   // -M in bytes as the reference counts them (0: one bucket)
  double     input_ratio;   // whole input / training block, in the reference's file bytes (0: by bases)
  int       *min_part, *d_min_part;   // Min_Part: the prefix trie with bucket numbers at its leaves
  int        scheme_pad, scheme_states, scheme_nparts;
  int        exact_tran[4];           // the base ranking the last exact_parts split used (the profile stitcher needs it)
  bool       exact_tran_set;
  bool       pf_own_reads;            // fk_make_profiles is encoding the pushed reads themselves
  int64_t    exact_wfirst[256];       // first-byte census of bucket 0's weighted k-mers (Table_Split's input)
  // fk_push_packed: the reads stay in two bits per base -- the codes in d_reads (reads_len counts its bytes), read
  // offsets and invalid stretches in pk[pk_cur] (pk[pk_cur ^ 1] belongs to the chunk the flush helper is splitting)
  bool       pk_mode;       // the reads pushed so far are packed (a run takes its reads in ONE form)
  int        pk_cur;
  struct fk_pkstore pk[2];
  int64_t   *h_pk;          // pinned staging of a push's offsets + stretches
  int64_t    h_pk_cap;      // bytes
  int64_t    pk_ascii_len;  // bytes the reads pushed so far take as 0-terminated ASCII (what fkx_unpack_store restores)
  int        push_form;     // 0: nothing pushed yet, 1: ASCII / text pushes, 2: fk_push_packed (until fk_reset)
  // fk_finish_device: where every ib-byte prefix of the sorted table ends (the index of the .ktab stub), for
  // fk_write_ktab_device; valid for a table of ktab_ends_ntab entries
  int64_t   *ktab_ends;
  int64_t    ktab_first[257];   // ... and where every first key byte begins
  unsigned char *h_wstage;      // pinned staging of the part writers (two pieces each), made before the release starts
  int64_t    wstage_cap;
  hipStream_t wstream[4];       // ... and their streams (writer t uses wstream[t % 4]; creating one per writer takes longer than the writing)
  int64_t    ktab_ends_ntab;
  int        ktab_ends_ib;
  char      *d_reads_alt;   // the idle read buffer (NULL until first needed)
  int64_t    reads_cap_alt;
  void      *flush_thread;  // std::thread * of the running flush, NULL if none
  int        flush_rc;      // its result
  char       flush_err[512];
  struct fk_slab *slabs;
  int        nslabs, slabs_cap;
  struct fk_spill_buf *spill_buf;
  int        nspill, spill_cap;
  char      *h_stage[2];   // pinned staging for fk_push_block
  int64_t    stage_cap;
  int        stage_idx;
  hipEvent_t stage_ev[2];
  uint8_t   *h_table;      // result table (host)
  int64_t    h_table_cap;
  void      *last_table;   // sorted table of the last resident run (HBM), for fk_make_profiles
  int64_t    last_ntab;
  bool       have_table;    // last_table holds every k-mer of the data set (or was installed by fk_set_table)
  bool       have_part_table; // last_table holds every k-mer of the records this context counted (cutoff 1)
  struct fk_block *blocks;  // push history (for the read order of the profiles)
  int64_t    nblocks, blocks_cap;
  bool       blocks_bad;    // reads also arrived through calls that carry no thread id
  int64_t   *h_prof_split; // reads of input thread t start at h_prof_split[t]
  int        h_prof_nsplit;
  const void *pf_dict_table; // the table the dictionary in FK_SLOT_PF_IDX was built from (NULL: none)
  int64_t    pf_dict_nt;
  uint8_t   *h_prof;       // profiles of fk_make_profiles (host)
  int64_t   *h_prof_off;
  int64_t    h_prof_cap, h_prof_off_cap;
  void      *push_lock;    // pthread mutex
  int64_t   *h_roff;       // exact_parts: byte offset of every pushed read (+ end), host
  int64_t    nroff, roff_cap;
  int        have_tran, tran[4];   // exact_parts: ranking from fk_train_block

  // chunked ingest (hbm_budget > 0): super-mers of the reads pushed so far, grouped by bucket
  struct fk_chunk *chunks;
  int        nchunks, chunks_cap;
  int64_t    chunk_bytes;      // split the pushed reads whenever this many bytes have accumulated (0: never)
  int64_t    chunk_ninst;      // k-mer instances of the chunks already split
  int64_t    chunk_hbm_bytes;  // chunk records currently held in HBM
  int64_t    spill_limit;      // > 0: chunks beyond this many HBM bytes go to pinned host memory
  int64_t    spilled_bytes;    // statistics: bytes spilled by the current run

  hipEvent_t ev0, ev1;
  hipEvent_t pass_ev[128];       // begin/end of each scatter launch of the current sort

  // HBM arena: one cached allocation per purpose, grown on demand and kept until fk_destroy,
  // so that a repeated workload performs no hipMalloc/hipFree inside the hot path
  // measurement aids, see fk_debug_set
  int        dbg_exact_segments;  // -1: the exact splitter keeps one thread per read (no segments inside long reads)
  int        dbg_exact_chain;     // 1..5: the exact splitter keeps that many entries of its minimizer chain (tests: the ring walk behind it)
  int        dbg_scatter_abl;     // -DFK_ABLATION builds: RX_ABL_* bits of the stream engine's scatter kernels (fk_radix.hip)
  int        dbg_radix_engine;    // 2 / 3: narrow / wide stream tiles whatever the record width; 4: stable first pass; 5: no carried digit
  int        dbg_kmer_stage;      // 1 = sort-collapse-sort k-mer stage instead of hash aggregation
  int        dbg_verbose;
  int        dbg_no_replay;       // 1: multi-pass split without entry replay
  int        dbg_smer_stage;      // 1: four hashed grouping passes for the super-mers (no LDS de-duplication)
  int        dbg_table_sort;      // 1: plain KMER_BYTES-pass table sort; >= 2: prefix length of the short sort
  int64_t    tsort_ties;          // records the last table sort had to repair
  int        dbg_aggr_variant;    // ablations of k_ag_count (wrong results), see fk_debug_set
  int        dbg_table_prefix;    // >= 2: bytes the table sort's LSD passes cover before the tie repair
  int        dbg_aggr_gshift;     // > 0: merge 2^(this - 1) bins per table fill instead of the automatic choice
  int        dbg_aggr_engine;     // 1: k_ag_count (counting sort in LDS, round 3) instead of k_ag_count2
  int        dbg_aggr_limit;      // > 0: pretend the LDS table of fk_aggr.hip takes only this many k-mers
  int64_t    dbg_slab_bytes;      // > 0: size of a slab of the chunk store instead of FK_SLAB_BYTES (8 GiB, which the driver
                                  //      clears on allocation: a harness that makes thousands of chunked contexts on
                                  //      kilobytes of reads, or dozens side by side on one GPU, asks for small ones)
  int64_t    aggr_extra_rounds;   // bins x rounds that had to be split in the last aggregation
  int        num_cus;
  int        aggr_sat;            // > 0: count that makes a k-mer's instances go to max_inst (merge: 0x8000)
  fk_result *acc_res;          // fk_rounds_*: totals over the pieces added so far
  int64_t    acc_ntab;
  double     acc_tm[4];
  int64_t    acc_ns, acc_ns_total;   // bucket streaming: super-mers counted so far / in all buckets
  int64_t    pre_hist_n;       // > 0: d_digit_hist (hash digits 0,1) and the DIG_A stream are valid for
                               // this many records (written by the expansion), see lsd_sort_stream_t
  bool       dig_lost;         // the digit stream slot was given up for memory (fk_slot): no bucket of this run may use one
  int64_t    dig2_off;         // bytes from a pointer into the splitter's digit stream to the same record's hash digit 1
  bool       dig_one_plane;    // a two-plane digit slot did not fit once: this context stays with one plane
  const uint8_t *pre_dig;      // != NULL: the stream of hash digit 0 of the pre_dig_n super-mer records the next grouping
  int64_t    pre_dig_n;        // sorts, written by the splitter beside the records (fk_split.hip)
  int64_t    ex_nweighted, ex_ndistinct;   // totals of the last expand sizing call
  void      *slot_ptr[FK_NSLOTS];
  int64_t    slot_cap[FK_NSLOTS];
};

enum { FK_SLOT_SM_A = 0, FK_SLOT_SM_B, FK_SLOT_KM_A, FK_SLOT_KM_B, FK_SLOT_EX_HEADS, FK_SLOT_EX_KMERS,
       FK_SLOT_EX_KOFF, FK_SLOT_CT_ENT, FK_SLOT_CT_OFF, FK_SLOT_CT_HIST, FK_SLOT_DIG_A, FK_SLOT_DIG_B,
       FK_SLOT_RX_TILE, FK_SLOT_RX_CHUNK, FK_SLOT_RX_SUPER, FK_SLOT_ROFF, FK_SLOT_AG_BOUNDS, FK_SLOT_TABLE, FK_SLOT_SM_G,
       FK_SLOT_RAW, FK_SLOT_FQ_INFO, FK_SLOT_FQ_PHASE, FK_SLOT_FQ_OFF, FK_SLOT_TIE_A, FK_SLOT_TIE_B,
       FK_SLOT_TIE_POS, FK_SLOT_SM_D, FK_SLOT_PF_IDX, FK_SLOT_PF_CNT, FK_SLOT_PF_ZC, FK_SLOT_PF_ZO,
       FK_SLOT_PF_ENDS, FK_SLOT_PF_LEN, FK_SLOT_PF_OFF, FK_SLOT_PF_OUT, FK_SLOT_ENT, FK_SLOT_TENT, FK_SLOT_TCNT, FK_SLOT_CBASE, FK_SLOT_PF_RID,
       FK_SLOT_PK_TIDX, FK_SLOT_PK_ASCII, FK_SLOT_SM_DIG, FK_SLOT_GATHER, FK_SLOT_XS_READS, FK_SLOT_XS_BLOCKS, FK_SLOT_XS_SCAN,
       FK_SLOT_REF_A, FK_SLOT_REF_B, FK_SLOT_COUNT_ };
static_assert(FK_SLOT_COUNT_ <= FK_NSLOTS, "more slots than fk_ctx holds");

// References to the pieces of distinct super-mers (fk_recut.hip): one 64-bit word, the key on top so that the sort takes
// its three highest bytes.  key: 22 bits of a mix of the piece's minimizer rank; idx: the super-mer among the
// bucket's de-duplicated records; off: the piece's first k-mer inside it; n: its k-mers (1 .. k - 4).
#define FK_REF_MLEN     16
#define FK_REF_KEY_BITS 22
#define FK_REF_IDX_BITS 28
#ifdef __HIPCC__
__host__ __device__
#endif
static inline u64 fk_ref_pack(u32 key, u32 idx, u32 off, u32 n)
{ return (((u64) key << 42) | ((u64) idx << 14) | ((u64) off << 7) | (u64) n); }
#ifdef __HIPCC__
__host__ __device__
#endif
static inline u32 fk_ref_n(u64 r)   { return ((u32) r & 127u); }
#ifdef __HIPCC__
__host__ __device__
#endif
static inline u32 fk_ref_off(u64 r) { return ((u32) (r >> 7) & 127u); }
#ifdef __HIPCC__
__host__ __device__
#endif
static inline u32 fk_ref_idx(u64 r) { return ((u32) (r >> 14) & ((1u << FK_REF_IDX_BITS) - 1u)); }
#ifdef __HIPCC__
__host__ __device__
#endif
static inline u32 fk_ref_key(u64 r) { return ((u32) (r >> 42)); }

// returns a device buffer of at least nbytes for the given purpose (NULL + error set on failure)
void *fk_slot(fk_ctx *ctx, int slot, int64_t nbytes);
// the splitter's digit stream for `cap` records: two planes (hash digits 0 and 1; ctx->dig2_off = distance between them)
// when the memory is there, one (dig2_off = 0) when only that fits, NULL when neither does -- the stream is an option
uint8_t *fkx_dig_slot(fk_ctx *ctx, int64_t cap);

void fk_set_error(fk_ctx *ctx, const char *fmt, ...);

#define FK_HIP(ctx, call)                                                             \
  do { hipError_t e_ = (call);                                                        \
       if (e_ != hipSuccess)                                                          \
         { fk_set_error(ctx, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),   \
                        __FILE__, __LINE__);                                          \
           return (e_ == hipErrorOutOfMemory ? FK_ENOMEM : FK_EHIP);                  \
         }                                                                            \
     } while (0)

#define FK_LAUNCH_CHECK(ctx)  FK_HIP(ctx, hipGetLastError())

// stage entry points implemented in the per-stage .hip files (C++ linkage, internal)
int fkx_lsd_sort(fk_ctx *ctx, int64_t nelem, void *d_src, void *d_trg, int rsize,
                 const int *bytes, int nbytes, void **result);
int fkx_group(fk_ctx *ctx, int64_t nelem, void *d_src, void *d_trg, int rsize, int key_bytes,
              int npasses, void **result);
int fkx_split(fk_ctx *ctx, const void *d_bases, int64_t nbytes, void *d_out, int64_t cap,
              int64_t *nsuper, int64_t *ninst, int64_t *bucket_counts, bool counts_known,
              void *d_pos = NULL, const fk_pkview *pk = NULL);
int fkx_split_plan(fk_ctx *ctx, const void *d_bases, int64_t nbytes, int64_t *cap, int64_t *offsets,
                   const fk_pkview *pk = NULL);
// wall clock in seconds (phase timers of verbose runs)
static inline double fk_wall(void)
{ struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ((double) ts.tv_sec + 1e-9 * (double) ts.tv_nsec);
}

// fk_ingest.hip: the chunk store (called with the push lock held where they touch the read buffers)
void  fkx_free_chunk(fk_ctx *ctx, fk_chunk *c);
void  fkx_rewind_slabs(fk_ctx *ctx);
void *fkx_slab_alloc(fk_ctx *ctx, int64_t bytes);
int   fkx_flush_join(fk_ctx *ctx);
int   fkx_flush_chunk(fk_ctx *ctx, bool async = false, bool carry = false);
int fkx_ktab_prepare(fk_ctx *ctx, int64_t ntable);          // fk_ktab_device.hip

int fkx_split_planned(fk_ctx *ctx, const void *d_bases, int64_t nbytes, void *d_out, int64_t cap,
                      const int64_t *offsets, int64_t *counts, int64_t *ninst, int b0 = 0, int b1 = -1, int mode = 0,
                      const fk_pkview *pk = NULL, uint8_t *d_dig = NULL);
int fkx_train_tran(fk_ctx *ctx, const void *d_bases, const int64_t *h_roff, int64_t train_reads,
                   int nthreads, int *tran);
int fkx_train_scheme(fk_ctx *ctx, const void *d_bases, const int64_t *d_roff, int64_t train, const int *tran,
                     int nparts_req);
int fkx_split_exact(fk_ctx *ctx, const void *d_bases, const int64_t *d_roff, int64_t nreads,
                    const int *tran, void **d_out, int64_t *nsuper, int64_t *ninst, int64_t *bucket_counts,
                    int64_t *bucket_offs);
int fkx_first_byte_census(fk_ctx *ctx, const void *d_recs, int64_t n, int rsize, int64_t *census);
int fkx_split_fast(fk_ctx *ctx, const void *d_bases, int64_t nbytes, void **d_out, int64_t *nsuper,
                   int64_t *ninst, int64_t *bucket_counts, int64_t *bucket_offsets, const fk_pkview *pk = NULL,
                   uint8_t **d_dig = NULL);
int fkx_expand(fk_ctx *ctx, const void *d_smers, int64_t nsuper, void *d_out, int64_t cap,
               int64_t *nweighted, int64_t *ndistinct, int64_t *overflow, bool reuse_counts = false,
               bool hash_stream = false, bool dedup = false);
// fk_recut.hip: the k-mer stage without W-sized grouping passes (references to minimizer domains, round 6)
bool fkx_recut_applies(const fk_ctx *ctx, int64_t nsx);
int fkx_recut(fk_ctx *ctx, const void *d_dd, int64_t nsx, u64 **d_refs, int64_t *nref);
int fkx_ref_count(fk_ctx *ctx, const u64 *d_refs, int64_t nref, u32 *d_tile_kmers);
int fkx_ref_bounds(fk_ctx *ctx, const u64 *d_refs, int64_t nref, const u64 *d_koff, int64_t W, int target,
                   u64 **d_bounds, int64_t *nfills);
int fkx_expand_refs(fk_ctx *ctx, const void *d_dd, int64_t nsx, const u64 *d_refs, int64_t nref, void *d_out, int64_t cap,
                    int64_t *nweighted, int64_t *overflow, const u64 **d_koff);
int fkx_aggregate_fills(fk_ctx *ctx, const void *d_grouped, int64_t n, int cutoff, int64_t *hist, int64_t *max_inst,
                        int64_t *ndistinct, void *d_table, int64_t cap, int64_t *ntable, const u64 *d_bounds, int64_t nfills);
int fkx_count(fk_ctx *ctx, const void *d_kmers, int64_t nweighted, int cutoff, int sorted_bytes,
              int64_t *hist, int64_t *max_inst, int64_t *ndistinct,
              void *d_table, int64_t cap, int64_t *ntable);
int fkx_collapse(fk_ctx *ctx, const void *d_kmers, int64_t n, void *d_out, int64_t cap,
                 int64_t *nout, int64_t *overflow);
// Copies between device memory and PAGEABLE host memory (a caller's buffer, malloc'ed results, a std::vector, a stack
// variable).  Never hipMemcpyAsync: the stream is waited for, then a blocking hipMemcpy moves the bytes -- complete in
// both memories when the call returns.  A conservative rule kept from round 5's investigation of corrupted host data:
// the asynchronous copies were suspected first and probed (138,624 of them, none wrong); the confirmed cause was the
// runtime's write into a stream object it had freed (next comment), hence the stream pool.  Pinned buffers -- h_scratch,
// the result table, the staging -- take asynchronous copies as before.
int fkx_d2h_pageable(fk_ctx *ctx, hipStream_t s, void *dst, const void *d_src, size_t nbytes);
int fkx_h2d_pageable(fk_ctx *ctx, hipStream_t s, void *d_dst, const void *src, size_t nbytes);
// Streams are never destroyed: a context (or shard) takes its non-blocking streams from a process-wide pool and gives
// them back, idle, when it goes.  Round 5: with the HIP runtime PyTorch-ROCm bundles (the one a Python host of this
// library runs on), the 920-byte stream object that hipStreamDestroy frees was still written afterwards by the runtime's
// own bookkeeping -- a counter at +152 decremented, a 32-bit word stored -- and under load (32 processes on one GPU)
// that memory had by then been handed out again by malloc: aligned 32-bit zeros in data the caller had just built
// (tests/csrc/freeguard.c names the block and the hipStreamDestroy of fk_destroy that freed it).
int  fkx_stream_get(int device, hipStream_t *s);
void fkx_stream_put(int device, hipStream_t s);
// Events likewise (round 6): the same runtime, the same class of object -- no hipEventDestroy anywhere in the library.
// An event comes from a process-wide pool (one list per device and kind: with timing / hipEventDisableTiming) and goes
// back when it is complete (fkx_event_put asks, it does not wait: a pending one is dropped alive); a pointer that is
// NULL is skipped and the caller's copy is cleared.
int  fkx_event_get(int device, bool timing, hipEvent_t *e);
void fkx_event_put(int device, bool timing, hipEvent_t *e);
int fkx_pinned_alloc(void **out, int64_t bytes);     // large buffers: huge pages touched in parallel + hipHostRegister
int fkx_pinned_free(void *p);
int fkx_reserve_host_table(fk_ctx *ctx, int64_t bytes);      // ctx->h_table: pinned host memory for the result table
int fkx_aggregate(fk_ctx *ctx, const void *d_grouped, int64_t n, int cutoff, int64_t *hist,
                  int64_t *max_inst, int64_t *ndistinct, void *d_table, int64_t cap, int64_t *ntable);
int fkx_parse_fastq(fk_ctx *ctx, const void *d_raw, int64_t nbytes, int flags, int *phase, void *d_dst,
                    int64_t *nkept, int64_t *nreads);
int fkx_unpack_reads(fk_ctx *ctx, hipStream_t s, const void *d_codes, int64_t pos0, int64_t nbases, const int64_t *d_roff,
                     int64_t nreads, const int64_t *d_inv, int64_t ninv, void *d_dst);
int fkx_unpack_store(fk_ctx *ctx, void **d_ascii, int64_t *nbytes);     // fk_ingest.hip: the packed reads of a resident run as ASCII
int fkx_pack_fixed(fk_ctx *ctx, const void *d_bases, int64_t nreads, u32 read_len, void *d_codes);
int fkx_parse_fasta(fk_ctx *ctx, const void *d_raw, int64_t nbytes, int state, void *d_dst,
                    int64_t *nkept, int64_t *nrecs);
int fkx_dedup_supermers(fk_ctx *ctx, const void *d_grouped, int64_t n, void *d_out, int64_t cap, int64_t *nout);
int fkx_sort_table(fk_ctx *ctx, int64_t n, void *d_tab, void *d_tmp, void **result, int64_t *wfirst);
int fkx_msd_sort(fk_ctx *ctx, int64_t n, void *d_a, void *d_b, int rsize, int ksize, void **result);
int fkx_profiles(fk_ctx *ctx, const void *d_bases, int64_t nbytes, const void *d_table, int64_t nt,
                 int64_t *nreads_out, int64_t *nprof_out, void **d_data, uint64_t **d_offs);
int fkx_profile_lookup_supermers(fk_ctx *ctx, const void *d_smers, int64_t ns, const void *d_table, int64_t nt,
                                 void *d_out, int64_t cap, int64_t *ninst);
int fkx_profile_scatter(fk_ctx *ctx, const void *d_smers, const void *d_pos, int64_t ns, const void *d_in,
                        int64_t nbytes, bool reset);
int fkx_profile_encode_counts(fk_ctx *ctx, const void *d_bases, int64_t nbytes, int64_t *nreads_out,
                              int64_t *nprof_out, void **d_data, uint64_t **d_offs);
int fkx_repack_table(fk_ctx *ctx, const void *d_in, int64_t n, void *d_out);   // device stride -> KMER_WORD records
int fkx_synth(fk_ctx *ctx, uint64_t seed, uint64_t genome_len, uint32_t read_len,
              uint32_t err_ppm, uint64_t first_read, int64_t nreads, void *d_bases);

// ---------------------------------------------------------------------------------------
// device helpers

#ifdef __HIPCC__

__device__ __forceinline__ u32 fk_lane() { return (threadIdx.x & 63u); }

__device__ __forceinline__ u64 fk_lanemask_lt()
{ u32 l = fk_lane();
  return (l == 0 ? 0ull : (~0ull >> (64 - l)));
}

// 64-bit mix (a, b) of the first hbytes bytes of a record of RW dwords: all of a super-mer record;
// the KMER_BYTES key of a weighted k-mer record, so that equal k-mers with different weights meet.
// Hashed digit p of the radix engine is byte p of b (p < 4) or byte p-4 of a.
// Two words per 32x32->64 multiply (v_mad_u64_u32), the halves of the product carried into the next
// one, wyhash-fashion: 3 wide multiplies for a 12-byte record.  Every hot kernel is bound by
// instruction issue and 32-bit integer multiplies are quarter rate, so the hash is kept this short;
// nothing depends on its quality beyond evenly filled bins (identical keys always meet).
template <int RW>
__device__ __forceinline__ void fk_rec_hash(const u32 *r, int hbytes, u32 &a_out, u32 &b_out)
{ const int full = hbytes >> 2;
  const u32 last = (hbytes & 3) ? ((1u << (8 * (hbytes & 3))) - 1u) : 0u;
  u32 lo = 0x9E3779B9u ^ (u32) hbytes, hi = 0x85EBCA6Bu;
#pragma unroll
  for (int w = 0; w < RW; w += 2)
    { const u32 x0 = (w < full) ? r[w] : (w == full) ? (r[w] & last) : 0u;
      const u32 x1 = (w + 1 >= RW) ? 0x27D4EB2Fu
                   : (w + 1 < full) ? r[w + 1 < RW ? w + 1 : 0]
                   : (w + 1 == full) ? (r[w + 1 < RW ? w + 1 : 0] & last) : 0u;
      const u64 p = (u64) (x0 ^ lo ^ 0xA0761D65u) * (u64) (x1 ^ hi ^ 0xE7037ED1u);
      lo = (u32) p;
      hi = (u32) (p >> 32);
    }
  const u64 p = (u64) (lo ^ 0x8EBC6AF1u) * (u64) (hi ^ 0x589965CDu);
  u32 a = (u32) p;
  b_out = (u32) (p >> 32) ^ a;
  a ^= a >> 15;
  a_out = a ^ (u32) (p >> 47);
}

// exclusive scan of one value per thread over a 256-thread block; returns exclusive prefix,
// *total gets the block sum.  tmp: 8 u32 of LDS.
template <typename T>
__device__ __forceinline__ T fk_block_exscan_256(T v, T *tmp, T *total)
{ const u32 lane = fk_lane();
  const u32 wave = threadIdx.x >> 6;
  T x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1)
    { T y = __shfl_up(x, o, 64);
      if ((int) lane >= o) x += y;
    }
  if (lane == 63) tmp[wave] = x;
  __syncthreads();
  T base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 4; w++)
    { T t = tmp[w];
      if ((u32) w < wave) base += t;
      tot += t;
    }
  __syncthreads();
  *total = tot;
  return (base + x - v);
}

// Stage `ndw` dwords from 16-byte aligned global memory into 16-byte aligned LDS with a 256-thread
// block.  NV = ceil(max ndw / 1024).  Every 16-byte load is issued before the first LDS write (a
// plain copy loop with a run-time trip count compiles to load / wait / store per iteration, i.e.
// one HBM latency per 16 bytes and thread).  SWAP: byte-swap every dword on the way.
template <int NV, bool SWAP>
__device__ __forceinline__ void fk_stage16(u32 *lds, const u32 *g, int ndw)
{ const uint4 *g4 = (const uint4 *) g;
  uint4       *l4 = (uint4 *) lds;
  const int    n4 = ndw >> 2;
  if (n4 > 0)
    { uint4 v[NV];
#pragma unroll
      for (int k = 0; k < NV; k++)
        { const int i = (int) threadIdx.x + k * 256;
          v[k] = g4[i < n4 ? i : n4 - 1];
        }
#pragma unroll
      for (int k = 0; k < NV; k++)
        { const int i = (int) threadIdx.x + k * 256;
          if (i < n4)
            { uint4 x = v[k];
              if (SWAP)
                { x.x = __builtin_bswap32(x.x); x.y = __builtin_bswap32(x.y);
                  x.z = __builtin_bswap32(x.z); x.w = __builtin_bswap32(x.w);
                }
              l4[i] = x;
            }
        }
    }
  for (int i = (n4 << 2) + (int) threadIdx.x; i < ndw; i += 256)
    lds[i] = SWAP ? __builtin_bswap32(g[i]) : g[i];
}

// exclusive scan of u32 per-tile counts into u64 offsets (single workgroup; one copy per TU).
// Each thread owns 16 consecutive counts per round, so a round covers 4096 tiles.
static __global__ __launch_bounds__(256) void k_exscan_tiles(const u32 *__restrict__ in, int64_t n,
                                                             u64 *__restrict__ out, u64 *__restrict__ total)
{ __shared__ u64 tmp[8];
  u64 carry = 0;
  for (int64_t b = 0; b < n; b += 4096)
    { const int64_t i0 = b + (int64_t) threadIdx.x * 16;
      u32 v[16];
      if (i0 + 16 <= n)
        {
#pragma unroll
          for (int k = 0; k < 4; k++)
            { const uint4 x = *(const uint4 *) (in + i0 + 4 * k);
              v[4 * k] = x.x; v[4 * k + 1] = x.y; v[4 * k + 2] = x.z; v[4 * k + 3] = x.w;
            }
        }
      else
        {
#pragma unroll
          for (int k = 0; k < 16; k++)
            v[k] = (i0 + k < n) ? in[i0 + k] : 0u;
        }
      u64 mine = 0;
#pragma unroll
      for (int k = 0; k < 16; k++)
        mine += v[k];
      u64 tot;
      u64 run = carry + fk_block_exscan_256<u64>(mine, tmp, &tot);
#pragma unroll
      for (int k = 0; k < 16; k++)
        { if (i0 + k < n)
            out[i0 + k] = run;
          run += v[k];
        }
      carry += tot;
    }
  if (threadIdx.x == 0)
    *total = carry;
}

#endif
