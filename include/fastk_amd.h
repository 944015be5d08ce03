/* fastk_amd.h -- C-ABI of the MI355X (gfx950) k-mer counting engine.
 *
 * Drop-in boundary for ONE path of FastK: split -> super-mer sort -> weighted k-mer list ->
 * k-mer sort -> count -> .hist / .ktab payloads.  Every entry point below replaces a stage
 * interface of the reference (file:line into thegenemyers/FASTK) and is what a maintainer
 * would bind from FastK's C host code.  Plain pointers and sizes only;
 * all functions return 0 on success or a negative FK_E* code, and fk_last_error() gives
 * the text the host prints before Clean_Exit(1) (FastK.c:181-221 convention).
 *
 * Pointers named d_* are device (HBM) pointers; everything else is host memory.
 * All byte layouts are the reference's (little-endian host, count.c:179-185):
 *   super-mer record  [SMER_BYTES 2-bit bases, 4/byte MSB first, zero padded][SLEN_BYTES n-1]
 *                     (count.c:226-251), device stride fk_widths.smer_stride (rounded to 4)
 *   k-mer record      [KMER_BYTES canonical 2-bit bases][pad][uint16 weight/count]
 *                     (count.c:497-512), device stride fk_widths.kmer_stride (rounded to 4)
 *   table entry       [KMER_BYTES][uint16 count], TMER_WORD bytes, host side (count.c:564-616)
 */
#ifndef FASTK_AMD_H
#define FASTK_AMD_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FK_OK            0
#define FK_EINVAL       -1   /* bad argument                                   */
#define FK_ENOMEM       -2   /* host or HBM allocation failed                  */
#define FK_EHIP         -3   /* a HIP runtime call or kernel failed            */
#define FK_EUNSUPPORTED -4   /* legal in FastK but not built yet (see DESIGN)  */
#define FK_ENODEVICE    -5   /* no gfx950 device visible                       */
#define FK_ESTATE       -6   /* call order violated                            */

#define FK_HIST_BINS  0x8000 /* count.c:1205 counts[0x8000]                    */
#define FK_MIN_RANKS  16384  /* rank values of the 7-mer minimizers (fk_bucket_census)  */

typedef struct fk_ctx fk_ctx;

/* Record widths implied by k (FastK.c:417,446-468 with PAD_LEN = 5: MAX_SUPER = k-4). */
typedef struct
  { int kmer;
    int min_len;       /* minimizer length (5, split.c:56)                    */
    int max_super;     /* MAX_SUPER                                           */
    int smer_bytes;    /* SMER_BYTES                                          */
    int slen_bytes;    /* SLEN_BYTES                                          */
    int smer_word;     /* SMER_WORD  (reference record width, bytes)          */
    int kmer_bytes;    /* KMER_BYTES                                          */
    int kmer_word;     /* KMER_WORD = TMER_WORD (reference width, bytes)      */
    int smer_stride;   /* device stride of a super-mer record (multiple of 4) */
    int kmer_stride;   /* device stride of a k-mer record (multiple of 4)     */
  } fk_widths;

int fk_get_widths(int kmer, fk_widths *w);

/* Options that exist on FastK's command line (FastK.c:34-37,250-319). */
typedef struct
  { int     kmer;          /* -k (default 40); 8 <= kmer <= 64, fk_create rejects anything else   */
    int     table_cutoff;  /* -t<n>: 0 = no table, else keep k-mers with count >= n      */
    int     nthreads;      /* -T: number of .ktab parts / first-byte ranges in outputs   */
    int     bc_prefix;     /* -bc<n>: ignore this many leading bases of every read       */
    int     device;        /* HIP device ordinal for this process (one process per GPU)  */
    int     nbuckets;      /* minimizer buckets, <= 256 (the role of NPARTS, split.c:617-766).  Equal
                              k-mers share a bucket.  fk_finish / fk_count_device_reads count the
                              buckets one after the other, so only one bucket's weighted k-mers
                              are in HBM at a time; the sharded calls (fk_split_plan ... +
                              fk_count_device_supermers) hand bucket b to rank b                 */
    int64_t hbm_budget;    /* > 0: bytes of HBM the run should fit in -- fk_push_block/_device then
                              split the reads into super-mers every hbm_budget/32 bytes and drop
                              them, so the ASCII reads are never resident as a whole (choose
                              nbuckets so that a bucket's working set fits); super-mer records
                              beyond hbm_budget/2 are kept in pinned host memory until their bucket
                              is counted.  0: everything stays resident (fastest)                 */
    int     exact_parts;   /* 1: replay the reference's own super-mer rule (split.c:1016-1393)
                              so that the hidden .ktab part files get the reference's first-byte
                              boundaries (Table_Split, count.c:1560-1565) -- byte-identical files
                              whenever the reference would use one bucket (k-mers fit -M);
                              needs fk_push_block (read offsets); the split stage ~2x the default one's.
                              2: the same for a run that will make profiles (-p): the reference then keeps
                              its super-mers on the read's strand (split.c:1245), a super-mer and its reverse
                              complement stay two records, and the parts are cut by THAT list's census       */
    int     split_passes;  /* fk_count_device_reads on reads that stay resident, nbuckets > 1: split the
                              reads this many times, each pass emitting the super-mers of one group of
                              buckets only, so that 1/split_passes of the super-mer records are in HBM
                              at a time (the reads are re-read instead of spilling records: the role
                              of NPARTS when reads + records exceed memory, split.c:617-766).
                              0 = automatic: 1, or with hbm_budget > 0 as many as it takes for one
                              pass's records to fit hbm_budget / 2                                   */
  } fk_params;

void fk_default_params(fk_params *p);

/* Context: owns the HIP stream, staging buffers and all HBM arenas.
   Replaces the globals of FastK.h:34-83 plus the SORT_PATH temp files (split.c:1454). */
int         fk_create(const fk_params *p, fk_ctx **ctx);
void        fk_destroy(fk_ctx *ctx);
/* exact_parts runs only.  sort_memory: the reference's -M in bytes (12e9 by default there, -M<int> x 1e9,
   FastK.c:235,291); input_ratio: whole input over first block in file bytes where the caller knows it (io.c:528,749;
   0: by bases).  The run then cuts the input into the NPARTS buckets the reference would use (FastK.c:417-429) under
   the reference's own scheme -- the padded-minimizer trie of Determine_Scheme and the drand48 deal of assign_pieces
   (split.c:289-381,437-472,617-766) -- so that every hidden .ktab part file is cut where the reference cuts it also
   when NPARTS > 1.  Up to 32 buckets and 15-base minimizers.  Not called: one bucket. */
int         fk_set_sort_memory(fk_ctx *ctx, int64_t sort_memory, double input_ratio);
/* Returns the context's device memory and pinned staging buffers while the results of the last fk_finish (host
   memory; with keep_table != 0 also the sorted table in HBM that fk_write_ktab_device reads) stay valid; thread-safe against readers of those results, so a driver can run it beside its file
   writers.  The reference has no counterpart: its buffers go back with free() at once (count.c:1870-1890). */
int         fk_release_device(fk_ctx *ctx, int keep_table);
const char *fk_last_error(const fk_ctx *ctx);   /* ctx may be NULL: last global error */

/* Use an externally owned HIP stream (e.g. torch's current stream) for all launches. */
int fk_set_stream(fk_ctx *ctx, void *hip_stream);
int fk_synchronize(fk_ctx *ctx);

/* ---- whole-path streaming interface ------------------------------------------------------
 * fk_push_block replaces  void Distribute_Block(DATA_BLOCK *block, int tid)  (FastK.h:123,
 * split.c:1016): same data as DATA_BLOCK (FastK.h:87-98): nreads 0-terminated reads
 * concatenated in bases, read i at bases+boff[i], boff[nreads] = total bytes; rem>0 means the
 * last read continues in the next block with a K-1 base overlap (io.c:557-570).
 * Thread-safe for distinct tid (split.c:1007-1014). */
int fk_push_block(fk_ctx *ctx, const char *bases, const int32_t *boff, int nreads, int rem,
                  int tid);

/* exact_parts only: hand over the training block the reference derives its base ranking from
   (what Determine_Scheme(DATA_BLOCK *) receives, split.c:491-575).  Optional: without it the
   ranking is taken from the first reads pushed, which is the same set when blocks are pushed in
   file order by one thread. */
int fk_train_block(fk_ctx *ctx, const char *bases, const int32_t *boff, int nreads);

/* Drop everything pushed so far (a new data set follows); arenas and buffers are kept. */
int fk_reset(fk_ctx *ctx);

/* Same, for reads already resident in HBM (any byte that is not acgtACGT separates reads). */
int fk_push_device(fk_ctx *ctx, const void *d_bases, int64_t nbytes);

typedef struct
  { int64_t  hist[FK_HIST_BINS]; /* hist[c] = # distinct k-mers with count c (c>=1), count.c:1543-1553 */
    int64_t  max_inst;           /* instances of k-mers with count >= 0x7fff  (.hist ihighcnt)        */
    int64_t  ninst;              /* valid k-mer instances                 (split.c:1638 "Sum")        */
    int64_t  nsuper;             /* super-mers                                                         */
    int64_t  ndistinct_super;    /* distinct super-mers                                                */
    int64_t  nweighted;          /* weighted k-mers            (count.c:1823 "wgt'd k-mers")           */
    int64_t  ndistinct;          /* distinct k-mers                                                    */
    int64_t  ntable;             /* table entries (count >= table_cutoff)                              */
    const uint8_t *table;        /* host, ntable entries of kmer_word bytes, sorted; owned by ctx      */
    int64_t  wfirst[256];        /* sorted k-mer records per first byte (role of Kparts, count.c:1527) */
    double   ms_split, ms_sort_super, ms_expand, ms_sort_kmer, ms_count, ms_total;  /* device time   */
    int      passes_super, passes_kmer;   /* radix digit passes executed by the two sorts              */
    double   ms_pass_super, ms_pass_kmer; /* summed duration of all kernels of those passes            */
    double   ms_scatter_super, ms_scatter_kmer; /* summed duration of the scatter kernels alone        */
    int64_t  ncollapsed;                  /* records left after collapsing grouped weighted k-mers     */
    int      passes_final;                /* digit passes of the final KMER_BYTES sort over ncollapsed */
    double   ms_pass_final;
    int64_t  launches_super, launches_kmer; /* scatter launches behind ms_scatter_super / _kmer (with
                                              bucket streaming: passes x buckets)                      */
    int      split_passes;                /* split passes over the reads (fk_params.split_passes)      */
    int      replay_passes;               /* of those, passes that rebuilt records from recorded entries
                                             instead of recomputing the minimizers                     */
    int      buckets_counted;             /* non-empty minimizer buckets counted one after the other   */
    int64_t  spilled_bytes;               /* super-mer records that went through host memory           */
    double   ms_table_sort;               /* device time of the table sort (all of its kernels)        */
    int64_t  nrefs;                       /* round 6: references to pieces of distinct super-mers that were sorted
                                             instead of the weighted k-mers (0: the k-mers themselves were grouped);
                                             passes_kmer / ms_pass_kmer / ms_scatter_kmer then describe THAT sort   */
    double   ms_scatter_final;            /* summed duration of the scatter kernels of the table sort's digit passes */
  } fk_result;

/* Replaces Sorting() + the merge of Merge_Tables() (count.c:1202, table.c:346): runs the
   device pipeline over everything pushed so far.  table memory stays valid until
   fk_destroy or the next fk_finish. */
int fk_finish(fk_ctx *ctx, fk_result *res);
/* fk_finish without the host copy of the table: res->table is NULL, res->ntable set, the sorted table stays in HBM for
   fk_write_ktab_device (and fk_make_profiles).  What a driver that only writes files wants: the table (36 GB for a
   50x human-size read set) then never exists in host memory. */
int fk_finish_device(fk_ctx *ctx, fk_result *res);

/* The same pipeline on reads that are already resident in HBM and stay owned by the caller
   (16-byte aligned; any byte that is not acgtACGT separates reads).  fetch_table = 0 leaves
   the table in HBM and only reports ntable (bench path: nothing crosses PCIe but counters). */
int fk_count_device_reads(fk_ctx *ctx, const void *d_bases, int64_t nbytes, int fetch_table,
                          fk_result *res);

/* fk_count_device_reads for reads that are resident in two bits per base and stay owned by the caller: d_codes =
   nbases positions, 16 per little-endian dword in the byte order of fk_push_packed (4-byte aligned, readable up to
   the dword that holds the last position); read r = positions d_roff[r] .. d_roff[r+1]-1 (d_roff[0] = 0,
   d_roff[nreads] = nbases, no terminators); d_inv = ninv pairs (first position, length), sorted and disjoint, of
   positions that belong to no k-mer (N stretches, padding).  All device memory.  With 37.5 GB instead of 150 GB of
   reads all super-mers of a 50x human-size set fit beside them: one split pass, no replay. */
int fk_count_device_packed(fk_ctx *ctx, const void *d_codes, int64_t nbases, const int64_t *d_roff, int64_t nreads,
                           const int64_t *d_inv, int64_t ninv, int fetch_table, fk_result *res);

/* Sort + expand + sort + count over super-mer records already in HBM: what a rank runs on the
   records it owns after the bucket exchange (the .T file shuffle of split.c:1263 <-> count.c:1347).
   d_smers (nsuper records of smer_stride bytes) is clobbered. */
int fk_count_device_supermers(fk_ctx *ctx, void *d_smers, int64_t nsuper, int fetch_table,
                              fk_result *res);

/* .hist / .ktab writers with the reference encodings (count.c:1893-1910, table.c:162-342,
   485-498).  Part boundaries follow the reference's rule (MSDsort.c:330-352 on wfirst). */
int fk_write_hist(const fk_result *res, int kmer, const char *path);
int fk_write_ktab(const fk_result *res, int kmer, int table_cutoff, int nthreads,
                  const char *dir, const char *root);
/* The same files straight from the table fk_finish_device left in HBM (Merge_Tables' output, table.c:346-533): one
   writer thread per part fetches its first-byte range in 16 MB pieces through two pinned buffers -- the next piece
   crosses PCIe while this one is stripped of its prefix bytes and written. */
int fk_write_ktab_device(fk_ctx *ctx, const fk_result *res, int nthreads, const char *dir, const char *root);

/* ---- stage interface on device buffers ---------------------------------------------------*/

/* Split: reads -> super-mer records.  Replaces Distribute_Block + supermer_list_thread
   (split.c:1016-1393, count.c:165-313).  d_out must hold cap records of smer_stride bytes;
   when nbuckets > 1 records are grouped by bucket and counts[b] receives each bucket's size
   (bucket = f(canonical minimizer), so equal k-mers share a bucket, FastK.h:3-7).
   cap == 0 only counts.  *nsuper is the total number of records needed. */
int fk_split_supermers(fk_ctx *ctx, const void *d_bases, int64_t nbytes,
                       void *d_out, int64_t cap, int64_t *nsuper, int64_t *ninst,
                       int64_t *bucket_counts);

/* Second half of fk_split_supermers for callers that sized their buffer from a cap == 0 call:
   bucket_counts[] is that call's result; only the emit kernel runs. */
int fk_split_supermers_emit(fk_ctx *ctx, const void *d_bases, int64_t nbytes, void *d_out,
                            int64_t cap, const int64_t *bucket_counts);

/* One-pass bucketed split (sharded path): fk_split_plan sizes padded per-bucket regions from a 1/32
   tile sample -- offsets[0..nbuckets] in records, *cap = offsets[nbuckets]; fk_split_planned emits
   bucket b at d_out + offsets[b] records and reports the real counts[b] and *ninst.  It returns
   FK_ESTATE when a region overflowed (very uneven input): then use the exact pair above. */
int fk_split_plan(fk_ctx *ctx, const void *d_bases, int64_t nbytes, int64_t *cap, int64_t *offsets);
int fk_split_planned(fk_ctx *ctx, const void *d_bases, int64_t nbytes, void *d_out, int64_t cap,
                     const int64_t *offsets, int64_t *counts, int64_t *ninst);

/* Stable LSD byte radix sort; same contract as
     void *LSD_Sort(int64 nelem, void *src, void *trg, int rsize, int *bytes)   (FastK.h:154,
   LSDsort.c:115): bytes[] is a -1 terminated list, least significant first; *result receives
   whichever of d_src/d_trg holds the sorted records.  rsize must be a multiple of 4. */
int fk_lsd_sort_records(fk_ctx *ctx, int64_t nelem, void *d_src, void *d_trg, int rsize,
                        const int *bytes, void **result);

/* Sort records on key bytes [0,ksize), most significant first; same result as
     Supermer_Sort / Weighted_Kmer_Sort(array,nelem,rsize,ksize,...)   (FastK.h:145-151,
   MSDsort.c:458,536) without the run-head byte trick: the sorted records end up in
   *result (d_array or d_tmp).  rsize must be a multiple of 4. */
int fk_msd_sort_records(fk_ctx *ctx, void *d_array, void *d_tmp, int64_t nelem, int rsize,
                        int ksize, void **result);

/* Bring identical records together (stable, deterministic for a given input order): four digit
   passes over a 32-bit hash of the whole record.  This is the only property of Supermer_Sort's
   output that the next stage uses (the run-length pass of count.c:421-426); records whose hashes
   collide simply stay un-merged, which cannot change any count because the weighted k-mer stage
   sums weights per k-mer.  The pipeline uses this instead of a 20-byte lexicographic sort (and, keyed on the k-mer
   bytes only, to group weighted k-mers before they are collapsed and really sorted). */
int fk_group_records(fk_ctx *ctx, int64_t nelem, void *d_src, void *d_trg, int rsize,
                     void **result);

/* Weighted k-mer list from GROUPED (or sorted) super-mers.  Replaces count_smers + kmer_list_thread
   (MSDsort.c:381-456, count.c:339-542).  Call with d_out == NULL to size: *nweighted and
   *ndistinct are set.  *overflow as count.c:455-458. */
int fk_expand_kmers(fk_ctx *ctx, const void *d_smers, int64_t nsuper, void *d_out, int64_t cap,
                    int64_t *nweighted, int64_t *ndistinct, int64_t *overflow);

/* Count SORTED weighted k-mers.  Replaces hist_kmers + table_write_thread
   (MSDsort.c:491-509, count.c:564-616).  hist (host, FK_HIST_BINS) and *max_inst are
   ACCUMULATED; d_table (may be NULL when cutoff == 0) receives *ntable records of
   kmer_stride bytes [KMER_BYTES][pad][u16 count] with count >= cutoff, in sorted order. */
int fk_count_kmers(fk_ctx *ctx, const void *d_kmers, int64_t nweighted, int cutoff,
                   int64_t *hist, int64_t *max_inst, int64_t *ndistinct,
                   void *d_table, int64_t cap, int64_t *ntable);

/* As fk_count_kmers, for a list that is sorted on its first sorted_bytes key bytes only (what
   four digit passes give): the rare runs of equal prefixes that hold several k-mers are resolved
   inside LDS.  Returns FK_ESTATE, with nothing accumulated, if some run cannot be resolved there;
   the caller then finishes the sort and calls fk_count_kmers. */
int fk_count_presorted_kmers(fk_ctx *ctx, const void *d_kmers, int64_t nweighted, int cutoff,
                             int sorted_bytes, int64_t *hist, int64_t *max_inst,
                             int64_t *ndistinct, void *d_table, int64_t cap, int64_t *ntable);

/* Count UNSORTED weighted k-mers: Weighted_Kmer_Sort + hist_kmers + table_write_thread in one call
   (MSDsort.c:536-544, 491-509, count.c:564-616) without sorting the list -- two hashed digit
   passes bring equal k-mers into one of 65,536 bins, a workgroup per bin sums them in an LDS hash
   table, and only the table records (count >= cutoff) are sorted on KMER_BYTES.  d_kmers and
   d_tmp (nweighted records each) are both clobbered; *d_table (when cutoff > 0) points into one of
   them, *ntable records in k-mer order.  hist and *max_inst are ACCUMULATED.  Returns FK_ESTATE,
   with nothing accumulated and all records still present in d_kmers/d_tmp order-scrambled, in the
   (never observed) case that a bin's distinct k-mers do not fit 64 rounds of the LDS table. */
int fk_count_unsorted_kmers(fk_ctx *ctx, void *d_kmers, void *d_tmp, int64_t nweighted, int cutoff,
                            int64_t *hist, int64_t *max_inst, int64_t *ndistinct, void **d_table,
                            int64_t *ntable);

/* The reads of a DATA_BLOCK in TWO BITS PER BASE -- the north-star's first verb done by the reader threads, so that a
   quarter of the bytes cross PCIe: codes = the bases of the block's reads back to back (no terminators), four to a
   byte, first base in the two high bits (a c g t = 0 1 2 3: the .ktab encoding, README.md:977-984; what Stuff_Seq
   makes of a super-mer, split.c:864-989); rlen[i] = bases of read i (they add up to nbases); inv = ninv pairs (first
   base, length), in the same concatenated coordinates, sorted and disjoint, of stretches that hold no acgt (N, IUPAC
   codes: their code bits are ignored); rem, tid as in fk_push_block.  Host memory (pinned or not).  Equivalent to
   fk_push_block of the ASCII block, but the reads STAY packed in HBM (a quarter of the memory) and the splitter reads
   them in that form -- nothing is converted back; only exact_parts runs and fk_make_profiles restore ASCII reads on
   the device.  A run takes its reads in one form: mixing with the ASCII / text pushes returns FK_ESTATE.  Not with -bc. */
int fk_push_packed(fk_ctx *ctx, const uint8_t *codes, int64_t nbases, const int32_t *rlen, int nreads,
                   const int64_t *inv, int ninv, int rem, int tid);

/* FASTQ text, any piece of a file cut anywhere (host memory; pinned memory from fk_host_alloc
   copies fastest): the record structure is resolved on the device -- replaces the FASTQ branch of
   fast_output_thread (io.c:574-759, line rules io.c:678-734: strictly four lines per record, every
   non-newline byte of the sequence line is a base) in front of fk_push_block.  *line_phase (parser state:
   line number mod 4 and the last byte of the previous piece) must be 0 before the first piece of a file
   and is carried between calls; *nreads and *nbases (may be NULL)
   are incremented.  Not available with -bc or exact_parts (they need host-side read offsets). */
#define FK_FASTQ_HOCO 1      /* flags: homopolymer-compress the reads (-c, io.c:284-294) */
int fk_push_fastq(fk_ctx *ctx, const char *raw, int64_t nbytes, int flags, int *line_phase,
                  int64_t *nreads, int64_t *nbases);
/* The same for FASTA text (a line starting with '>' is a header, all other lines of a record are one
   read, io.c:700-734).  *state must be 2 before the first piece of a file; last != 0 with the final
   piece of a file. */
int fk_push_fasta(fk_ctx *ctx, const char *raw, int64_t nbytes, int last, int *state, int64_t *nreads,
                  int64_t *nbases);
int fk_host_alloc(int64_t nbytes, void **ptr);      /* pinned host memory */
int fk_host_free(void *ptr);

/* Bucket training, the role of the trie balancing in Determine_Scheme (split.c:617-766).
   fk_bucket_census: host-side pass over a SAMPLE of reads (any bytes, non-ACGT separates) that adds,
   per canonical minimizer rank, 4 x super-mer starts + k-mer instances to counts[FK_MIN_RANKS].
   fk_set_bucket_weights: deal the ranks to the nbuckets buckets by those weights (longest processing
   time first) instead of the default serpentine deal.  In a sharded run every process must pass the
   same counts (all-reduce the census first).  Results do not depend on the assignment. */
int fk_bucket_census(fk_ctx *ctx, const char *bases, int64_t nbytes, int64_t *counts);
int fk_set_bucket_weights(fk_ctx *ctx, const int64_t *counts);

/* Sum-merge of k-mer tables, the table/histogram part of Fastmerge (Fastmerge.c:168-457, 985-1030):
   records = the entries of all input tables in any order, KMER_WORD bytes each (host memory);
   max_inst_in = sum of the inputs' histogram "high count" fields (hist[0x8001], 0 if unknown).
   res->hist / max_inst / table (sorted, counts saturated at 0x7fff) as Fastmerge computes them. */
int fk_merge_tables(fk_ctx *ctx, const uint8_t *records, int64_t n, int64_t max_inst_in, fk_result *res);

/* fk_write_ktab with an explicit prefix-index width (1..3; 0 = FastK's rule). */
int fk_write_ktab_ex(const fk_result *res, int kmer, int table_cutoff, int nthreads, int idx_bytes,
                     const char *dir, const char *root);

/* ---- read profiles (FastK -p) --------------------------------------------------------------
 * Replaces the profile output of the reference's counting pass (count.c:868-947, the per-super-mer
 * fragments) and their stitching into read order (merge.c, Merge_Profiles): for every read, in input
 * order, the counts of its k-mers (0 where the k-mer contains a non-acgt base, capped at 32767; with
 * bc_prefix, of the read without its first bc_prefix bases, for reads pushed with fk_push_block),
 * compressed with the codec of README.md:1029-1069.  The bytes are the canonical
 * one-byte-form-whenever-possible stream; they decode (libfastk.c:1657, Fetch_Profile) to the same
 * counts as the reference's files, whose zero-run splits follow its internal work panels
 * (merge.c:65,711-716).
 *
 * fk_make_profiles runs after fk_finish / fk_count_device_reads with table_cutoff 1 -- the table
 * left in HBM is the dictionary.  d_bases NULL: the reads pushed into the context (resident runs,
 * hbm_budget 0); otherwise a caller-owned, 16-byte aligned buffer of reads of the data set just counted -- after a
 * chunked run (hbm_budget > 0), whose reads were dropped on the way, the caller passes them again
 * piece by piece (whole reads per piece) and concatenates the results.  Reads end at 0 bytes
 * (a last read without one ends at nbytes); other non-acgt bytes stay inside their read.
 * Read order: blocks pushed with distinct tid interleave arbitrarily in time; the profiles come out
 * in the data set's order -- all reads of tid 0 in push order, then tid 1, ... (io.c hands every
 * input thread a contiguous range of the input, io.c:2455-2521) -- and split[] gives those ranges,
 * which are the reference's part files when fk_write_prof is called with nparts = nsplit.  A read that
 * was pushed in pieces (blocks with rem > 0, the next block of that tid repeating K-1 bases) gets ONE
 * profile: the pieces' counts are joined on the host and encoded again.
 * data / offsets / split are host memory owned by ctx, valid until the next call or fk_destroy. */
typedef struct
  { int64_t        nreads;
    int64_t        nbytes;     /* total compressed bytes                                   */
    const uint8_t *data;       /* profile of read i = data[offsets[i] .. offsets[i+1])     */
    const int64_t *offsets;    /* nreads + 1 entries, offsets[0] = 0                       */
    int            nsplit;     /* input threads seen by fk_push_block (0: not applicable)  */
    const int64_t *split;      /* reads of thread t: split[t] .. split[t+1] (nsplit + 1)   */
  } fk_profiles;

int fk_make_profiles(fk_ctx *ctx, const void *d_bases, int64_t nbytes, fk_profiles *out);

/* Install another table as the dictionary of fk_make_profiles: relative profiles (FastK -p:<table>,
   FastK.c:270-282 + the cmer_merge path count.c:675-815 -- counts are those of the given table, 0 for
   k-mers it does not hold), and the sharded run, where every rank installs the union of all ranks'
   tables.  records: n entries of kmer_word bytes ([KMER_BYTES][uint16 count]) in any order, k-mers
   distinct, counts >= 1; they are copied to HBM.  No counting run is needed: push the reads,
   call fk_set_table, then fk_make_profiles(ctx, NULL, 0, &out). */
int fk_set_table(fk_ctx *ctx, const uint8_t *records, int64_t n);

/* Profiles in the sharded run without replicating the table (the reference carries run ordinals
 * through its sorts for the same purpose, count.c:639-1181): the rank that reads a stripe keeps, for
 * every super-mer record it sends away, the position it was cut from; the owning rank looks the
 * record's k-mers up in ITS table after counting and sends the counts back (2 bytes per k-mer
 * instance, same order as the records came); the reader scatters them to the positions and encodes.
 *   fk_split_supermers_emit_pos  fk_split_supermers_emit that also writes d_pos[record] =
 *                                (byte offset of the record's first k-mer in d_bases << 1) | flip
 *                                (flip: the record holds the reverse strand, its k-mers run backwards)
 *   fk_profile_lookup_supermers  owner: counts of the k-mers of nsuper records (device stride, e.g. a
 *                                copy of the inbox taken before counting clobbered it), record after
 *                                record, into d_counts (uint16, cap entries; cap 0 only sizes);
 *                                needs the table of a finished cutoff-1 run of this context
 *   fk_profile_scatter           reader: counts of the records it sent (same order) to their positions
 *                                in a per-position array of nbytes entries (reset != 0 clears it first)
 *   fk_profile_encode            reader: read boundaries + codec over that array -> profiles of the
 *                                reads in d_bases (as fk_make_profiles returns them) */
int fk_split_supermers_emit_pos(fk_ctx *ctx, const void *d_bases, int64_t nbytes, void *d_out, int64_t cap,
                                const int64_t *bucket_counts, void *d_pos);
int fk_profile_lookup_supermers(fk_ctx *ctx, const void *d_smers, int64_t nsuper, void *d_counts, int64_t cap,
                                int64_t *ninst);
int fk_profile_scatter(fk_ctx *ctx, const void *d_smers, const void *d_pos, int64_t nsuper, const void *d_counts,
                       int64_t nbytes, int reset);
int fk_profile_encode(fk_ctx *ctx, const void *d_bases, int64_t nbytes, fk_profiles *out);

/* <dir>/<root>.prof stub + hidden .<root>.pidx.N / .<root>.prof.N, N = 1..nparts (README.md:1010-1027);
   part t holds the reads of input thread t when nparts == p->nsplit, else the reads are divided evenly
   over the parts in input order. */
int fk_write_prof(const fk_profiles *p, int kmer, int nparts, const char *dir, const char *root);
/* The same for the hidden parts part0+1 .. part0+nhere of an nparts-part set only, from reads whose first one is read
   read_base of the whole data set (the .pidx files name it); the <root>.prof stub with stub != 0.  What a rank of a
   sharded run writes (fk_shard_write_prof). */
int fk_write_prof_range(const fk_profiles *p, int kmer, int nparts, int part0, int nhere, int64_t read_base, int stub,
                        const char *dir, const char *root);

/* ---- writing one table from several ranks ----------------------------------------------------
 * The pieces of fk_write_ktab (Merge_Tables, table.c:346-533), for the sharded run where rank r ends
 * up holding the k-mers of a contiguous first-byte range and writes the hidden part files of that
 * range itself (README.md:988: parts are ordered ranges of the table):
 *   fk_ktab_idx_bytes   prefix-index width for a table of ntable entries (count.c:1620-1626)
 *   fk_ktab_split       Table_Split's first-byte boundaries (count.c:1560-1565, MSDsort.c:330-352)
 *                       from the summed first-byte census of the weighted k-mers; split[nparts+1]
 *   fk_write_ktab_range writes .<root>.ktab.<part0+1 ..> from sorted records (kmer_word bytes each)
 *                       and adds the per-prefix entry counts to prefix_counts[256^idx_bytes]
 *   fk_write_ktab_stub  writes <root>.ktab from the prefix counts summed over all ranks
 * fk_write_ktab is these three in one process. */
int fk_ktab_idx_bytes(int kmer, int64_t ntable);
int fk_ktab_split(const int64_t *wfirst, int kmer, int nparts, int *split);
int fk_write_ktab_range(const uint8_t *records, int64_t n, int kmer, int idx_bytes, const int *split,
                        int part0, int nhere, const char *dir, const char *root, int64_t *prefix_counts);
int fk_write_ktab_stub(int kmer, int nparts, int table_cutoff, int idx_bytes, const int64_t *prefix_counts,
                       const char *dir, const char *root);

/* The records a rank owns, counted piece by piece (one piece per exchange round of the sharded run,
   so that the exchange of the next piece overlaps the counting of this one).  A piece must consist of
   whole minimizer buckets.  fk_rounds_add clobbers d_smers; fk_rounds_finish gives the result over all
   pieces (what fk_count_device_supermers gives for their union). */
int fk_rounds_begin(fk_ctx *ctx);
int fk_rounds_add(fk_ctx *ctx, void *d_smers, int64_t nsuper);
int fk_rounds_finish(fk_ctx *ctx, int fetch_table, fk_result *res);

/* ---- the sharded run from a C host: one process per GPU, RCCL over xGMI ------------
 * Replaces the reference's file traffic between its phases on a node with several GPUs: the ".T"
 * super-mer shuffle (split.c:1263 <-> count.c:1347) becomes grouped ncclSend/ncclRecv of SMER_WORD
 * records keyed by minimizer bucket, the histogram reduce (count.c:1543-1553) an all-reduce, and the
 * merge of the per-bucket tables (table.c:346-533) a second exchange keyed by the first k-mer byte after
 * which every rank writes the hidden .ktab parts of its own first-byte range.  Reads are striped over
 * the ranks by the host (io.c:2455-2521 stripes them over threads).  RCCL is loaded at run time.
 *
 *   rank 0:      fk_shard_unique_id(id)            and hands the 128 bytes to the other ranks (file, pipe)
 *   every rank:  fk_create(nbuckets = world * rounds, device = its GPU, hbm_budget 0)
 *                fk_shard_create(ctx, rank, world, id, &sh)
 *                fk_push_block / fk_push_fastq / fk_push_fasta / fk_push_device   its stripe of the reads
 *                fk_shard_count(sh, &res)          C1: bucket r*world+d -> rank d in round r, round r+1 travels
 *                                                  while round r is counted; C2: res = GLOBAL histogram,
 *                                                  totals and first-byte census, identical on all ranks
 *                                                  (res.table is NULL: each rank's share stays in HBM);
 *                                                  fails if records or k-mer instances were not conserved
 *                fk_shard_gather(sh, &res, nparts, &tab, &n)   C3 alone: this rank's range of the table, host memory
 *                fk_shard_write(sh, &res, nparts, dir, root)   C3 + files, byte for byte those of a one-GPU
 *                                                  run with -T nparts (nparts a multiple of world)
 *                fk_shard_destroy(sh); fk_destroy(ctx)
 */
typedef struct fk_shard fk_shard;
int  fk_shard_unique_id(char *id128);
int  fk_shard_create(fk_ctx *ctx, int rank, int world, const char *id128, fk_shard **sh);
int  fk_shard_count(fk_shard *sh, fk_result *res);
/* fk_shard_count over a stripe that is already resident in HBM and stays owned by the caller (16-byte aligned;
   any byte that is not acgtACGT separates reads), the sharded twin of fk_count_device_reads */
int  fk_shard_count_device(fk_shard *sh, const void *d_bases, int64_t nbytes, fk_result *res);
/* The same over a stripe resident in two bits per base (the arguments of fk_count_device_packed). */
int  fk_shard_count_device_packed(fk_shard *sh, const void *d_codes, int64_t nbases, const int64_t *d_roff, int64_t nreads,
                                  const int64_t *d_inv, int64_t ninv, fk_result *res);
/* this rank's own share of the last fk_shard_count (counts and per-kernel device times; table NULL) */
int  fk_shard_local_result(fk_shard *sh, fk_result *res);
/* C3 alone, the final gather of the north-star ("... per-GPU sort+count, then a final gather"; the reference's
   counterpart is the heap merge of table.c:346-533): this rank's first-byte range of the whole table -- the parts
   rank*m+1 .. rank*m+m of an nparts-part .ktab, m = nparts / world -- exchanged over RCCL, ordered, in pinned host
   memory (KMER_BYTES + 2 bytes per entry; valid until the next gather / fk_shard_destroy). */
int  fk_shard_gather(fk_shard *sh, const fk_result *res, int nparts, const uint8_t **table, int64_t *nentries);
int  fk_shard_write(fk_shard *sh, const fk_result *res, int nparts, const char *dir, const char *root);
/* Profiles of a sharded run (FastK -p on several GPUs; count.c:639-1181 + merge.c:761-1006 in the reference): after
   fk_shard_count with table_cutoff 1, a piece of this rank's reads (d_bases: 0-terminated ASCII in HBM, 16-byte aligned)
   is cut into super-mers that remember their positions, these travel to the ranks that own their minimizer buckets,
   the owners look the k-mers up in their share of the table and send the counts back, and the piece is encoded; out as
   fk_make_profiles.  COLLECTIVE: every rank calls it the same number of times -- a rank that has run out of reads
   passes nbytes 0 until *active_ranks (ranks that passed reads in this call) comes back 0.
   fk_shard_write_prof: every rank's profiles are those of a contiguous range of the data set's reads, ranks in file
   order; it writes the hidden parts of its range (nparts / world each) and rank 0 the stub. */
int  fk_shard_profiles(fk_shard *sh, const void *d_bases, int64_t nbytes, fk_profiles *out, int *active_ranks);
int  fk_shard_write_prof(fk_shard *sh, const fk_profiles *p, int kmer, int nparts, const char *dir, const char *root);
/* What the last fk_shard_count (C1) and fk_shard_gather (C3) of this rank moved, and over how many ranks. */
typedef struct
  { int     comm_ranks;          /* ncclCommCount of the communicator the exchanges run on              */
    int     rounds;              /* exchange rounds of C1 (nbuckets / world)                             */
    int64_t sent_bytes;          /* C1: super-mer records sent to other ranks                            */
    int64_t recv_bytes;          /*     ... received from them                                           */
    int64_t kept_bytes;          /*     ... kept (a device copy)                                         */
    double  exchange_ms;         /*     device time of the sends / receives (they overlap the counting)  */
    int64_t gather_sent_bytes;   /* C3: table entries sent to other ranks                                */
    double  gather_exchange_ms, gather_sort_ms, gather_d2h_ms;   /* C3 by phase, wall clock             */
    int64_t table_entries_written; /* C3: entries of the whole table that passed the write cutoff (all ranks)    */
  } fk_shard_stats;
int  fk_shard_get_stats(fk_shard *sh, fk_shard_stats *st);
/* A table cutoff for the FILES only (FastK -t<n> with -p, FastK.c:491-540: the profiles look every k-mer up, the
   .ktab keeps those that occur n or more times): the context counts with table_cutoff 1, and fk_shard_gather /
   fk_shard_write drop the entries below `cutoff` from this rank's table before the second exchange (what the one-GPU
   driver and the shim do at the write, host/gpu_path.c:113-128).  The table in HBM -- the dictionary of
   fk_shard_profiles -- keeps every k-mer.  0 or 1: write what was counted.  Every rank passes the same value. */
int  fk_shard_set_write_cutoff(fk_shard *sh, int cutoff);
/* vals[0..n) summed element-wise over all ranks, in place (host memory). */
int  fk_shard_sum_i64(fk_shard *sh, int64_t *vals, int n);
void fk_shard_destroy(fk_shard *sh);

/* ---- utilities ---------------------------------------------------------------------------*/

/* Fill d_bases with synthetic reads first_read .. first_read+nreads-1 of include/fk_synth.h,
   each followed by a 0 byte (DATA_BLOCK layout): nreads*(read_len+1) bytes. */
int fk_synth_reads(fk_ctx *ctx, uint64_t seed, uint64_t genome_len, uint32_t read_len,
                   uint32_t err_ppm, uint64_t first_read, int64_t nreads, void *d_bases);

/* The two-bit form fk_push_packed takes, made on the device from nreads rows of read_len bases + terminator (acgt in
   either case; anything else becomes code 0): (nreads * read_len + 3) / 4 bytes at d_codes.  A measurement helper. */
int fk_pack_fixed_reads(fk_ctx *ctx, const void *d_bases, int64_t nreads, uint32_t read_len, void *d_codes);

/* What a plain streaming copy reaches on this device: d_src -> d_dst (nbytes each, 16-byte aligned) with a uint4 copy
   kernel (a tile per workgroup, four loads in flight per thread), best of reps runs; *gbps = (read + write) bytes per
   second / 1e9.  The ceiling the scatter passes are priced against beside the 8 TB/s peak: hipMemcpy device-to-device
   is 10 % below it on this stack.  A measurement helper. */
int fk_copy_rate(fk_ctx *ctx, void *d_dst, const void *d_src, int64_t nbytes, int reps, double *gbps);

int   fk_device_alloc(fk_ctx *ctx, int64_t nbytes, void **d_ptr);
int   fk_device_free(fk_ctx *ctx, void *d_ptr);
int   fk_copy_to_device(fk_ctx *ctx, void *d_dst, const void *src, int64_t nbytes);
int   fk_copy_to_host(fk_ctx *ctx, void *dst, const void *d_src, int64_t nbytes);

/* Per-kernel device time of the most recent sort call, for roofline accounting:
   passes executed, records, record width (reference bytes), total ms over the passes. */
typedef struct
  { int     passes;
    int64_t nelem;
    int     rsize;
    double  pass_ms_total;   /* all kernels of the digit passes (HIP events on the ctx stream)    */
    double  scatter_ms_total;/* the scatter kernels alone (k_rx_scatter), event pair per launch   */
    double  hist_ms;         /* digit histogram kernel                                         */
  } fk_sort_stats;
int fk_get_sort_stats(fk_ctx *ctx, fk_sort_stats *st);

/* Measurement and test aid: selects ablated kernel variants for profiles/ (results are invalid while a non-zero
   variant is set) and alternative routes with the SAME results for the tests ("aggr_engine", "radix_engine",
   "slab_bytes", "exact_segments" 0: the exact splitter follows whole reads, "exact_chain" 1..5: it keeps that many
   entries of its minimizer chain in registers, so that the ring walk behind it is exercised).  The product path never
   calls it. */
int fk_debug_set(fk_ctx *ctx, const char *key, int64_t value);
/* Reads back a counter of the last run ("aggr_extra_rounds": bin rounds that had to be split). */
int fk_debug_get(fk_ctx *ctx, const char *key, int64_t *value);

const char *fk_version(void);

#ifdef __cplusplus
}
#endif
#endif
