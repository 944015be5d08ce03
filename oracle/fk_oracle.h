/* fk_oracle.h -- CPU restatement of FastK's split / sort / count hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (fastk_amd/, include/) may include,
 * link or call this; it is used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg as the checker.  It is a restatement written from the algorithm description in
 * SURVEY.md and from reading the reference (file:line citations into /root/reference on every
 * function), not a copy.  Parity status: PINNED -- tests/test_oracle_vs_reference.py checks
 * it byte-for-byte against the reference FastK built by oracle/Makefile into oracle/_ref
 * (when /root/reference is present) and against the committed fixtures in tests/golden/
 * that were captured from that build.
 */
#ifndef FK_ORACLE_H
#define FK_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct
  { int kmer;        /* K                                             FastK.c:232  */
    int min_len;     /* PAD_LEN = 5 + PAD, minimizer length           split.c:56,617 */
    int max_super;   /* MAX_SUPER = K - (PAD_LEN-1)                   split.c:628  */
    int smer;        /* SMER = MAX_SUPER + K - 1 bases                FastK.c:454  */
    int slen_bits;   /*                                               FastK.c:456-458 */
    int slen_bytes;
    int kmer_bytes;  /* (2K+7)/8                                      FastK.c:417  */
    int smer_bytes;  /* (2*SMER+7)/8                                  FastK.c:462  */
    int smer_word;   /* SMER_BYTES + SLEN_BYTES                       FastK.c:463  */
    int kmer_word;   /* KMER_BYTES + 2                                FastK.c:464  */
    int tran[4];     /* rank of a,c,g,t used for minimizer order      split.c:529-575 */
  } orc_params;

typedef struct
  { int64_t  hist[0x8000];   /* hist[c], c in 1..0x7fff; [0] unused    count.c:1543-1553 */
    int64_t  max_inst;       /* instances of saturated k-mers                             */
    int64_t  ninst;          /* valid k-mer instances (I)                                 */
    int64_t  nsuper;         /* super-mers (S)                                            */
    int64_t  ndistinct_super;/* distinct super-mers (D)                                   */
    int64_t  nweighted;      /* weighted k-mers (W)                                       */
    int64_t  ndistinct;      /* distinct k-mers (U)                                       */
    int64_t  ntable;         /* table entries with count >= cutoff                        */
    uint8_t *table;          /* ntable records of kmer_word bytes: [KMER_BYTES][u16 count] */
    int64_t  wfirst[256];    /* weighted k-mers per canonical first byte (Kparts/KMER_WORD) */
  } orc_result;

/* the bucket scheme of Determine_Scheme (split.c:617-766): part[] = Min_Part, the prefix trie over the
   (5+pad)-base minimizers: part[x] < 0 => children at -part[x]+base, else the bucket of the leaf */
typedef struct
  { int  pad, states, nparts;
    int *part;
  } orc_scheme;

/* widths for k with PAD extra minimizer bases (FastK.c:417,446-468; split.c:617-628) */
void orc_params_init(orc_params *P, int kmer, int pad);

/* 1: the following orc_distribute_block / orc_fastk / orc_fastk_parts calls restate a run with -p (split.c:1245: the
   super-mers are not turned to their minimizer's strand); counts are the same, the hidden .ktab part boundaries are not */
void orc_set_profile_mode(int on);

/* base-frequency ranking -> tran[] (split.c:95-112,529-575) over the training block; depends on
   the thread count because the reference counts thread 0's read stripe twice (split.c:536-539) */
void orc_train_tran(orc_params *P, const char *bases, const int64_t *boff, int64_t nreads,
                    int nthreads);

/* Distribute one block of 0-terminated reads into fixed-width super-mer records
   (split.c:1016-1393 + count.c:165-313 layout).  bases/boff as in DATA_BLOCK
   (FastK.h:87-98): read i is bases[boff[i] .. boff[i+1]-1) followed by a 0 byte.
   Appends to *out (realloc'd), nout/cap in records.  Returns k-mer instances emitted. */
int64_t orc_distribute_block(const orc_params *P, const char *bases, const int64_t *boff,
                             int64_t nreads, int bc_prefix,
                             uint8_t **out, int64_t *nout, int64_t *cap);

/* In-place MSD byte radix sort of n records of rsize bytes on key bytes [0,ksize)
   (MSDsort.c:129-261 radix_sort + :105-127 shell_sort, restated) */
void orc_msd_sort(uint8_t *array, int64_t n, int rsize, int ksize);

/* Stable LSD byte radix on the -1 terminated byte list (LSDsort.c:115-271).
   Returns whichever of src/trg holds the result. */
void *orc_lsd_sort(int64_t n, void *src, void *trg, int rsize, const int *bytes);

/* Expand sorted super-mers into weighted canonical k-mer records (count.c:339-542).
   Returns number of records W written to *out (malloc'd); *overflow as count.c:455-458;
   *ndistinct = number of distinct super-mers. */
int64_t orc_kmer_list(const orc_params *P, const uint8_t *smers, int64_t nsmers,
                      uint8_t **out, int64_t *overflow, int64_t *ndistinct);

/* Collapse sorted weighted k-mers: histogram + table (MSDsort.c:491-509, count.c:564-616). */
void orc_count_sorted(const orc_params *P, const uint8_t *kmers, int64_t nk, int cutoff,
                      orc_result *R);

/* Whole path on one block set: distribute -> sort -> expand -> sort -> count. */
int orc_fastk(const orc_params *P, const char *bases, const int64_t *boff, int64_t nreads,
              int bc_prefix, int cutoff, orc_result *R);

/* Determine_Scheme on the reads of the first block (P->tran trained): census by the trainer's own super-mer rule
   (split.c:116-270), refine_tree (split.c:437-472), assign_pieces with drand48 from its default seed
   (split.c:289-381).  nparts: the buckets asked for (FastK.c:429); S->nparts what the scheme settles on. */
int orc_scheme_train(const orc_params *P, const char *bases, const int64_t *boff, int64_t nreads,
                     int bc_prefix, int nparts, orc_scheme *S);

/* orc_fastk bucket by bucket under a scheme (count.c:1202): same histogram and table, wfirst = bucket 0's census */
int orc_fastk_parts(const orc_params *P, const orc_scheme *S, const char *bases, const int64_t *boff,
                    int64_t nreads, int bc_prefix, int cutoff, orc_result *R);

/* Independent brute-force definition: sort every canonical k-mer instance directly. */
int orc_brute(int kmer, const char *bases, const int64_t *boff, int64_t nreads,
              int bc_prefix, int cutoff, orc_result *R);

void orc_result_free(orc_result *R);

/* .hist bytes (count.c:1893-1910): returns 12+16+8*0x7fff = 262164 bytes in buf */
int64_t orc_hist_bytes(int kmer, const orc_result *R, uint8_t *buf);

/* Part boundaries as first-byte values (MSDsort.c:330-352 applied to wfirst, count.c:1560-1565) */
void orc_table_split(const orc_params *P, const orc_result *R, int nthreads, int *split);

/* IDX_BYTES rule (count.c:1620-1626) */
int orc_idx_bytes(int kmer, int64_t ntable);

/* Write <dir>/<root>.hist and, if cutoff>0, <dir>/<root>.ktab + <dir>/.<root>.ktab.<1..T>
   (table.c:162-342,485-498).  split==NULL -> orc_table_split. */
int orc_write_outputs(const orc_params *P, const orc_result *R, int cutoff, int nthreads,
                      const int *split, const char *dir, const char *root);

/* Synthetic reads (include/fk_synth.h) as a DATA_BLOCK-style buffer: returns malloc'd bases,
   fills boff[0..nreads] (caller allocates nreads+1). */
char *orc_synth_block(uint64_t seed, uint64_t genome_len, uint32_t read_len, uint32_t err_ppm,
                      uint64_t first_read, int64_t nreads, int64_t *boff);

/* Load FASTA/FASTQ (plain text) with the reference's line rules (io.c:678-734) into a
   DATA_BLOCK-style buffer.  Returns malloc'd bases; *boff malloc'd with *nreads+1 entries. */
char *orc_load_fastx(const char *path, int64_t **boff, int64_t *nreads);

/* Decoder of one profile part (libfastk.c:1657-1780, Fetch_Profile): n profiles whose END offsets in
   data are ends[0..n).  Writes, per read, int32 length + that many u16 counts (little-endian) -- the byte
   stream orc.profiles_digest hashes -- and returns the bytes that stream takes; nothing is written when
   out is NULL or cap is too small.  Returns -1 on a profile that ends inside a two-byte code. */
int64_t orc_profile_decode_stream(const uint8_t *data, const int64_t *ends, int64_t n, uint8_t *out, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif
