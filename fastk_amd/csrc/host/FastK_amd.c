/* FastK_amd.c -- host driver with FastK's command line over libfastk_amd.so.
 *
 *   FastK_amd [-k<int(40)>] [-t[<int(1)>]] [-p[:<table>[.ktab]]] [-c] [-bc<int>] [-v] [-x] [-N<path_name>] [-P<dir>] [-M<int>]
 *             [-T<int(4)>] [-G<int(1)>] <source>[.fa|.fasta|.fq|.fastq][.gz]|.sam|.bam ...
 *
 * Same flags, defaults, output names and encodings as the reference driver (FastK.c:34-37,
 * 250-319, 361-409): <root>.hist always, <root>.ktab + hidden .<root>.ktab.<1..T> with -t,
 * <root>.prof + hidden .<root>.pidx/.prof.<1..T> with -p (only those with -p:<table>).
 * The host side is plain C: it parses FASTA/FASTQ with the reference's line rules
 * (io.c:678-734: FASTQ strictly 4-line, FASTA possibly multi-line, every non-newline byte of a
 * sequence line is a base), cuts the input into DATA_BLOCK-shaped blocks (FastK.h:87-98; a read
 * longer than a block continues in the next one with a K-1 base overlap, io.c:557-570) and hands
 * them to fk_push_block -- the call that replaces Distribute_Block (FastK.h:123).  Everything
 * per-base and per-record runs on the GPU inside the library.
 *
 * FASTQ and FASTA text is parsed on the GPU (fk_push_fastq / fk_push_fasta) unless -bc, -x or -H
 * (extension: host parser) is given.
 *
 * -x (extension, no reference counterpart): exact part files -- replays the reference's own super-mer
 * rule so that the hidden .ktab parts are cut at the reference's first bytes (fk_params.exact_parts);
 * without it the table's canonical stream is identical but the part boundaries are our own.
 *
 * -M<int>: GB of HBM the run may use (the reference's -M is host memory for the same purpose, it
 * fixes the number of super-mer buckets, split.c:617-766): the reads are split into super-mers chunk
 * by chunk and the minimizer buckets are counted one after the other; the number of buckets is
 * derived from the input size.  Without -M everything stays resident in one bucket (fastest).
 * -G<int>: that many GPUs of this node, one process each (extension; the reference has threads, not devices):
 * the program starts -G copies of itself, every copy reads its stripe of the input (plain FASTA / FASTQ: a byte
 * range cut at record starts, io.c:2455-2521; anything else: every -G'th block of reads), the super-mers travel
 * to the GPU that owns their minimizer bucket with RCCL (fk_shard_count), and every rank writes the hidden
 * .ktab parts of its own first-byte range (fk_shard_write); -T is rounded up to a multiple of -G.  The files
 * are those of the one-GPU run with that -T.
 * SAM and BAM input follow io.c:1314-1495 (secondary / supplementary records skipped).
 * Accepted for compatibility and ignored: -P (no temporary files exist).  Rejected with a message:
 * CRAM and Dazzler inputs.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include <zlib.h>
#include <sys/time.h>
#include <fcntl.h>
#include <unistd.h>
#include <pthread.h>
#include <time.h>
#include <sys/types.h>
#include <sys/wait.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <signal.h>

#include "../../../include/fastk_amd.h"

#define BLOCK_BYTES (8 << 20)
#define BLOCK_READS 100000

static char *Prog_Name = "FastK_amd";

#include "ktab_io.h"

static double now(void)
{ struct timeval tv;
  gettimeofday(&tv,NULL);
  return (tv.tv_sec + 1e-6*tv.tv_usec);
}

static int       KMER = 40, NTHREADS = 4, DO_TABLE = 0, BC_PREFIX = 0, VERBOSE = 0, EXACT = 0, MEM_GB = 0, NGPUS = 1;
static int       HOST_PARSE = 0, COMPRESS = 0, PROFILE = 0;
static int       DEVICE_TEXT = 0;   /* FASTK_AMD_DEVICE_TEXT=1: the file text is parsed on the device even where the readers could pack it */
static char     *PRO_NAME = NULL;
static char     *OUT_NAME = NULL;
/* -G<n>: one process per GPU.  The parent starts n copies of itself (before it has touched any GPU) with
   --fk-rank=<r> --fk-idfile=<path>; RANK >= 0 marks such a copy. */
static int       RANK = -1, ROUNDS = 4;
static char     *ID_FILE = NULL;
static int64_t   BLOCK_NO = 0;     /* host-parsed input of a sharded run: block j belongs to rank j mod n */

#include "input_formats.h"

struct Feeder
  { fk_ctx  *ctx;
    char    *bases;
    int32_t *boff;
    int      nreads;
    int64_t  olen;
    int64_t  totbps, totrds;
    int64_t  striped_rds;    /* of totrds: reads of this rank's STRIPE of a file (a rank of a -G run reads a byte range of
                                a plain file; a file the host parser walks is seen whole by every rank) */
    int      lastc;
    int64_t  cap_bytes;      /* block capacity (bases incl. terminators) and reads */
    int      cap_reads;
    /* second pass of -p with -M: the blocks are not counted, their reads are looked up in the table of
       the first pass piece by piece and the profiles collected here */
    int      to_profiles;    /* 1: fk_make_profiles per block; 2: fk_shard_profiles (a rank of a -G run) */
    void    *shard;          /* to_profiles == 2: the rank's fk_shard */
    int64_t  rd_lo, rd_hi, rd_seen;   /* rd_hi > 0: only reads rd_lo .. rd_hi-1 of the data set (in file order) are
                                         kept -- the contiguous range a rank of a sharded run makes profiles of */
    void    *d_piece;        /* device buffer of cap_bytes */
    uint8_t *pdata;
    int64_t *poffs;
    int64_t  pbytes, pbytes_cap, preads, preads_cap;
    /* -x -p on one plain file: the reads are dealt to the input threads the reference would use (io.c:2420-2521: ITHREADS
       byte ranges of the file, each moved to the next record start), because its .prof parts are those ranges and its
       run numbers count per input thread (merge.c:263-267) */
    int      tid;            /* input thread of the block being filled (fk_push_block's tid) */
    int      nstarts;        /* 0: everything is thread 0's */
    int      st_file[256];   /* thread t begins in input file st_file[t] (0-based) ... */
    off_t    starts[256];    /* ... at this offset: the first byte of a record */
    int      cur_file;       /* the file being scanned */
  };

static void die(fk_ctx *ctx, const char *what)
{ fprintf(stderr,"%s: %s: %s\n",Prog_Name,what,fk_last_error(ctx));
  if (ctx != NULL)
    fk_destroy(ctx);
  exit (1);
}

static void flush_block(Feeder *f, int rem)
{ if (f->nreads == 0)
    return;
  f->boff[f->nreads] = (int32_t) f->olen;
  if (f->to_profiles)
    { fk_profiles pr;
      int64_t i;
      if (rem)
        { fprintf(stderr,"%s: -p with -M: a read is longer than %lld bases\n",Prog_Name,(long long) f->cap_bytes);
          exit (1);
        }
      if (BC_PREFIX > 0)                  /* as fk_push_block does in the counting pass */
        for (i = 0; i < f->nreads; i++)
          { int64_t j, e = f->boff[i+1]-1;
            for (j = f->boff[i]; j < e && j < f->boff[i]+BC_PREFIX; j++)
              f->bases[j] = 'N';
          }
      if (fk_copy_to_device(f->ctx,f->d_piece,f->bases,f->olen) != FK_OK)
        die(f->ctx,"fk_copy_to_device");
      if (f->to_profiles == 2)
        { int active = 0;                 /* collective: the other ranks are in this call with a block of theirs (or 0 bytes) */
          if (fk_shard_profiles((fk_shard *) f->shard,f->d_piece,f->olen,&pr,&active) != FK_OK)
            die(f->ctx,"fk_shard_profiles");
        }
      else if (fk_make_profiles(f->ctx,f->d_piece,f->olen,&pr) != FK_OK)
        die(f->ctx,"fk_make_profiles");
      if (f->pdata == NULL || f->pbytes+pr.nbytes > f->pbytes_cap)    /* (a piece of reads shorter than k has profiles of no bytes) */
        { f->pbytes_cap = (f->pbytes+pr.nbytes)*2 + (1 << 20);
          f->pdata = realloc(f->pdata,(size_t) f->pbytes_cap);
        }
      if (f->preads+pr.nreads+1 > f->preads_cap)
        { f->preads_cap = (f->preads+pr.nreads+1)*2 + 1024;
          f->poffs = realloc(f->poffs,sizeof(int64_t)*(size_t) f->preads_cap);
        }
      if (f->pdata == NULL || f->poffs == NULL)
        { fprintf(stderr,"%s: Out of memory\n",Prog_Name); exit (1); }
      if (pr.nbytes > 0)
        memcpy(f->pdata+f->pbytes,pr.data,(size_t) pr.nbytes);
      f->poffs[0] = 0;
      for (i = 0; i < pr.nreads; i++)
        f->poffs[f->preads+i+1] = f->pbytes + pr.offsets[i+1];
      f->pbytes += pr.nbytes;
      f->preads += pr.nreads;
    }
  else if (RANK >= 0 && NGPUS > 1 && (BLOCK_NO++ % NGPUS) != RANK)
    ;                               /* another rank's block (any division of the reads gives the same counts; a piece
                                       of a read cut at a block edge carries its K-1 overlap, so it is dealt like any
                                       other block -- every rank pushing it would count its k-mers NGPUS times) */
  else if (fk_push_block(f->ctx,f->bases,f->boff,f->nreads,rem,f->tid) != FK_OK)
    die(f->ctx,"fk_push_block");
  f->nreads  = 0;
  f->olen    = 0;
  f->boff[0] = 0;
}

static int KEEP_TABLE = 0;

static void *release_thread(void *arg)
{ double t0 = now();
  fk_release_device((fk_ctx *) arg,KEEP_TABLE);
  if (VERBOSE)
    fprintf(stderr,"  device memory released in %.3f s (beside the file writers)\n",now()-t0);
  return (NULL);
}

/* a range of the data set's reads only (the profile pass of a -G rank): is the read being scanned outside it? */
static inline int read_dropped(const Feeder *f)
{ return (f->rd_hi > 0 && (f->rd_seen < f->rd_lo || f->rd_seen >= f->rd_hi)); }

/* ... and has the scan passed the range's last read?  (the scanners stop there: the rest would be parsed and dropped) */
static inline int range_done(const Feeder *f)
{ return (f->rd_hi > 0 && f->rd_seen >= f->rd_hi); }

static inline void add_base(Feeder *f, int c)
{ if (read_dropped(f))                    /* not this rank's read: its bases are not kept at all -- a long one would */
    return;                               /*  otherwise fill the block and trip the "read is longer than" exit below */
  if (COMPRESS)                           /* -c: homopolymer compression, io.c:284-294,558 */
    { if (c == f->lastc)
        return;
      f->lastc = c;
    }
  if (f->olen >= f->cap_bytes-2 && f->to_profiles && f->nreads > 0)
    { /* profile pass: the complete reads of the block go first, the read being scanned moves to the front of the
         empty block (a read of more than a MB that arrives on a nearly full block is not "longer than a block") */
      const int64_t start = f->boff[f->nreads], part = f->olen-start;
      f->olen = start;
      flush_block(f,0);
      memmove(f->bases,f->bases+start,(size_t) part);
      f->olen = part;
    }
  if (f->olen >= f->cap_bytes-2)
    { /* the read is longer than a block: close it here, continue with a K-1 overlap */
      char keep[256];
      int  ov = KMER-1;
      int64_t start = f->boff[f->nreads];
      if (f->olen - start < ov)
        ov = (int) (f->olen - start);
      memcpy(keep,f->bases+f->olen-ov,ov);
      f->bases[f->olen++] = 0;
      f->nreads += 1;
      flush_block(f,1);
      f->boff[0] = 0;
      memcpy(f->bases,keep,ov);
      f->olen = ov;
    }
  f->bases[f->olen++] = (char) c;
  f->totbps += 1;
}

static inline void end_read(Feeder *f)
{ f->lastc = 0;
  if (f->rd_hi > 0)                       /* a range of the data set's reads only: the others are parsed and dropped */
    { const int64_t idx = f->rd_seen++;   /* (add_base kept none of their bases) */
      if (idx < f->rd_lo || idx >= f->rd_hi)
        return;
    }
  f->bases[f->olen++] = 0;
  f->nreads += 1;
  f->totrds += 1;
  f->boff[f->nreads] = (int32_t) f->olen;
  if (f->olen > f->cap_bytes - (1 << 20) || f->nreads >= f->cap_reads)
    flush_block(f,0);
}

/* returns 1 for FASTQ, 0 for FASTA, 2 for SAM, 3 for BAM, -1 unknown; sets *root (malloc'd) and *dir */
static int classify(const char *path, char **root, char **dir)
{ static const char *suf[] = { ".fastq.gz", ".fasta.gz", ".fq.gz", ".fa.gz", ".fastq", ".fasta",
                               ".fq", ".fa", ".sam", ".bam", NULL };
  static const int   isq[] = { 1, 0, 1, 0, 1, 0, 1, 0, 2, 3 };
  const char *slash = strrchr(path,'/');
  const char *base  = slash ? slash+1 : path;
  size_t      bl    = strlen(base);
  int i;

  for (i = 0; suf[i] != NULL; i++)
    { size_t sl = strlen(suf[i]);
      if (bl > sl && strcmp(base+bl-sl,suf[i]) == 0)
        { *root = strndup(base,bl-sl);
          *dir  = slash ? strndup(path,(size_t) (slash-path)) : strdup(".");
          return (isq[i]);
        }
    }
  return (-1);
}

/* The reference's name resolution (Fetch_File, io.c:136-160): the argument may omit its extension, or
   name the uncompressed file when only the .gz exists; candidates are tried in the reference's order
   and the first that opens wins.  Returns a malloc'd path, or exits like the reference. */
static char *resolve_input(const char *arg)
{ static const char *strip[] = { ".cram", ".bam", ".sam", ".db", ".dam", ".fastq", ".fasta", ".fq", ".fa",
                                 ".fastq.gz", ".fasta.gz", ".fastq", ".fasta", ".fq.gz", ".fa.gz", ".fq", ".fa" };
  static const char *add[]   = { ".cram", ".bam", ".sam", ".db", ".dam", ".fastq", ".fasta", ".fq", ".fa",
                                 ".fastq.gz", ".fasta.gz", ".fastq.gz", ".fasta.gz", ".fq.gz", ".fa.gz",
                                 ".fq.gz", ".fa.gz" };
  size_t al = strlen(arg);
  int    i;

  for (i = 0; i < 17; i++)
    { size_t sl = strlen(strip[i]);
      size_t keep = (al > sl && strcmp(arg+al-sl,strip[i]) == 0) ? al-sl : al;
      char  *path = malloc(keep+strlen(add[i])+1);
      int    fd;
      if (path == NULL)
        { fprintf(stderr,"%s: Out of memory\n",Prog_Name); exit (1); }
      memcpy(path,arg,keep);
      strcpy(path+keep,add[i]);
      fd = open(path,O_RDONLY);
      if (fd >= 0)
        { close(fd);
          return (path);
        }
      free(path);
    }
  fprintf(stderr,"\n%s: Cannot open %s as a .cram|[bs]am|f{ast}[aq][.gz]|db|dam file\n",Prog_Name,arg);
  exit (1);
}

/* Hand the file text to the library in large pieces; the record structure is resolved on the GPU
   (fk_push_fastq / fk_push_fasta).  Used unless -bc / -x ask for host-side read offsets (or -c on
   FASTA: compression across the line breaks of a record is left to the host parser). */
#define RAW_BYTES (64 << 20)

/* A chunk of a plain file is copied out of the page cache by READERS threads at once: one thread's
   read() moves ~5 GB/s, which is what would otherwise bound the ingest of FASTQ text that the
   device parses at several hundred GB/s.  The next piece is read (by a helper thread) while the
   current one crosses PCIe and is parsed. */
#define READERS 16

typedef struct
  { int    fd;
    char  *dst;
    off_t  off;
    size_t len;
    ssize_t got;
  } Read_Job;

static void *read_job(void *arg)
{ Read_Job *j = (Read_Job *) arg;
  size_t done = 0;
  while (done < j->len)
    { ssize_t r = pread(j->fd,j->dst+done,j->len-done,j->off+(off_t) done);
      if (r <= 0) break;
      done += (size_t) r;
    }
  j->got = (ssize_t) done;
  return (NULL);
}

/* up to max bytes at offset off; returns the number of bytes read (short only at the end of the file) */
static int parallel_read(int fd, char *dst, off_t off, size_t max)
{ Read_Job  job[READERS];
  pthread_t th[READERS];
  size_t    per = (max/READERS + 4095) & ~(size_t) 4095;
  int       t, n = 0;

  for (t = 0; t < READERS; t++)
    { size_t lo = per*t;
      job[t].fd = fd; job[t].dst = dst+lo; job[t].off = off+(off_t) lo;
      job[t].len = (lo >= max) ? 0 : ((lo+per > max) ? max-lo : per);
      job[t].got = 0;
      if (t > 0 && job[t].len > 0)
        pthread_create(th+t,NULL,read_job,job+t);
    }
  read_job(job);
  for (t = 1; t < READERS; t++)
    if (job[t].len > 0)
      pthread_join(th[t],NULL);
  for (t = 0; t < READERS; t++)
    { n += (int) job[t].got;
      if ((size_t) job[t].got < job[t].len)
        break;
    }
  return (n);
}

/* First record start at or after byte `from` of a plain FASTA / FASTQ file (the file size if there is none):
   the reference stripes its input over threads the same way (io.c:2455-2521).  FASTA: a '>' that begins a
   line which follows a line that does not begin with '>' (the line after a header is sequence to the reference
   whatever it begins with, io.c:689-734: a record without bases swallows the header of the next one).  FASTQ: a line that begins with '@' whose next-but-one line begins with '+' (a quality line may begin
   with '@' too, but then the next-but-one line is a sequence). */
static off_t record_start(int fd, off_t from, off_t size, int fastq)
{ enum { WIN = 1 << 20 };
  static char *win = NULL;
  off_t  pos = from;
  off_t  cand = -1;           /* FASTQ: start of a line that begins with '@' */
  int    lines_after = 0, atbol;
  int    prev_gt = -1, cur_gt = -1;   /* FASTA: does the previous / this line begin with '>' (-1: not seen) */

  if (from <= 0) return (0);
  if (win == NULL && (win = malloc(WIN)) == NULL)
    { fprintf(stderr,"%s: Out of memory\n",Prog_Name); exit (1); }
  { char c;                   /* are we at the beginning of a line? */
    atbol = (pread(fd,&c,1,from-1) == 1 && c == '\n');
  }
  while (pos < size)
    { ssize_t n = pread(fd,win,(pos == from) ? 65536 : WIN,pos), i;       /* mostly found within the first lines */
      if (n <= 0) break;
      for (i = 0; i < n; i++)
        { char c = win[i];
          if (atbol)
            { if (!fastq)
                { if (c == '>' && prev_gt == 0) return (pos+i);
                  cur_gt = (c == '>');
                }
              else
                { if (cand >= 0)
                    { lines_after += 1;
                      if (lines_after == 2)
                        { if (c == '+') return (cand);
                          cand = -1;
                        }
                    }
                  if (cand < 0 && c == '@')
                    { cand = pos+i; lines_after = 0; }
                }
            }
          atbol = (c == '\n');
          if (atbol)
            { prev_gt = cur_gt; cur_gt = -1; }
        }
      pos += n;
    }
  return (size);
}

typedef struct
  { gzFile in;
    int    fd;
    char  *dst;
    off_t  off, fend;
    int    got;
  } Fetch_Job;

static void *fetch_job(void *arg)
{ Fetch_Job *j = (Fetch_Job *) arg;
  if (j->in != NULL)
    j->got = gzread(j->in,j->dst,RAW_BYTES);
  else
    { size_t max = RAW_BYTES;
      if (j->fend >= 0 && j->fend - j->off < (off_t) RAW_BYTES)
        max = (j->fend > j->off) ? (size_t) (j->fend - j->off) : 0;
      j->got = (max > 0) ? parallel_read(j->fd,j->dst,j->off,max) : 0;
    }
  if (j->got < 0) j->got = 0;
  return (NULL);
}

static void scan_text_on_device(Feeder *f, const char *path, int fastq)
{ gzFile in = NULL;
  int    fd = -1;
  static char *raw = NULL, *raw2 = NULL;
  int     phase = fastq ? 0 : 2, n;
  size_t  pl = strlen(path);

  /* plain files are read straight into the pinned buffer; zlib only for .gz */
  if (pl > 3 && strcmp(path+pl-3,".gz") == 0)
    { in = gzopen(path,"rb");
      if (in != NULL) gzbuffer(in,1 << 20);
    }
  else
    fd = open(path,O_RDONLY);
  if (in == NULL && fd < 0)
    { fprintf(stderr,"%s: Cannot open %s for reading\n",Prog_Name,path);
      exit (1);
    }
  if (raw == NULL && fk_host_alloc(RAW_BYTES,(void **) &raw) != FK_OK)
    die(NULL,"pinned read buffer");
  flush_block(f,0);                        /* keep the order of reads across input files */
  off_t foff = 0, fend = -1;
  if (fd >= 0 && RANK >= 0 && NGPUS > 1)   /* this rank's stripe of the file, cut at record starts */
    { off_t size = lseek(fd,0,SEEK_END);
      foff = record_start(fd,(off_t) ((double) size*RANK/NGPUS),size,fastq);
      fend = (RANK+1 == NGPUS) ? size : record_start(fd,(off_t) ((double) size*(RANK+1)/NGPUS),size,fastq);
    }
  if (raw2 == NULL && fk_host_alloc(RAW_BYTES,(void **) &raw2) != FK_OK)
    die(NULL,"pinned read buffer");
  { Fetch_Job fj;
    pthread_t th;
    char     *cur = raw, *nxt = raw2;
    int       pending = 0;
    fj.in = in; fj.fd = fd; fj.fend = fend;
    /* piece 0 is read here; from then on piece i+1 is fetched by a helper while piece i is pushed */
    fj.dst = cur; fj.off = foff;
    fetch_job(&fj);
    n = fj.got;
    foff += n;
    while (n > 0)
      { int64_t nr = 0, nb = 0;
        fj.dst = nxt; fj.off = foff;
        pthread_create(&th,NULL,fetch_job,&fj);
        pending = 1;
        if (fastq)
          { if (fk_push_fastq(f->ctx,cur,n,COMPRESS ? FK_FASTQ_HOCO : 0,&phase,&nr,&nb) != FK_OK)
              die(f->ctx,"fk_push_fastq");
          }
        else if (fk_push_fasta(f->ctx,cur,n,0,&phase,&nr,&nb) != FK_OK)
          die(f->ctx,"fk_push_fasta");
        f->totrds += nr;
        f->totbps += nb;
        if (fend >= 0) f->striped_rds += nr;
        pthread_join(th,NULL);
        pending = 0;
        n = fj.got;
        foff += n;
        { char *t = cur; cur = nxt; nxt = t; }
      }
    (void) pending;
  }
  if (!fastq && fk_push_fasta(f->ctx,NULL,0,1,&phase,NULL,NULL) != FK_OK)     /* ends the last record */
    die(f->ctx,"fk_push_fasta");
  if (in != NULL) gzclose(in); else close(fd);
}

/* Plain FASTA / FASTQ text, packed by the reader threads.  The file is cut at record starts into pieces of PK_PIECE
   bytes; the workers (one per core, 64 at most) each take the next piece, resolve its lines and pack the bases two
   bits each (plus the stretches that hold no acgt), and hand the piece to fk_push_packed: a quarter of the bytes
   cross PCIe, none of the header / quality text does, and the line scan runs on all host cores at once instead of on
   one copy stream.  The line rules are those of the device parsers (io.c:678-734: four lines per FASTQ record, every
   byte of a sequence line is a base; a FASTA record runs to the next line that begins with '>').  The order in
   which the pieces arrive is not that of the file: used for the runs whose outputs do not depend on it (no -p, -x, -bc, -c). */
static off_t PK_PIECE = (off_t) 64 << 20;     /* FASTK_AMD_PIECE=<bytes> overrides (tests cut small files into many pieces) */

static uint8_t PK_PAIR[65536];     /* two bases -> 4 bits (first base high), 0x80 if either is not acgtACGT */
static int8_t  PK_ONE[256];
static int     PK_AVX2 = 0;       /* the host has AVX2: 32 bases a step */

typedef struct
  { Feeder         *f;
    int             fd, fastq;
    const off_t    *cut;          /* ncut+1 piece boundaries, all at record starts */
    int             ncut;
    int             next;         /* next piece to take (under lock) */
    int64_t         totrds, totbps;
    double          t_read, t_pack, t_push, t_hold;   /* summed over the workers (-v) */
    const char     *map;          /* the whole file mapped, or NULL (FASTK_AMD_MMAP=0, mmap failed): pieces are pread */
    pthread_mutex_t push;
    uint8_t        *pinned;       /* one pinned region for all workers (64 hipHostMalloc calls at once take >1 s) */
    size_t          slice;
    int             nslices, taken;
    double          t_setup;
    int             failed;
    pthread_mutex_t lock;
  } Pk_Job;

typedef struct
  { char    *text;  size_t text_cap;
    uint8_t *codes; size_t codes_cap;        /* pinned: a slice of the job's region, or its own if a piece is larger */
    int      codes_own;
    int32_t *rlen;  size_t rlen_cap;
    int64_t *inv;   size_t inv_cap;          /* pairs */
    int64_t  nb;
    int      nreads, ninv;
  } Pk_Buf;

static void pk_tables(void)
{ static int done = 0;
  int a, b;
  if (done) return;
  memset(PK_ONE,-1,sizeof(PK_ONE));
  PK_ONE['a'] = PK_ONE['A'] = 0; PK_ONE['c'] = PK_ONE['C'] = 1;
  PK_ONE['g'] = PK_ONE['G'] = 2; PK_ONE['t'] = PK_ONE['T'] = 3;
  for (a = 0; a < 256; a++)
    for (b = 0; b < 256; b++)
      PK_PAIR[a | (b << 8)] = (PK_ONE[a] < 0 || PK_ONE[b] < 0) ? 0x80 : (uint8_t) ((PK_ONE[a] << 2) | PK_ONE[b]);
#if defined(__x86_64__)
  PK_AVX2 = __builtin_cpu_supports("avx2") && !(getenv("FASTK_AMD_SCALAR") != NULL);
#endif
  done = 1;
}

static void pk_oom(void)
{ fprintf(stderr,"%s: Out of memory\n",Prog_Name); exit (1); }

static inline void pk_invalid(Pk_Buf *b, int64_t pos)
{ if (b->ninv > 0 && b->inv[2*(b->ninv-1)] + b->inv[2*(b->ninv-1)+1] == pos)
    { b->inv[2*(b->ninv-1)+1] += 1; return; }
  if ((size_t) b->ninv+1 > b->inv_cap)
    { b->inv_cap = b->inv_cap*2 + 1024;
      if ((b->inv = realloc(b->inv,sizeof(int64_t)*2*b->inv_cap)) == NULL) pk_oom();
    }
  b->inv[2*b->ninv] = pos; b->inv[2*b->ninv+1] = 1;
  b->ninv += 1;
}

#if defined(__x86_64__)
#include <immintrin.h>

/* 32 bases -> 8 bytes, first base in the two high bits of the first byte; returns 0 (nothing stored) if one of
   them is not acgtACGT */
__attribute__((target("avx2")))
static inline int pk_pack32(const unsigned char *s, uint64_t *out)
{ const __m256i v  = _mm256_loadu_si256((const __m256i *) s);
  const __m256i u  = _mm256_and_si256(v,_mm256_set1_epi8((char) 0xDF));
  const __m256i ok = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(u,_mm256_set1_epi8('A')),
                                                     _mm256_cmpeq_epi8(u,_mm256_set1_epi8('C'))),
                                     _mm256_or_si256(_mm256_cmpeq_epi8(u,_mm256_set1_epi8('G')),
                                                     _mm256_cmpeq_epi8(u,_mm256_set1_epi8('T'))));
  if (_mm256_movemask_epi8(ok) != -1)
    return (0);
  { /* (c >> 1) & 3 = a 0  c 1  t 2  g 3;  x ^ (x >> 1) = a 0  c 1  g 2  t 3 */
    const __m256i x  = _mm256_and_si256(_mm256_srli_epi16(v,1),_mm256_set1_epi8(3));
    const __m256i c  = _mm256_xor_si256(x,_mm256_and_si256(_mm256_srli_epi16(x,1),_mm256_set1_epi8(1)));
    const __m256i p2 = _mm256_maddubs_epi16(c,_mm256_set1_epi16(0x0104));        /* b0*4 + b1 per 16-bit lane */
    const __m256i p4 = _mm256_madd_epi16(p2,_mm256_set1_epi32(0x00010010));      /* *16 + next per 32-bit lane */
    const __m256i sh = _mm256_shuffle_epi8(p4,_mm256_setr_epi8(0,4,8,12,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,
                                                               0,4,8,12,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1,-1));
    const __m256i pm = _mm256_permutevar8x32_epi32(sh,_mm256_setr_epi32(0,4,1,1,1,1,1,1));
    *out = (uint64_t) _mm_cvtsi128_si64(_mm256_castsi256_si128(pm));
  }
  return (1);
}

/* the whole 32-base groups of s[0..n) appended at bit offset sh of *o; returns the bases consumed (stops early at a
   group that holds a byte that is no base) */
__attribute__((target("avx2")))
static int64_t pk_bases_avx2(const unsigned char *s, int64_t n, uint8_t **op, int sh, unsigned *carryp)
{ int64_t  i = 0;
  uint8_t *o = *op;
  uint64_t w;
  if (sh == 0)
    { for (; i+32 <= n; i += 32)
        { if (!pk_pack32(s+i,&w)) break;
          memcpy(o,&w,8);
          o += 8;
        }
    }
  else
    { uint64_t carry = *carryp;                                  /* the taken high bits of the current byte */
      for (; i+32 <= n; i += 32)
        { uint64_t be, st;
          if (!pk_pack32(s+i,&w)) break;
          be = __builtin_bswap64(w);                             /* first base most significant */
          st = __builtin_bswap64((be >> sh) | (carry << 56));
          memcpy(o,&st,8);
          o += 8;
          carry = (be << (8-sh)) & 0xff;
        }
      *carryp = (unsigned) carry;
    }
  *op = o;
  return (i);
}
#endif

/* append the n bases at s to the packed block (codes has room: it is sized from the piece) */
static void pk_bases(Pk_Buf *b, const unsigned char *s, int64_t n)
{ int64_t  i = 0, nb = b->nb;
  uint8_t *o = b->codes + (nb >> 2);
  const int sh = (int) (nb & 3)*2;          /* bits of *o already taken */
  unsigned carry = sh ? *o : 0;

#if defined(__x86_64__)
  if (PK_AVX2)
    while (i+32 <= n)
      { int64_t j;
        i += pk_bases_avx2(s+i,n-i,&o,sh,&carry);
        if (i+32 > n) break;
        for (j = 0; j < 32; j += 4)         /* a group with a non-base: its eight bytes the slow way */
          { unsigned g = 0;
            int      q;
            for (q = 0; q < 4; q++)
              { int v = PK_ONE[s[i+j+q]];
                if (v < 0) { pk_invalid(b,nb+i+j+q); v = 0; }
                g = (g << 2) | (unsigned) v;
              }
            *o++  = (uint8_t) (carry | (g >> sh));
            carry = (g << (8-sh)) & 0xff;
          }
        i += 32;
      }
#endif
  for (; i+4 <= n; i += 4)
    { unsigned g, hi = PK_PAIR[s[i] | (s[i+1] << 8)], lo = PK_PAIR[s[i+2] | (s[i+3] << 8)];
      if ((hi | lo) & 0x80)
        { int j;
          g = 0;
          for (j = 0; j < 4; j++)
            { int v = PK_ONE[s[i+j]];
              if (v < 0) { pk_invalid(b,nb+i+j); v = 0; }
              g = (g << 2) | (unsigned) v;
            }
        }
      else
        g = (hi << 4) | lo;
      *o++  = (uint8_t) (carry | (g >> sh));
      carry = (g << (8-sh)) & 0xff;
    }
  { int used = sh;                           /* the tail, base by base */
    for (; i < n; i++)
      { int v = PK_ONE[s[i]];
        if (v < 0) { pk_invalid(b,nb+i); v = 0; }
        carry |= (unsigned) v << (6-used);
        used += 2;
        if (used == 8)
          { *o++ = (uint8_t) carry; carry = 0; used = 0; }
      }
    if (used > 0)
      *o = (uint8_t) carry;
  }
  b->nb = nb+n;
}

static inline void pk_end_read(Pk_Buf *b, int64_t start)
{ if ((size_t) b->nreads+1 > b->rlen_cap)
    { b->rlen_cap = b->rlen_cap*2 + 4096;
      if ((b->rlen = realloc(b->rlen,sizeof(int32_t)*b->rlen_cap)) == NULL) pk_oom();
    }
  b->rlen[b->nreads++] = (int32_t) (b->nb-start);
}

/* One piece of text -- whole records: it begins at a record start and ends where the next piece begins or the file
   ends -- into b: the packed bases, the read lengths, the stretches without acgt.  The reference's line rules
   (io.c:678-738), its oddities included: see the comments inside. */
static void pk_parse_piece(Pk_Buf *b, const unsigned char *p, const unsigned char *e, int fastq)
{ b->nb = 0; b->nreads = 0; b->ninv = 0;
  if (fastq)
    while (p < e)
      { const unsigned char *nl;
        int64_t start = b->nb;
        nl = memchr(p,'\n',(size_t) (e-p));                 /* header */
        if (nl == NULL) break;
        p = nl+1;
        nl = memchr(p,'\n',(size_t) (e-p));                 /* sequence */
        if (nl == NULL)                                     /* the file ends inside it: the reference never ends */
          break;                                            /* this read (io.c:698-705,738) */
        pk_bases(b,p,nl-p);
        pk_end_read(b,start);
        p = (nl < e) ? nl+1 : e;
        if (p < e)
          { nl = memchr(p,'\n',(size_t) (e-p));             /* + */
            p  = (nl == NULL) ? e : nl+1;
          }
        if (p < e)
          { nl = memchr(p,'\n',(size_t) (e-p));             /* quality */
            p  = (nl == NULL) ? e : nl+1;
          }
      }
  else
    { int     open = 0;
      int64_t start = 0;
      int     first = 1, after_header = 0, cut = 0;
      while (p < e)
        { const unsigned char *nl = memchr(p,'\n',(size_t) (e-p));
          if (nl == NULL) { nl = e; cut = 1; }
          /* the reference's machine (io.c:685-734): the line after a header is sequence whatever it begins with */
          if (first || (*p == '>' && !after_header))
            { if (open) pk_end_read(b,start);
              open = 1; start = b->nb; first = 0; after_header = 1;
            }
          else
            { pk_bases(b,p,nl-p);
              after_header = 0;
            }
          p = (nl < e) ? nl+1 : e;
        }
      /* a file that does not end in a newline: the reference ends a read at the '>' or the end of file that follows
         a newline (io.c:717-738), so the bases of the last record are dropped */
      if (open && after_header)                  /* ... and a header at the very end opens no read */
        open = 0;
      if (open && cut)
        { b->nb = start;
          while (b->ninv > 0 && b->inv[2*(b->ninv-1)] >= start)
            b->ninv -= 1;
          if (b->ninv > 0 && b->inv[2*(b->ninv-1)]+b->inv[2*(b->ninv-1)+1] > start)
            b->inv[2*(b->ninv-1)+1] = start-b->inv[2*(b->ninv-1)];
          open = 0;
        }
      if (open) pk_end_read(b,start);
    }
}

static void *pk_worker(void *arg)
{ Pk_Job *job = (Pk_Job *) arg;
  Pk_Buf  b;
  memset(&b,0,sizeof(b));
  for (;;)
    { int    k;
      off_t  beg, end;
      size_t len, done = 0;
      const unsigned char *p;

      pthread_mutex_lock(&job->lock);
      k = (job->failed || job->next >= job->ncut) ? -1 : job->next++;
      pthread_mutex_unlock(&job->lock);
      if (k < 0) break;
      beg = job->cut[k]; end = job->cut[k+1];
      len = (size_t) (end-beg);
      if (len == 0) continue;
      if (job->map == NULL && len+8 > b.text_cap)
        { free(b.text);
          b.text_cap = len + len/8 + 64;
          if ((b.text = malloc(b.text_cap)) == NULL) pk_oom();
        }
      if (b.codes == NULL)
        { pthread_mutex_lock(&job->lock);
          if (job->taken < job->nslices)
            { b.codes = job->pinned + job->slice*(size_t) job->taken++;
              b.codes_cap = job->slice;
            }
          pthread_mutex_unlock(&job->lock);
        }
      if (len/4+64 > b.codes_cap)
        { if (b.codes_own) fk_host_free(b.codes);
          b.codes_cap = len/4 + len/32 + 64;
          b.codes_own = 1;
          if (fk_host_alloc((int64_t) b.codes_cap,(void **) &b.codes) != FK_OK) pk_oom();
        }
      double t0 = now(), t1, t2, t3;
      if (job->map != NULL)
        { off_t lo = beg & ~(off_t) 4095;
#ifndef MADV_POPULATE_READ
#define MADV_POPULATE_READ 22
#endif
          madvise((void *) (job->map+lo),(size_t) (end-lo),MADV_POPULATE_READ);     /* one fault for the piece */
          p = (const unsigned char *) job->map+beg;
        }
      else
        { while (done < len)
            { ssize_t r = pread(job->fd,b.text+done,len-done,beg+(off_t) done);
              if (r <= 0) break;
              done += (size_t) r;
            }
          if (done < len)
            { fprintf(stderr,"%s: short read of the input\n",Prog_Name); exit (1); }
          p = (const unsigned char *) b.text;
        }
      t1 = now();
      pk_parse_piece(&b,p,p+len,job->fastq);
      if (job->map != NULL)                      /* drop the piece's page-table entries here, in parallel: one munmap */
        { off_t lo = (beg+4095) & ~(off_t) 4095, hi = end & ~(off_t) 4095;     /* of 150 GB at the end takes over a second */
          if (hi > lo)
            madvise((void *) (job->map+lo),(size_t) (hi-lo),MADV_DONTNEED);
        }
      t2 = now();
      { int rc = FK_OK;
        double th;
        pthread_mutex_lock(&job->push);
        th = now();
        if (b.nreads > 0)
          rc = fk_push_packed(job->f->ctx,b.codes,b.nb,b.rlen,b.nreads,b.inv,b.ninv,0,k % NTHREADS);
        th = now()-th;
        pthread_mutex_unlock(&job->push);
        if (rc != FK_OK)
          { pthread_mutex_lock(&job->lock);
            job->failed = 1;
            pthread_mutex_unlock(&job->lock);
            break;
          }
        t3 = now();
        pthread_mutex_lock(&job->lock);
        job->t_hold += th;
        pthread_mutex_unlock(&job->lock);
      }
      pthread_mutex_lock(&job->lock);
      job->t_read += t1-t0; job->t_pack += t2-t1; job->t_push += t3-t2;
      job->totrds += b.nreads;
      job->totbps += b.nb;
      pthread_mutex_unlock(&job->lock);
    }
  free(b.text); free(b.rlen); free(b.inv);
  if (b.codes_own) fk_host_free(b.codes);
  return (NULL);
}

static void scan_text_packed(Feeder *f, const char *path, int fastq)
{ int     fd = open(path,O_RDONLY);
  off_t   size, foff, fend, *cut;
  int     ncut = 0, nthr, t;
  long    ncpu = sysconf(_SC_NPROCESSORS_ONLN);
  Pk_Job  job;
  pthread_t th[64];

  if (fd < 0)
    { fprintf(stderr,"%s: Cannot open %s for reading\n",Prog_Name,path);
      exit (1);
    }
  pk_tables();
  if (getenv("FASTK_AMD_PIECE") != NULL && atoll(getenv("FASTK_AMD_PIECE")) > 0)
    PK_PIECE = (off_t) atoll(getenv("FASTK_AMD_PIECE"));
  flush_block(f,0);
  size = lseek(fd,0,SEEK_END);
  if (size < 0)                            /* (main sends only regular files here) */
    { fprintf(stderr,"%s: %s is not a seekable file\n",Prog_Name,path);
      exit (1);
    }
  foff = 0; fend = size;
  if (RANK >= 0 && NGPUS > 1)              /* this rank's stripe of the file, cut at record starts */
    { foff = record_start(fd,(off_t) ((double) size*RANK/NGPUS),size,fastq);
      fend = (RANK+1 == NGPUS) ? size : record_start(fd,(off_t) ((double) size*(RANK+1)/NGPUS),size,fastq);
    }
  double t_begin = now(), t_cut = now();
  cut = malloc(sizeof(off_t)*(size_t) ((fend-foff)/PK_PIECE+3));
  if (cut == NULL) pk_oom();
  cut[0] = foff;
  while (cut[ncut] < fend)
    { off_t nx = cut[ncut]+PK_PIECE;
      nx = (nx >= fend) ? fend : record_start(fd,nx,size,fastq);
      if (nx > fend) nx = fend;
      cut[++ncut] = nx;
    }
  t_cut = now()-t_cut;
  memset(&job,0,sizeof(job));
  job.f = f; job.fd = fd; job.fastq = fastq; job.cut = cut; job.ncut = ncut;
  pthread_mutex_init(&job.lock,NULL);
  pthread_mutex_init(&job.push,NULL);
  /* the pieces are packed straight out of the page cache (the file mapped, a piece populated by one madvise): a
     pread into a private buffer first moves every byte twice more and costs 10x the thread time (tools/ingest_probe.py;
     FASTK_AMD_MMAP=0 goes back to it) */
  if (!(getenv("FASTK_AMD_MMAP") != NULL && atoi(getenv("FASTK_AMD_MMAP")) == 0) && size > 0)
    { void *m = mmap(NULL,(size_t) size,PROT_READ,MAP_SHARED,fd,0);
      if (m != MAP_FAILED) job.map = (const char *) m;
    }
  /* 32 readers at most: the pushes are DMA reads of host memory, and host memory is what the packers themselves keep
     busy (150 GB of text in, 37.5 GB of codes out) -- a 17 MB block crosses PCIe at 56 GB/s on an idle host and at
     23 GB/s beside 48 threads that write memory (tools/probe/h2d_probe.cpp); 24-32 readers scan the file in 1.03-1.05 s,
     48-64 in 1.11-1.18 s (FASTK_AMD_READERS overrides) */
  nthr = (ncpu > 32) ? 32 : (int) ncpu;
  if (RANK >= 0 && NGPUS > 1) nthr /= NGPUS;
  if (getenv("FASTK_AMD_READERS") != NULL && atoi(getenv("FASTK_AMD_READERS")) > 0)
    nthr = atoi(getenv("FASTK_AMD_READERS"));
  if (nthr > 64) nthr = 64;
  if (nthr > ncut) nthr = ncut;
  if (nthr < 1) nthr = 1;
  job.t_setup = now();
  job.slice   = (size_t) (PK_PIECE/4 + PK_PIECE/16 + 65536) & ~(size_t) 4095;
  job.nslices = nthr;
  if (fk_host_alloc((int64_t) (job.slice*(size_t) nthr),(void **) &job.pinned) != FK_OK)
    { job.pinned = NULL; job.nslices = 0; }
  job.t_setup = now()-job.t_setup;
  { int made = 1;                          /* threads that really started (the pieces are handed out by a counter: fewer
                                              threads only take longer) */
    for (t = 1; t < nthr; t++)
      { if (pthread_create(th+made,NULL,pk_worker,&job) != 0)
          break;
        made += 1;
      }
    pk_worker(&job);
    for (t = 1; t < made; t++)
      pthread_join(th[t],NULL);
    nthr = made;
  }
  pthread_mutex_destroy(&job.lock);
  if (job.failed)
    die(f->ctx,"fk_push_packed");
  f->totrds += job.totrds;
  f->totbps += job.totbps;
  if (RANK >= 0 && NGPUS > 1) f->striped_rds += job.totrds;
  if (VERBOSE)
    fprintf(stderr,"  %s: %d pieces packed by %d reader threads; thread seconds: read %.2f  pack %.2f  push %.2f of which inside fk_push_packed %.2f; cutting %.2f s, pinned buffers %.2f s, wall %.2f s\n",
            path,ncut,nthr,job.t_read,job.t_pack,job.t_push,job.t_hold,t_cut,job.t_setup,now()-t_begin);
  if (job.pinned != NULL)
    fk_host_free(job.pinned);
  if (job.map != NULL)
    munmap((void *) job.map,(size_t) size);
  pthread_mutex_destroy(&job.push);
  free(cut);
  close(fd);
}

/* Where the reference's thread that is given byte `from` of a plain file begins (fast_nearest, io.c:409-490; -1: nowhere).
   FASTA: the first '>' behind a newline that is read at or behind `from` -- a '>' AT `from` is passed over, whatever is in
   front of it.  FASTQ: the first '@' behind a newline whose line was neither a single character nor ended in '+' (the
   quality line of a record may begin with '@'; the separator line in front of it is "+"). */
static off_t io_nearest(int fd, off_t from, off_t size, int fastq)
{ enum { WIN = 1 << 16 };
  static char win[WIN];
  off_t pos = from;
  /* FASTQ: the two characters in front of the one looked at, as "newline?" and "plus?" -- both start as if a separator
     line had just been read, so that nothing is accepted before a whole line has gone by.  FASTA: has a newline been
     seen, and is this the character right behind one? */
  int   before_is_nl = 1, before2_is_nl = 1, before_is_plus = 1;
  int   at_may_start = 0, behind_nl = 0;
  while (pos < size)
    { ssize_t n = pread(fd,win,WIN,pos), i;
      if (n <= 0) break;
      for (i = 0; i < n; i++)
        { const char c = win[i];
          if (fastq)
            { if (at_may_start && c == '@')
                return (pos+i);
              /* the next character begins a line that may be a header iff this one is a newline and the line it ends
                 has at least two characters and does not end in '+' */
              at_may_start   = (c == '\n') && !before2_is_nl && !before_is_plus;
              before2_is_nl  = before_is_nl;
              before_is_nl   = (c == '\n');
              before_is_plus = (c == '+');
            }
          else
            { if (behind_nl && c == '>')
                return (pos+i);
              behind_nl = (c == '\n');
            }
        }
      pos += n;
    }
  return (-1);
}

/* The input threads the reference would read these FASTA / FASTQ files with, and where each of them begins
   (io.c:2340-2521).  Any file compressed: whole files, one thread per file up to 1.5 x -T files, else -T threads of
   nfiles / T files each.  Plain files: as many threads as -T unless that leaves a thread less than 2 % of an IO block
   (200,000 bytes) -- then one per 200,000 bytes, at least one --; thread i at the first record start fast_nearest finds
   behind byte i * (all bytes) / threads, the files taken as one; an i whose byte lies at or in front of the previous
   thread's start gets no thread.  Only the .prof parts and their run numbers depend on it. */
static void input_threads(Feeder *f, char **paths, int nfiles, int fastq)
{ int64_t size[256], work = 0, w;
  int     zipped = 0, it, i, fi;
  f->nstarts = 0;
  if (nfiles > 256)
    return;
  for (i = 0; i < nfiles; i++)
    { struct stat st;
      size_t pl = strlen(paths[i]);
      if (stat(paths[i],&st) != 0)
        return;
      size[i] = (int64_t) st.st_size;
      work += size[i];
      if (pl > 3 && strcmp(paths[i]+pl-3,".gz") == 0)
        zipped = 1;
    }
  if (zipped)
    { it = (nfiles <= 1.5*NTHREADS) ? nfiles : NTHREADS;
      if (it > 256) it = 256;
      for (i = 0; i < it; i++)
        { f->st_file[i] = (int) (((int64_t) i*nfiles)/it);
          f->starts[i]  = 0;
        }
      f->nstarts = it;
      return;
    }
  if ((double) (work/NTHREADS) < .02*10000000ll)
    { it = (int) ((double) work/(.02*10000000ll));
      if (it <= 0) it = 1;
    }
  else
    it = NTHREADS;
  if (it > 256) it = 256;
  fi = 0;
  w  = size[0];
  for (i = 0; i < it; i++)
    { const int64_t target = ((int64_t) i*work)/it;
      int64_t b;
      off_t   at;
      while ((double) w < (double) target - .01*10000000ll && fi+1 < nfiles)
        { fi += 1;
          w  += size[fi];
        }
      b = target - (w-size[fi]);
      if (b < 0) b = 0;
      if (f->nstarts > 0 && (fi < f->st_file[f->nstarts-1] || (fi == f->st_file[f->nstarts-1] && b <= f->starts[f->nstarts-1])))
        continue;
      if (b == 0)
        { f->st_file[f->nstarts] = fi; f->starts[f->nstarts++] = 0; }
      else
        { int fd = open(paths[fi],O_RDONLY);
          at = (fd >= 0) ? io_nearest(fd,(off_t) b,(off_t) size[fi],fastq) : -1;
          if (fd >= 0) close(fd);
          if (at < 0)                      /* no record behind b in this file: the next file's first, if there is one */
            { if (fi+1 >= nfiles)
                break;
              f->st_file[f->nstarts] = fi+1; f->starts[f->nstarts++] = 0;
            }
          else
            { f->st_file[f->nstarts] = fi; f->starts[f->nstarts++] = at; }
        }
    }
  if (f->nstarts > 0)
    { f->st_file[0] = 0; f->starts[0] = 0; }
}

static void scan_file(Feeder *f, const char *path, int fastq)
{ gzFile  in = gzopen(path,"rb");
  static unsigned char buf[1 << 20];
  int     state = 0;    /* 0 record start, 1 header, 2 fastq seq, 3 '+' line, 4 quality, 5 fasta seq, 6 fasta eol */
  int     n, i;
  off_t   base = 0;     /* bytes of the file in front of buf */
#define RECORD_AT(off) do { while (f->tid+1 < f->nstarts && (f->cur_file > f->st_file[f->tid+1]                          \
                                                          || (f->cur_file == f->st_file[f->tid+1] && (off) >= f->starts[f->tid+1])))   \
                              { flush_block(f,0); f->tid += 1; } } while (0)

  if (in == NULL)
    { fprintf(stderr,"%s: Cannot open %s for reading\n",Prog_Name,path);
      exit (1);
    }
  gzbuffer(in,1 << 20);
  while (!range_done(f) && (n = gzread(in,buf,sizeof(buf))) > 0)
   { for (i = 0; i < n; i++)
      { int c = buf[i];
        switch (state)
        { case 0: RECORD_AT(base+i); state = 1; break;
          case 1: if (c == '\n') state = fastq ? 2 : 5; break;
          case 2: if (c != '\n') add_base(f,c); else { end_read(f); state = 3; } break;
          case 3: if (c == '\n') state = 4; break;
          case 4: if (c == '\n') state = 0; break;
          case 6: if (c == '>') { end_read(f); RECORD_AT(base+i); state = 1; }
                  else if (c != '\n') { add_base(f,c); state = 5; }
                  break;
          case 5: if (c == '\n') state = 6; else add_base(f,c); break;
        }
      }
     base += n;
   }
#undef RECORD_AT
  if (state == 6)
    end_read(f);
  else if (state == 5 || state == 2)     /* the file ends inside a sequence line: the reference ends a read only at the */
    { int64_t start = f->boff[f->nreads];     /* newline (FASTQ) or at the '>' / end of file behind one (FASTA), */
      f->totbps -= f->olen-start;             /* io.c:698-738 -- these bases belong to no read */
      f->olen = start;
    }
  gzclose(in);
}

/* SAM and BAM input lives in input_formats.c (SURVEY section 2 row 14 keeps only the FASTA / FASTQ feed in scope: those
   two readers are complete as they are and are not extended); they hand bases and read ends to the block feeder
   through these two calls. */
void feeder_base(Feeder *f, int c)   { add_base(f,c); }
void feeder_end_read(Feeder *f)      { end_read(f); }
int  feeder_done(const Feeder *f)    { return (range_done(f)); }
const char *feeder_prog_name(void)   { return (Prog_Name); }

/* -G<n>: the parent never touches a GPU.  It starts n copies of this program, one per GPU, each told its rank
   and the path of a file through which rank 0 hands the RCCL unique id to the others, and waits for them.
   FK_RANKS_SHARE_GPU=1 (test rig for boxes with a single GPU): all ranks use device 0 and get a NCCL_HOSTID of
   their own, so that RCCL accepts them and moves the payload over its socket transport. */
static int launch_ranks(int argc, char **argv)
{ char  iddir[64], idfile[128], self[4096];
  pid_t pid[64];
  int   r, left, status, rc = 0;
  ssize_t sl = readlink("/proc/self/exe",self,sizeof(self)-1);

  if (sl <= 0 || NGPUS > 64)
    { fprintf(stderr,"%s: cannot start %d ranks\n",Prog_Name,NGPUS); exit (1); }
  self[sl] = '\0';
  /* the id travels through a file in a directory of our own (mode 0700): nobody else can plant or follow it */
  snprintf(iddir,sizeof(iddir),"/tmp/fastk_amd_XXXXXX");
  if (mkdtemp(iddir) == NULL)
    { fprintf(stderr,"%s: cannot create a directory in /tmp\n",Prog_Name); exit (1); }
  snprintf(idfile,sizeof(idfile),"%s/id",iddir);
  for (r = 0; r < NGPUS; r++)
    { pid[r] = fork();
      if (pid[r] < 0)
        { fprintf(stderr,"%s: fork failed\n",Prog_Name);
          for (left = 0; left < r; left++)
            kill(pid[left],SIGKILL);
          exit (1);
        }
      if (pid[r] == 0)
        { char **av = malloc(sizeof(char *)*(argc+3));
          char  *a1 = malloc(64), *a2 = malloc(300);
          int    k;
          for (k = 0; k < argc; k++) av[k] = argv[k];
          snprintf(a1,64,"--fk-rank=%d",r);
          snprintf(a2,300,"--fk-idfile=%s",idfile);
          av[argc] = a1; av[argc+1] = a2; av[argc+2] = NULL;
          if (getenv("FK_RANKS_SHARE_GPU") != NULL)
            { char host[64];
              snprintf(host,sizeof(host),"fk-rank-%d",r);
              setenv("NCCL_HOSTID",host,1);
              setenv("NCCL_SOCKET_IFNAME","lo",0);
              setenv("NCCL_IB_DISABLE","1",0);
            }
          execv(self,av);
          fprintf(stderr,"%s: cannot start rank %d\n",Prog_Name,r);
          _exit (1);
        }
    }
  /* whichever rank ends first is looked at first: when one fails its peers sit in a collective that will never
     complete, so they are ended instead of waited for */
  for (left = NGPUS; left > 0; left--)
    { pid_t w = waitpid(-1,&status,0);
      if (w < 0)
        { rc = 1; break; }
      for (r = 0; r < NGPUS; r++)
        if (pid[r] == w)
          pid[r] = -1;
      if (!WIFEXITED(status) || WEXITSTATUS(status) != 0)
        { if (rc == 0)
            { fprintf(stderr,"%s: a rank failed; ending the others\n",Prog_Name);
              for (r = 0; r < NGPUS; r++)
                if (pid[r] > 0)
                  kill(pid[r],SIGTERM);
              usleep(200000);
              for (r = 0; r < NGPUS; r++)
                if (pid[r] > 0)
                  kill(pid[r],SIGKILL);
            }
          rc = 1;
        }
    }
  unlink(idfile);
  snprintf(self,sizeof(self),"%s.tmp",idfile);
  unlink(self);
  rmdir(iddir);
  return (rc);
}

/* rank 0 writes the 128 bytes (temporary name, then rename: readers never see a partial file) */
static void share_unique_id(char *id)
{ char tmp[400];
  int  tries;
  if (RANK == 0)
    { FILE *f;
      if (fk_shard_unique_id(id) != FK_OK)
        die(NULL,"fk_shard_unique_id");
      snprintf(tmp,sizeof(tmp),"%s.tmp",ID_FILE);
      f = fopen(tmp,"wb");
      if (f == NULL || fwrite(id,1,128,f) != 128 || fclose(f) != 0 || rename(tmp,ID_FILE) != 0)
        { fprintf(stderr,"%s: cannot write %s\n",Prog_Name,ID_FILE); exit (1); }
      return;
    }
  for (tries = 0; tries < 6000; tries++)         /* up to a minute */
    { FILE *f = fopen(ID_FILE,"rb");
      if (f != NULL)
        { size_t n = fread(id,1,128,f);
          fclose(f);
          if (n == 128) return;
        }
      usleep(10000);
    }
  fprintf(stderr,"%s: rank %d never received the RCCL id\n",Prog_Name,RANK);
  exit (1);
}

int main(int argc, char *argv[])
{ fk_params  prm;
  fk_ctx    *ctx;
  fk_result *res;
  Feeder     feed;
  char      *root = NULL, *dir = NULL, name[4096];
  int        i, j, nfiles, ftype = -1, all_plain = 1;

  int    argc0 = argc;
  char **argv0 = malloc(sizeof(char *)*(argc+4));
  for (i = 0; i < argc; i++)
    argv0[i] = argv[i];

  /* Option rules of the reference (FastK.c:250-319 with ARG_FLAGS / ARG_POSITIVE / ARG_NON_NEGATIVE of
     gene_core.h:37-70): the letters v c p t combine in one argument (-vt, -tp); -t<int>, -k, -T, -M,
     -bc take a whole decimal number or the run ends with the reference's message.  -x and -H are this
     driver's own flags and combine with the reference's. */
  { int flags[128], k, g;
    memset(flags,0,sizeof(flags));
    for (i = j = 1; i < argc; i++)
      if (strncmp(argv[i],"--fk-rank=",10) == 0)
        RANK = atoi(argv[i]+10);
      else if (strncmp(argv[i],"--fk-idfile=",12) == 0)
        ID_FILE = argv[i]+12;
      else if (argv[i][0] == '-')
        { int   isflags = 0;
          char *num = NULL, *what = NULL;
          int  *var = NULL, nonneg = 0;
          switch (argv[i][1])
          { case 'k': num = argv[i]+2; var = &KMER; what = "K-mer length"; break;
            case 'T': num = argv[i]+2; var = &NTHREADS; what = "Number of threads"; break;
            case 'M': num = argv[i]+2; var = &MEM_GB; what = "GB of memory for sorting step"; break;
            case 'G': num = argv[i]+2; var = &NGPUS; what = "Number of GPUs"; break;
            case 'b':
              if (argv[i][2] != 'c')
                { fprintf(stderr,"\n%s: -%s is not a legal optional argument\n",Prog_Name,argv[i]); exit (1); }
              num = argv[i]+3; var = &BC_PREFIX; what = "Bar code prefiex"; nonneg = 1;
              break;
            case 't':
              if (argv[i][2] == '\0' || isalpha((unsigned char) argv[i][2]))
                isflags = 1;
              else
                { num = argv[i]+2; var = &DO_TABLE; what = "Cutoff for k-mer table"; }
              break;
            case 'p':
              if (argv[i][2] == ':')
                { PRO_NAME = argv[i]+3;      /* profiles relative to this table, FastK.c:270-282 */
                  PROFILE  = 1;
                }
              else
                isflags = 1;
              break;
            case 'N': OUT_NAME = argv[i]+2; break;
            case 'P': break;
            default:  isflags = 1; break;
          }
          if (num != NULL)
            { char *eptr;
              long  v = strtol(num,&eptr,10);
              if (*eptr != '\0' || num[0] == '\0')
                { fprintf(stderr,"%s: -%c '%s' argument is not an integer\n",Prog_Name,argv[i][1],argv[i]+2);
                  exit (1);
                }
              if (nonneg ? (v < 0) : (v <= 0))
                { fprintf(stderr,"%s: %s must be %s (%ld)\n",Prog_Name,what,nonneg ? "non-negative" : "positive",v);
                  exit (1);
                }
              *var = (int) v;
            }
          if (isflags)
            for (k = 1; argv[i][k] != '\0'; k++)
              { if (strchr("vcptxH",argv[i][k]) == NULL)
                  { fprintf(stderr,"%s: -%c is an illegal option\n",Prog_Name,argv[i][k]);
                    exit (1);
                  }
                flags[(int) argv[i][k]] = 1;
              }
        }
      else
        argv[j++] = argv[i];
    VERBOSE    = flags['v'];
    COMPRESS   = flags['c'];
    EXACT      = flags['x'];
    HOST_PARSE = flags['H'];
    DEVICE_TEXT = (getenv("FASTK_AMD_DEVICE_TEXT") != NULL && atoi(getenv("FASTK_AMD_DEVICE_TEXT")) > 0);
    if (flags['t']) DO_TABLE = 1;
    if (flags['p']) PROFILE = 1;
    (void) g;
  }
  nfiles = j-1;
  if (KMER > 64 || KMER < 8)         /* the device expansion is built for k-mers of up to four 32-bit words */
    { fprintf(stderr,"%s: K-mer length must be between 8 and 64 in this engine (%d)\n",Prog_Name,KMER);
      exit (1);
    }
  if (nfiles < 1 || KMER <= 0 || NTHREADS <= 0 || DO_TABLE < 0 || BC_PREFIX < 0)
    { fprintf(stderr,"\nUsage: %s [-k<int(40)>] [-t[<int(1)>]] [-p[:<table>[.ktab]]] [-c] [-bc<int>] [-v] [-x] [-N<path_name>]\n",Prog_Name);
      fprintf(stderr,"       %*s [-P<dir>] [-M<int>] [-T<int(4)>] [-G<int(1)>] <source>[.fa|.fasta|.fq|.fastq][.gz]|.sam|.bam ...\n",
              (int) strlen(Prog_Name),"");
      fprintf(stderr,"\n      -G: number of GPUs (one process per GPU, minimizer buckets exchanged over RCCL; k-mer counting only)\n");
      fprintf(stderr,"      -x: reproduce the reference's super-mer cuts, hence its hidden .ktab part boundaries (slower)\n");
      exit (1);
    }

  for (i = 1; i <= nfiles; i++)
    argv[i] = resolve_input(argv[i]);


  if (NGPUS > 1 && (EXACT || PRO_NAME != NULL))
    { fprintf(stderr,"%s: -x and -p:<table> run on one GPU (-G%d takes -t, -p -- also together -- and -M)\n",Prog_Name,NGPUS);
      exit (1);
    }
  if (NGPUS > 1 && RANK < 0)
    return (launch_ranks(argc0,argv0));

  double t_start = now(), t_create = 0., t_created = 0., t_ingest, t_count, t_write;

  fk_default_params(&prm);
  prm.kmer = KMER; prm.table_cutoff = PROFILE ? 1 : DO_TABLE;   /* -p looks every k-mer up */
  prm.nthreads = NTHREADS; prm.bc_prefix = BC_PREFIX;
  prm.exact_parts = EXACT ? (PROFILE ? 2 : 1) : 0;   /* (2: with -p the reference's super-mers stay on the read's strand) */
  if (MEM_GB > 0 && !EXACT)
    { /* bases ~ file bytes (FASTA), half of them (FASTQ), x4 when gzipped; the super-mers (~1 byte
         per base) stay in HBM up to half the budget and spill to host memory beyond, a bucket's
         working set is ~7.5 bytes per base of that bucket */
      double bases = 0., budget = 1e9 * MEM_GB;
      for (i = 1; i <= nfiles; i++)
        { char *r, *d;
          int   q = classify(argv[i],&r,&d);
          FILE *fp = fopen(argv[i],"rb");
          if (q >= 0 && fp != NULL)
            { size_t l = strlen(argv[i]);
              double sz;
              fseek(fp,0,SEEK_END);
              sz = (double) ftell(fp);
              if (l > 3 && strcmp(argv[i]+l-3,".gz") == 0) sz *= 4.;
              bases += (q == 3) ? 2.*sz : (q ? sz/2. : sz);
            }
          if (fp != NULL) fclose(fp);
          if (q >= 0) { free(r); free(d); }
        }
      { /* the library keeps the super-mer records (~1.1 bytes per base) in 3/4 of the budget and spills beyond;
           a bucket's working set (its records twice, its weighted k-mers twice: ~6 bytes per base of the bucket)
           gets a sixteenth, which leaves the rest for the table, its sort and the read buffers */
        double room = budget/16.;
        int    nb   = 1;
        while (nb < 255 && 6.*bases/nb > room)
          nb += 1;
        prm.nbuckets   = nb;
        prm.hbm_budget = (int64_t) budget;
        if (VERBOSE)
          fprintf(stderr,"  -M%d: ~%.2f Gbp of input, %d minimizer bucket(s), reads split in chunks\n",
                  MEM_GB,bases/1e9,nb);
      }
    }
  fk_shard *shard = NULL;
  if (RANK >= 0 && NGPUS > 1)
    { /* one of the ranks of a -G run: its own GPU, world x ROUNDS minimizer buckets (bucket r*world+d goes to
         rank d in exchange round r), its stripe of the reads resident */
      prm.device     = (getenv("FK_RANKS_SHARE_GPU") != NULL) ? 0 : RANK;
      { /* -M<GB> is the budget of every rank: its stripe is then split chunk by chunk as it is read (records beyond
           the budget spill to pinned host memory) and the exchange rounds are as many as it takes for a round's
           records and working set to stay within a sixteenth of it -- the buckets the one-GPU run would use */
        int rounds = ROUNDS;
        if (prm.hbm_budget > 0 && prm.nbuckets > NGPUS*rounds)
          rounds = (prm.nbuckets+NGPUS-1)/NGPUS;
        prm.nbuckets = NGPUS*rounds;
      }
      if (prm.nbuckets > 256)
        prm.nbuckets = NGPUS*(256/NGPUS);
    }
  t_create = now();
  if (fk_create(&prm,&ctx) != FK_OK)
    die(NULL,"fk_create");
  t_created = now();
  if (getenv("FASTK_AMD_DEBUG") != NULL && atoi(getenv("FASTK_AMD_DEBUG")) > 0)     /* per-chunk / per-bucket lines on stderr */
    fk_debug_set(ctx,"verbose",atoi(getenv("FASTK_AMD_DEBUG")));
  /* tests only (the library takes these knobs from processes that set FASTK_AMD_TEST_KNOBS=1): chunk the ingest and
     spill at sizes a fixture reaches */
  if (getenv("FASTK_AMD_CHUNK_BYTES") != NULL && fk_debug_set(ctx,"chunk_bytes",atoll(getenv("FASTK_AMD_CHUNK_BYTES"))) != FK_OK)
    die(ctx,"FASTK_AMD_CHUNK_BYTES");
  if (getenv("FASTK_AMD_SPILL_LIMIT") != NULL && fk_debug_set(ctx,"spill_limit",atoll(getenv("FASTK_AMD_SPILL_LIMIT"))) != FK_OK)
    die(ctx,"FASTK_AMD_SPILL_LIMIT");
  if (EXACT)          /* -x -M<int>: the reference's sort memory (12 GB unless given, FastK.c:235,291), hence its buckets */
    fk_set_sort_memory(ctx,(int64_t) (MEM_GB > 0 ? MEM_GB : 12)*1000000000ll,0.);
  if (RANK >= 0 && NGPUS > 1)
    { char id[128];
      share_unique_id(id);
      if (fk_shard_create(ctx,RANK,NGPUS,id,&shard) != FK_OK)
        die(ctx,"fk_shard_create");
      if (PROFILE && DO_TABLE > 1          /* -t<n> -p: counted with cutoff 1 for the look-ups, the files keep count >= n */
          && fk_shard_set_write_cutoff(shard,DO_TABLE) != FK_OK)
        die(ctx,"fk_shard_set_write_cutoff");
    }

  memset(&feed,0,sizeof(feed));
  feed.ctx   = ctx;
  feed.cap_bytes = BLOCK_BYTES;
  feed.cap_reads = BLOCK_READS;
  feed.bases = malloc(BLOCK_BYTES+16);
  feed.boff  = malloc(sizeof(int32_t)*(BLOCK_READS+2));
  res        = malloc(sizeof(fk_result));
  if (feed.bases == NULL || feed.boff == NULL || res == NULL)
    { fprintf(stderr,"%s: Out of memory\n",Prog_Name); exit (1); }
  feed.boff[0] = 0;

  /* a run takes its reads in one form (fk_push_packed keeps them in two bits per base): the reader threads pack
     the text only when every input is a plain FASTA / FASTQ file */
  all_plain = 1;
  for (i = 1; i <= nfiles; i++)
    { char *r, *d;
      int   q = classify(argv[i],&r,&d);
      if (q < 0 || q > 1 || (strlen(argv[i]) > 3 && strcmp(argv[i]+strlen(argv[i])-3,".gz") == 0))
        all_plain = 0;
      { struct stat sb;                   /* the reader threads map the file and cut it by offset: regular files only
                                             (a pipe or FIFO goes through the sequential scanner) */
        if (stat(argv[i],&sb) != 0 || !S_ISREG(sb.st_mode))
          all_plain = 0;
      }
      if (q >= 0)
        { free(r); free(d); }
    }

  for (i = 1; i <= nfiles; i++)
    { char *r, *d;
      int   q = classify(argv[i],&r,&d);
      if (q < 0)
        { fprintf(stderr,"%s: %s: CRAM and Dazzler inputs are not built in this engine\n",
                  Prog_Name,argv[i]);
          exit (1);
        }
      if (i == 1)
        ftype = q;
      else if (q != ftype)                /* io.c: one input type per run */
        { fprintf(stderr,"%s: All files must be of the same type\n",Prog_Name);
          exit (1);
        }
      if (i == 1)
        { root = r; dir = d; }            /* outputs take the first file's root, FastK.c:402-405 */
      else
        { free(r); free(d); }
      /* profiles need the reads exactly as the reference numbers them: the device FASTQ parser keeps
         one terminated read per record; FASTA text goes through the host scanner */
      if (q == 2)
        scan_sam(&feed,argv[i]);
      else if (q == 3)
        scan_bam(&feed,argv[i]);
      else if (all_plain && !EXACT && BC_PREFIX == 0 && !HOST_PARSE && !COMPRESS && !PROFILE && !DEVICE_TEXT)
        scan_text_packed(&feed,argv[i],q);
      else if (!EXACT && BC_PREFIX == 0 && !HOST_PARSE && (q == 1 || !(COMPRESS || PROFILE))
               && !(NGPUS > 1 && strlen(argv[i]) > 3 && strcmp(argv[i]+strlen(argv[i])-3,".gz") == 0))
        scan_text_on_device(&feed,argv[i],q);
      else
        { if (EXACT && PROFILE && q <= 1 && i == 1)
            input_threads(&feed,argv+1,nfiles,q);
          feed.cur_file = i-1;
          scan_file(&feed,argv[i],q);
        }
    }
  flush_block(&feed,0);
  if (OUT_NAME != NULL)                   /* -N, FastK.c:406-409 */
    { char *slash = strrchr(OUT_NAME,'/');
      free(root); free(dir);
      root = strdup(slash ? slash+1 : OUT_NAME);
      dir  = slash ? strndup(OUT_NAME,(size_t) (slash-OUT_NAME)) : strdup(".");
    }

  t_ingest = now();
  if (PRO_NAME != NULL)
    { /* relative profiles: the counts are those of the given table; only profiles are produced
         (README.md:112-120) */
      char *tdir, *troot;
      uint8_t *recs = NULL;
      int64_t  n = 0, cap = 0;
      int      tk = 0, mv = 0x7fff;
      fk_profiles pr;
      fk_widths   wd;
      fk_get_widths(KMER,&wd);
      split_path(PRO_NAME,".ktab",&tdir,&troot);
      load_table(tdir,troot,wd.kmer_word,&tk,&mv,&recs,&n,&cap,KMER);
      if (DO_TABLE > 0)
        fprintf(stderr,"%s: Warning: -p:%s overides -t option\n",Prog_Name,PRO_NAME);
      if (fk_set_table(ctx,recs,n) != FK_OK)
        die(ctx,"fk_set_table");
      if (fk_make_profiles(ctx,NULL,0,&pr) != FK_OK)
        die(ctx,"fk_make_profiles");
      if (fk_write_prof(&pr,KMER,NTHREADS,dir,root) != FK_OK)
        die(ctx,"writing .prof");
      if (VERBOSE)
        fprintf(stderr,"  Profiles of %lld reads relative to %lld %d-mers of %s in %lld bytes\n",
                (long long) pr.nreads,(long long) n,KMER,PRO_NAME,(long long) pr.nbytes);
      fk_destroy(ctx);
      exit (0);
    }
  if (shard != NULL)
    { /* exchange + count + all-reduce, then the second exchange and this rank's share of the files */
      int nparts = ((NTHREADS+NGPUS-1)/NGPUS)*NGPUS;
      if (fk_shard_count(shard,res) != FK_OK)
        die(ctx,"fk_shard_count");
      t_count = now();
      if (DO_TABLE > 0)
        { if (fk_shard_write(shard,res,nparts,dir,root) != FK_OK)
            die(ctx,"fk_shard_write");
        }
      else if (RANK == 0)                 /* -p without -t: the histogram only (the table stays sharded in HBM) */
        { snprintf(name,sizeof(name),"%s/%s.hist",dir,root);
          if (fk_write_hist(res,KMER,name) != FK_OK)
            die(ctx,"writing .hist");
        }
      t_write = now();
      if (PROFILE)
        { /* The profile pass (count.c:639-1181 + merge.c in the reference): every rank takes a contiguous range of the
             data set's reads -- read n r / G .. n (r+1) / G, in file order, whatever the stripes of the counting pass
             were -- scans the input for them (the others are parsed and dropped) and hands them block by block to
             fk_shard_profiles: their k-mers are looked up on the ranks that own them.  The calls are collective:
             a rank that has run out of blocks keeps answering with empty ones until every rank has. */
          Feeder pf;
          fk_profiles pr;
          int64_t tot = feed.striped_rds;   /* reads of the data set: the stripes add up; what the host parser walked
                                               (gzipped files, -c) every rank saw whole */
          int     active = 1;
          double  t0 = now();
          if (fk_shard_sum_i64(shard,&tot,1) != FK_OK)
            die(ctx,"fk_shard_sum_i64");
          tot += feed.totrds-feed.striped_rds;
          memset(&pf,0,sizeof(pf));
          pf.ctx = ctx;
          pf.to_profiles = 2;
          pf.shard = shard;
          pf.rd_lo = tot*RANK/NGPUS;
          pf.rd_hi = tot*(RANK+1)/NGPUS;
          if (pf.rd_hi == 0) pf.rd_hi = -1;  /* (no reads at all: rd_hi > 0 switches the filter on) */
          pf.cap_bytes = 256 << 20;
          pf.cap_reads = 4 << 20;
          pf.bases = malloc((size_t) pf.cap_bytes+16);
          pf.boff  = malloc(sizeof(int32_t)*((size_t) pf.cap_reads+2));
          if (pf.bases == NULL || pf.boff == NULL || fk_device_alloc(ctx,pf.cap_bytes+64,&pf.d_piece) != FK_OK)
            { fprintf(stderr,"%s: Out of memory\n",Prog_Name); exit (1); }
          pf.boff[0] = 0;
          if (pf.rd_hi > pf.rd_lo)
            for (i = 1; i <= nfiles && !range_done(&pf); i++)
              { if (ftype == 2) scan_sam(&pf,argv[i]);
                else if (ftype == 3) scan_bam(&pf,argv[i]);
                else scan_file(&pf,argv[i],ftype);
              }
          flush_block(&pf,0);
          while (active > 0)              /* this rank is through: empty blocks until all are */
            if (fk_shard_profiles(shard,NULL,0,&pr,&active) != FK_OK)
              die(ctx,"fk_shard_profiles");
          if (pf.poffs == NULL)
            { pf.poffs = malloc(sizeof(int64_t)); pf.poffs[0] = 0; }
          memset(&pr,0,sizeof(pr));
          pr.nreads = pf.preads; pr.nbytes = pf.pbytes; pr.data = pf.pdata; pr.offsets = pf.poffs;
          if (pr.nreads != ((pf.rd_hi > 0) ? pf.rd_hi-pf.rd_lo : 0))
            { fprintf(stderr,"%s: rank %d made %lld profiles for its %lld reads\n",Prog_Name,RANK,(long long) pr.nreads,
                      (long long) (pf.rd_hi-pf.rd_lo));
              exit (1);
            }
          if (fk_shard_write_prof(shard,&pr,KMER,nparts,dir,root) != FK_OK)
            die(ctx,"fk_shard_write_prof");
          if (VERBOSE)
            fprintf(stderr,"  rank %d: profiles of reads %lld .. %lld in %lld bytes (%.3f s)\n",RANK,(long long) pf.rd_lo,
                    (long long) pf.rd_hi-1,(long long) pr.nbytes,now()-t0);
          fk_device_free(ctx,pf.d_piece);
          free(pf.bases); free(pf.boff);
        }
      if (VERBOSE && RANK == 0)
        { fprintf(stderr,"\n  %d ranks: %lld %d-mers in %lld super-mers, %lld weighted k-mers, %lld distinct, %lld in the table (%d parts)\n",
                  NGPUS,(long long) res->ninst,KMER,(long long) res->nsuper,(long long) res->nweighted,
                  (long long) res->ndistinct,(long long) res->ntable,nparts);
          fprintf(stderr,"  Wall s (rank 0): start-up + ingest %.3f  exchange + count %.3f  table exchange + write %.3f\n",
                  t_ingest-t_start,t_count-t_ingest,t_write-t_count);
        }
      fk_shard_destroy(shard);
      fk_destroy(ctx);
      exit (0);
    }
  /* without profiles the table is only written: it stays in HBM and the part writers fetch it piece by piece */
  KEEP_TABLE = (!PROFILE && DO_TABLE > 0);
  if ((KEEP_TABLE ? fk_finish_device(ctx,res) : fk_finish(ctx,res)) != FK_OK)
    die(ctx,"fk_finish");
  t_count = now();

  if (PROFILE)
    { fk_profiles pr;
      double t0 = now();
      if (MEM_GB > 0 && !EXACT)
        { /* the counting pass dropped the reads chunk by chunk: scan the input again and look the reads
             up piece by piece in the table the pass left in HBM */
          Feeder pf;
          memset(&pf,0,sizeof(pf));
          pf.ctx = ctx;
          pf.to_profiles = 1;
          pf.cap_bytes = 256 << 20;
          pf.cap_reads = 4 << 20;
          pf.bases = malloc((size_t) pf.cap_bytes+16);
          pf.boff  = malloc(sizeof(int32_t)*((size_t) pf.cap_reads+2));
          if (pf.bases == NULL || pf.boff == NULL || fk_device_alloc(ctx,pf.cap_bytes+64,&pf.d_piece) != FK_OK)
            { fprintf(stderr,"%s: Out of memory\n",Prog_Name); exit (1); }
          pf.boff[0] = 0;
          for (i = 1; i <= nfiles; i++)
            { if (ftype == 2) scan_sam(&pf,argv[i]);
              else if (ftype == 3) scan_bam(&pf,argv[i]);
              else scan_file(&pf,argv[i],ftype);
            }
          flush_block(&pf,0);
          if (pf.poffs == NULL)
            { pf.poffs = malloc(sizeof(int64_t)); pf.poffs[0] = 0; }
          memset(&pr,0,sizeof(pr));
          pr.nreads = pf.preads; pr.nbytes = pf.pbytes; pr.data = pf.pdata; pr.offsets = pf.poffs;
          fk_device_free(ctx,pf.d_piece);
          free(pf.bases); free(pf.boff);
        }
      else if (fk_make_profiles(ctx,NULL,0,&pr) != FK_OK)
        die(ctx,"fk_make_profiles");
      /* -x -p on FASTA / FASTQ files: one part per input thread of the reference (input_threads), else one per -T thread */
      if (fk_write_prof(&pr,KMER,(feed.nstarts > 0 && pr.nsplit == feed.nstarts) ? pr.nsplit : NTHREADS,dir,root) != FK_OK)
        die(ctx,"writing .prof");
      if (VERBOSE)
        fprintf(stderr,"  Profiles of %lld reads in %lld bytes (%.3f s)\n",(long long) pr.nreads,
                (long long) pr.nbytes,now()-t0);
      if (DO_TABLE > 1)                   /* the table was built with cutoff 1: keep count >= -t */
        { const int kw = ((2*KMER+7)>>3) + 2;
          const uint8_t *t = res->table;
          uint8_t *keep;
          int64_t  n = 0, x;
          for (x = 0; x < res->ntable; x++)
            if ((t[x*kw+kw-2] | (t[x*kw+kw-1] << 8)) >= DO_TABLE)
              n += 1;
          keep = malloc((size_t) (n > 0 ? n : 1)*kw);
          if (keep == NULL)
            { fprintf(stderr,"%s: Out of memory\n",Prog_Name); exit (1); }
          for (n = x = 0; x < res->ntable; x++)
            if ((t[x*kw+kw-2] | (t[x*kw+kw-1] << 8)) >= DO_TABLE)
              memcpy(keep+(n++)*kw,t+x*kw,(size_t) kw);
          res->table  = keep;             /* freed at exit */
          res->ntable = n;
        }
    }

  if (VERBOSE)
    { fprintf(stderr,"\n  There are %lld reads totalling %lld bps\n",(long long) feed.totrds,(long long) feed.totbps);
      fprintf(stderr,"  %lld %d-mers in %lld super-mers (%lld distinct), %lld weighted k-mers, %lld distinct\n",
              (long long) res->ninst,KMER,(long long) res->nsuper,(long long) res->ndistinct_super,
              (long long) res->nweighted,(long long) res->ndistinct);
      fprintf(stderr,"  Device ms: split %.2f  sort %.2f  expand %.2f  sort %.2f  count %.2f  total %.2f\n",
              res->ms_split,res->ms_sort_super,res->ms_expand,res->ms_sort_kmer,res->ms_count,res->ms_total);
      if (DO_TABLE > 0)
        fprintf(stderr,"  There are %lld %d-mers that occur %d-or-more times\n",(long long) res->ntable,KMER,DO_TABLE);
    }

  /* the device memory (280 GB at the size of a human genome) goes back while the files are written: the driver
     takes seconds over it, about as long as the writers take over a 36 GB table */
  { pthread_t rel;
    int       relt = (pthread_create(&rel,NULL,release_thread,ctx) == 0);

    /* (a writer that fails -- a full disk -- must not tear the context down under the release thread: join first) */
    snprintf(name,sizeof(name),"%s/%s.hist",dir,root);
    if (fk_write_hist(res,KMER,name) != FK_OK)
      { if (relt) pthread_join(rel,NULL);
        die(ctx,"writing .hist");
      }
    if (DO_TABLE > 0 && (KEEP_TABLE ? fk_write_ktab_device(ctx,res,NTHREADS,dir,root)
                                    : fk_write_ktab(res,KMER,DO_TABLE,NTHREADS,dir,root)) != FK_OK)
      { if (relt) pthread_join(rel,NULL);
        die(ctx,"writing .ktab");
      }
    t_write = now();
    if (relt)
      pthread_join(rel,NULL);
    /* the table's own buffer (36 GB at the size of a human genome) goes with the process: hipFree takes 0.6 s over
       it, the kernel at exit next to nothing (three runs each: 4.5 s against 3.9-4.1 s in all) */
  }
  if (VERBOSE)
    { fprintf(stderr,"  Wall s: start-up + ingest %.3f  count + table fetch %.3f  write %.3f\n",
              t_ingest-t_start,t_count-t_ingest,t_write-t_count);
      fprintf(stderr,"  Wall s: fk_create %.3f (began %.3f after start)  clean-up %.3f  (process start to here %.3f)\n",
              t_created-t_create,t_create-t_start,now()-t_write,now()-t_start);
    }
  /* every output file is closed and all device memory but the table's buffer is back; what is left (that buffer, the
     context, pinned staging) goes with the process.  fk_destroy here would take seconds (3.3 s measured at configs[2]). */
  fflush(stdout);
  fflush(stderr);
  if (getenv("FASTK_AMD_ATEXIT") != NULL)       /* under a profiler: its exit handlers write the trace */
    exit (0);
  _exit (0);
}
