/* fk_oracle.c -- CPU restatement of FastK's split / sort / count hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see fk_oracle.h).  Written from SURVEY.md section 3/8 and from
 * reading the reference; every function cites the reference lines whose behaviour it restates.
 * Parity: PINNED against the reference build in oracle/_ref and tests/golden fixtures (also on reads full of minimizer
 * ties and, orc_set_profile_mode, against runs of the reference with -p: tests/test_oracle_vs_reference.py).
 *
 * Deliberate simplifications (none changes any output byte):
 *   - the bit-stuffed ".T" spill encoding (split.c:828-989, count.c:80-149) is skipped: the
 *     distributor writes the fixed-width SMER_WORD records that supermer_list_thread
 *     (count.c:165-313) would unpack; byte 0 keeps the first four bases instead of the
 *     reference's 0/1 run-head flag (the reference pre-buckets on that byte and sorts from
 *     byte 1, which is the same total order);
 *   - one bucket (NPARTS=1): bucket assignment never changes .hist or the .ktab canonical
 *     stream, and at NPARTS=1 the reference trie stays at PAD=0.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <sys/stat.h>
#include <fcntl.h>
#include <unistd.h>

#include "fk_oracle.h"
#include "../include/fk_synth.h"

/* ---------------------------------------------------------------------------------------
 * widths                                                   FastK.c:417,446-468  split.c:617-628
 */
void orc_params_init(orc_params *P, int kmer, int pad)
{ int v;

  P->kmer       = kmer;
  P->min_len    = 5 + pad;
  P->max_super  = kmer - (P->min_len-1);
  P->smer       = P->max_super + kmer - 1;
  P->slen_bits  = 0;
  for (v = P->max_super; v > 0; v >>= 1)
    P->slen_bits += 1;
  P->slen_bytes = (P->slen_bits+7) >> 3;
  P->kmer_bytes = (2*kmer+7) >> 3;
  P->smer_bytes = (2*P->smer+7) >> 3;
  P->smer_word  = P->smer_bytes + P->slen_bytes;
  P->kmer_word  = P->kmer_bytes + 2;
  for (v = 0; v < 4; v++)
    P->tran[v] = v;
}

/* ---------------------------------------------------------------------------------------
 * base mapping by frequency rank                          split.c:95-112 (counts), 529-575
 */
void orc_train_tran(orc_params *P, const char *bases, const int64_t *boff, int64_t nreads,
                    int nthreads)
{ int64_t freq[256], f4[4];
  int64_t i, stripe0;
  int     a, b;

  /* Per-thread byte counts over read stripes [nreads*t/T, nreads*(t+1)/T) (split.c:95-112,
     507-512) are summed INTO thread 0's own vector starting from j = 0 (split.c:536-539), so
     stripe 0 is counted twice.  The ranking below therefore depends on -T exactly as the
     reference's does. */
  memset(freq,0,sizeof(freq));
  for (i = boff[0]; i < boff[nreads]; i++)
    freq[(uint8_t) bases[i]] += 1;
  stripe0 = (nthreads > 1) ? (nreads * 1) / nthreads : nreads;
  for (i = boff[0]; i < boff[stripe0]; i++)
    freq[(uint8_t) bases[i]] += 1;
  f4[0] = freq['a'] + freq['A'];
  f4[1] = freq['c'] + freq['C'];
  f4[2] = freq['g'] + freq['G'];
  f4[3] = freq['t'] + freq['T'];
  for (a = 0; a < 4; a++)
    { int rank = 0;
      for (b = 0; b < 4; b++)
        if (f4[b] < f4[a] || (f4[b] == f4[a] && b < a))
          rank += 1;
      P->tran[a] = rank;
    }
}

/* ---------------------------------------------------------------------------------------
 * super-mer distribution                                 split.c:1016-1393, count.c:165-313
 */
static const int8_t *code_table(void)
{ static int8_t tab[256];
  static int    init = 0;
  if (!init)
    { memset(tab,4,sizeof(tab));
      tab['a'] = tab['A'] = 0;
      tab['c'] = tab['C'] = 1;
      tab['g'] = tab['G'] = 2;
      tab['t'] = tab['T'] = 3;
      init = 1;
    }
  return (tab);
}

typedef struct
  { const orc_params *P;
    uint8_t **out;
    int64_t  *nout, *cap;
    int64_t   kmers;
  } Emitter;

/* pack n k-mers' worth of bases starting at s (flip => reverse complement) into one record:
   [SMER_BYTES packed 2-bit, MSB first, zero padded][n-1 big-endian in SLEN_BYTES]
   split.c:864-989 (Stuff_Seq), count.c:226-251 (record fill) */
static void emit_supermer(Emitter *E, const char *s, int n, int flip)
{ const orc_params *P = E->P;
  const int8_t     *code = code_table();
  int      len = n-1 + P->kmer;
  uint8_t *rec;
  int      i, v;

  if (*E->nout >= *E->cap)
    { *E->cap = (*E->cap)*2 + 1024;
      *E->out = realloc(*E->out,(size_t) (*E->cap) * P->smer_word);
      if (*E->out == NULL)
        { fprintf(stderr,"fk_oracle: out of memory\n"); exit (1); }
    }
  rec = *E->out + (*E->nout) * (int64_t) P->smer_word;
  memset(rec,0,P->smer_word);
  for (i = 0; i < len; i++)
    { if (flip)
        v = 3 - code[(uint8_t) s[len-1-i]];
      else
        v = code[(uint8_t) s[i]];
      rec[i>>2] |= (uint8_t) (v << (6 - 2*(i&3)));
    }
  v = n-1;
  for (i = P->slen_bytes-1; i >= 0; i--)
    { rec[P->smer_bytes+i] = (uint8_t) (v & 0xff);
      v >>= 8;
    }
  *E->nout += 1;
  E->kmers += n;
}

static int Dist_Profile = 0;            /* a run with -p: the super-mers keep the read's strand (split.c:1245: under
                                           DO_PROFILE Stuff_Seq is called with flip 0), so a super-mer and its reverse
                                           complement are two records and the weighted k-mer list -- whose first-byte
                                           census cuts the .ktab parts, count.c:1560-1565 -- is another one */
void orc_set_profile_mode(int on)
{ Dist_Profile = (on != 0); }

static const orc_scheme *Dist_Scheme;   /* set by orc_fastk_parts around its distribute call */
static uint8_t **Dist_Bucket;           /* bucket of every record emitted (parallel to *out), realloc'd */
static int64_t  *Dist_Bcap;

int64_t orc_distribute_block(const orc_params *P, const char *bases, const int64_t *boff,
                             int64_t nreads, int bc_prefix,
                             uint8_t **out, int64_t *nout, int64_t *cap)
{ const int8_t *code = code_table();
  const int K   = P->kmer;
  const int ML1 = P->min_len-1;
  const int MS  = P->max_super;
  const uint64_t ptot = 1ull << (2*P->min_len);
  const uint64_t pmsk = ptot-1;

  uint64_t fw[256], rc[256];     /* Tran / Cran of split.c:560-575, 630-639 */
  uint64_t *ring;
  uint8_t  *rflp;
  int       rlen, rmsk;
  Emitter   E;
  int64_t   r;
  int       x;

  for (x = 0; x < 256; x++)
    { fw[x] = (uint64_t) P->tran[0];
      rc[x] = ((uint64_t) P->tran[3]) << (2*ML1);
    }
  fw['a'] = fw['A'] = P->tran[0];  rc['a'] = rc['A'] = ((uint64_t) P->tran[3]) << (2*ML1);
  fw['c'] = fw['C'] = P->tran[1];  rc['c'] = rc['C'] = ((uint64_t) P->tran[2]) << (2*ML1);
  fw['g'] = fw['G'] = P->tran[2];  rc['g'] = rc['G'] = ((uint64_t) P->tran[1]) << (2*ML1);
  fw['t'] = fw['T'] = P->tran[3];  rc['t'] = rc['T'] = ((uint64_t) P->tran[0]) << (2*ML1);

  rlen = 1;                       /* MOD_LEN, FastK.c:446-450 */
  while (rlen < K)
    rlen <<= 1;
  rlen <<= 1;
  rmsk = rlen-1;
  ring = malloc(sizeof(uint64_t)*rlen);
  rflp = malloc(rlen);

  E.P = P; E.out = out; E.nout = nout; E.cap = cap; E.kmers = 0;

  for (r = 0; r < nreads; r++)
    { const char *s = bases + boff[r] + bc_prefix;
      int  q = (int) (boff[r+1] - boff[r]) - 1 - bc_prefix;   /* split.c:1077-1079 */
      uint64_t c, u, mp, mc;
      int  m, p, last;
      int  ilo, ihi;      /* current invalid k-mer-end interval [ilo,ihi)   (nfst,nlst) */
      int  plo, phi;      /* previous one                                    (pfst,plst) */
      int  closing, force, done;

      if (q < K)
        continue;

      m = 0; mc = ptot; c = u = 0; mp = 0;
      ilo = ihi = phi = -1; plo = 0;
      for (p = 0; p < K; p++)                               /* split.c:1096-1134 */
        { x = (uint8_t) s[p];
          c = ((c << 2) | fw[x]) & pmsk;
          u = (u >> 2) | rc[x];
          if (p >= ML1)
            { rflp[p] = (u < c);
              mp = rflp[p] ? u : c;
              ring[p] = mp;
              if (mp < mc)
                { m = p; mc = mp; }
            }
          if (code[x] >= 4)
            { if (p > ihi)
                ilo = K-1;
              ihi = p+K;
            }
        }

      last = K-1;
      done = 0;
      for (p = K; !done; p++)                               /* split.c:1136-1347 */
        { if (p < q)
            { x = (uint8_t) s[p];
              c = ((c << 2) | fw[x]) & pmsk;
              u = (u >> 2) | rc[x];
              rflp[p&rmsk] = (u < c);
              mp = rflp[p&rmsk] ? u : c;
              ring[p&rmsk] = mp;
              force   = (p-m >= MS);
              closing = (force || mp < mc);
            }
          else                                               /* end-of-read flush :1342-1347 */
            { if (ihi == q)
                break;
              x = 'a';
              mp = mc;
              force = closing = 1;
              done = 1;
            }

          if (closing)
            { int n;

              if (ihi >= last)                               /* split.c:1167-1232 */
                { if (ihi <= p)
                    { last = ihi;
                      ihi  = -1;
                      n = p-last;
                    }
                  else
                    { if (phi > last)
                        { last = phi;
                          phi  = -1;
                        }
                      n = ilo-last;
                    }
                }
              else
                n = p-last;

              if (n > 0)                                     /* split.c:1234-1302 */
                { if (Dist_Scheme != NULL)                   /* the bucket of its minimizer, split.c:1149-1157 */
                    { const orc_scheme *S = Dist_Scheme;
                      int o = (int) (mc >> (2*S->pad)), b = S->part[o], y = 2*S->pad-2;
                      while (b < 0)
                        { o = (int) ((mc >> y) & 3) - b;
                          b = S->part[o];
                          y -= 2;
                        }
                      if (*nout >= *Dist_Bcap)
                        { *Dist_Bcap = (*nout)*2 + 1024;
                          *Dist_Bucket = realloc(*Dist_Bucket,(size_t) *Dist_Bcap);
                        }
                      (*Dist_Bucket)[*nout] = (uint8_t) b;
                    }
                  emit_supermer(&E,s+(last-(K-1)),n,Dist_Profile ? 0 : rflp[m&rmsk]);
                }

              if (done)
                break;
              if (force)                                     /* split.c:1304-1320 */
                { int j;
                  m += 1;
                  mc = ring[m&rmsk];
                  for (j = m+1; j <= p; j++)
                    if (ring[j&rmsk] <= mc)
                      { m = j; mc = ring[j&rmsk]; }
                }
              else
                { m = p; mc = mp; }
              last = p;
            }

          if (code[x] >= 4)                                  /* split.c:1323-1330 */
            { if (p > ihi)
                { plo = ilo; phi = ihi; ilo = p; }
              ihi = p+K;
            }
        }
      (void) plo;
    }

  free(rflp);
  free(ring);
  return (E.kmers);
}

/* ---------------------------------------------------------------------------------------
 * MSD radix sort                                     MSDsort.c:129-261 (radix), 105-127 (small)
 */
static void small_sort(uint8_t *a, int64_t n, int rsize, int d, int ksize)
{ uint8_t tmp[256];
  int64_t i, j;
  int     cmp = ksize-d;

  for (i = 1; i < n; i++)
    { if (memcmp(a+(i-1)*rsize+d,a+i*rsize+d,cmp) <= 0)
        continue;
      memcpy(tmp,a+i*rsize,rsize);
      for (j = i-1; j >= 0 && memcmp(a+j*rsize+d,tmp+d,cmp) > 0; j--)
        memcpy(a+(j+1)*rsize,a+j*rsize,rsize);
      memcpy(a+(j+1)*rsize,tmp,rsize);
    }
}

static void msd_rec(uint8_t *a, int64_t n, int rsize, int d, int ksize)
{ int64_t cnt[256], beg[256], end[256];
  uint8_t tmp[256], hold[256];
  int64_t i;
  int     x;

  while (1)
    { if (d >= ksize || n <= 1)
        return;
      if (n <= 15)                       /* THR0, MSDsort.c:32,245-247 */
        { small_sort(a,n,rsize,d,ksize);
          return;
        }
      x = a[d];                          /* skip constant digits, MSDsort.c:144-158 */
      for (i = 1; i < n; i++)
        if (a[i*rsize+d] != x)
          break;
      if (i < n)
        break;
      d += 1;
    }

  memset(cnt,0,sizeof(cnt));
  for (i = 0; i < n; i++)
    cnt[a[i*rsize+d]] += 1;
  { int64_t o = 0;
    for (x = 0; x < 256; x++)
      { beg[x] = o; o += cnt[x]; end[x] = o; }
  }
  for (x = 0; x < 256; x++)              /* in-place cycle permutation, MSDsort.c:201-240 */
    while (beg[x] < end[x])
      { int t = a[beg[x]*rsize+d];
        if (t == x)
          { beg[x] += 1; continue; }
        memcpy(hold,a+beg[x]*rsize,rsize);
        while (t != x)
          { int64_t dst = beg[t]++;
            int     nt;
            memcpy(tmp,a+dst*rsize,rsize);
            memcpy(a+dst*rsize,hold,rsize);
            memcpy(hold,tmp,rsize);
            nt = hold[d];
            t  = nt;
          }
        memcpy(a+beg[x]*rsize,hold,rsize);
        beg[x] += 1;
      }
  { int64_t o = 0;
    for (x = 0; x < 256; x++)
      { if (cnt[x] > 1)
          msd_rec(a+o*rsize,cnt[x],rsize,d+1,ksize);
        o += cnt[x];
      }
  }
}

void orc_msd_sort(uint8_t *array, int64_t n, int rsize, int ksize)
{ if (rsize > 256)
    { fprintf(stderr,"fk_oracle: record too wide\n"); exit (1); }
  msd_rec(array,n,rsize,0,ksize);
}

/* ---------------------------------------------------------------------------------------
 * LSD radix sort                                                       LSDsort.c:55-94,115-271
 */
void *orc_lsd_sort(int64_t n, void *src, void *trg, int rsize, const int *bytes)
{ uint8_t *s = (uint8_t *) src;
  uint8_t *t = (uint8_t *) trg;
  int      b;

  for (b = 0; bytes[b] >= 0; b++)
    { int64_t ptr[256], o;
      int64_t i;
      int     x;
      uint8_t *w;

      memset(ptr,0,sizeof(ptr));
      for (i = 0; i < n; i++)
        ptr[s[i*rsize+bytes[b]]] += 1;
      o = 0;
      for (x = 0; x < 256; x++)
        { int64_t c = ptr[x]; ptr[x] = o; o += c; }
      for (i = 0; i < n; i++)
        { x = s[i*rsize+bytes[b]];
          memcpy(t+(ptr[x]++)*rsize,s+i*rsize,rsize);
        }
      w = s; s = t; t = w;
    }
  return ((void *) s);
}

/* ---------------------------------------------------------------------------------------
 * weighted k-mer list                                                      count.c:339-542
 */
int64_t orc_kmer_list(const orc_params *P, const uint8_t *smers, int64_t nsmers,
                      uint8_t **out, int64_t *overflow, int64_t *ndistinct)
{ const int K  = P->kmer;
  const int SW = P->smer_word;
  const int KB = P->kmer_bytes;
  const int KW = P->kmer_word;
  int64_t   i, j, w, tot, nd;
  uint8_t  *kl;
  uint8_t   fb[64], rb[64];
  int8_t    base[1024];

  *overflow = 0;

  tot = 0; nd = 0;                         /* size of the list = sum of khist, MSDsort.c:381-456 */
  for (i = 0; i < nsmers; i = j)
    { int sln = 0, b;
      for (b = 0; b < P->slen_bytes; b++)
        sln = (sln << 8) | smers[i*SW+P->smer_bytes+b];
      for (j = i+1; j < nsmers && memcmp(smers+i*SW,smers+j*SW,SW) == 0; j++)
        ;
      tot += sln+1;
      nd  += 1;
    }
  *ndistinct = nd;
  kl = malloc((size_t) (tot+1)*KW);

  w = 0;
  for (i = 0; i < nsmers; i = j)
    { int     sln = 0, b, o, len;
      int64_t ct;

      for (b = 0; b < P->slen_bytes; b++)
        sln = (sln << 8) | smers[i*SW+P->smer_bytes+b];
      for (j = i+1; j < nsmers && memcmp(smers+i*SW,smers+j*SW,SW) == 0; j++)
        ;
      ct = j-i;
      if (ct >= 0x8000)                    /* count.c:455-458 */
        { *overflow += (ct-0x7fff)*(sln+1);
          ct = 0x7fff;
        }
      len = sln+K;
      for (b = 0; b < len; b++)
        base[b] = (smers[i*SW+(b>>2)] >> (6-2*(b&3))) & 3;

      for (o = 0; o <= sln; o++)           /* count.c:468-523 */
        { uint8_t *rec = kl + w*KW;
          const uint8_t *can;

          memset(fb,0,KB);
          memset(rb,0,KB);
          for (b = 0; b < K; b++)
            { fb[b>>2] |= (uint8_t) (base[o+b] << (6-2*(b&3)));
              rb[b>>2] |= (uint8_t) ((3-base[o+K-1-b]) << (6-2*(b&3)));
            }
          can = (memcmp(fb,rb,KB) < 0) ? fb : rb;
          memcpy(rec,can,KB);
          rec[KB]   = (uint8_t) (ct & 0xff);      /* native little-endian uint16, count.c:512 */
          rec[KB+1] = (uint8_t) (ct >> 8);
          w += 1;
        }
    }
  *out = kl;
  return (w);
}

/* ---------------------------------------------------------------------------------------
 * histogram + table                        MSDsort.c:491-509 (hist_kmers), count.c:564-616
 */
void orc_count_sorted(const orc_params *P, const uint8_t *kmers, int64_t nk, int cutoff,
                      orc_result *R)
{ const int KB = P->kmer_bytes;
  const int KW = P->kmer_word;
  int64_t   i, j, nt, cap;

  cap = 1024; nt = 0;
  R->table = (cutoff > 0) ? malloc((size_t) cap*KW) : NULL;
  for (i = 0; i < nk; i = j)
    { int64_t cnt = 0;
      for (j = i; j < nk && memcmp(kmers+i*KW,kmers+j*KW,KB) == 0; j++)
        cnt += kmers[j*KW+KB] | (kmers[j*KW+KB+1] << 8);
      R->ndistinct += 1;
      if (cnt >= 0x7fff)
        { R->hist[0x7fff] += 1;
          R->max_inst += cnt;
          cnt = 0x7fff;
        }
      else
        R->hist[cnt] += 1;
      if (cutoff > 0 && cnt >= cutoff)
        { uint8_t *rec;
          if (nt >= cap)
            { cap *= 2;
              R->table = realloc(R->table,(size_t) cap*KW);
            }
          rec = R->table + nt*KW;
          memcpy(rec,kmers+i*KW,KB);
          rec[KB]   = (uint8_t) (cnt & 0xff);
          rec[KB+1] = (uint8_t) (cnt >> 8);
          nt += 1;
        }
    }
  R->ntable = nt;
}

/* ---------------------------------------------------------------------------------------
 * the whole path                                       FastK.c:491-540 -> count.c:1202-1914
 */
int orc_fastk(const orc_params *P, const char *bases, const int64_t *boff, int64_t nreads,
              int bc_prefix, int cutoff, orc_result *R)
{ uint8_t *smers = NULL, *kl = NULL;
  int64_t  ns = 0, cap = 0, nw, ovf, nd, i;

  memset(R,0,sizeof(*R));
  R->ninst  = orc_distribute_block(P,bases,boff,nreads,bc_prefix,&smers,&ns,&cap);
  R->nsuper = ns;
  orc_msd_sort(smers,ns,P->smer_word,P->smer_word);                    /* count.c:1459 */
  nw = orc_kmer_list(P,smers,ns,&kl,&ovf,&nd);
  R->ndistinct_super = nd;
  R->nweighted = nw;
  for (i = 0; i < nw; i++)
    R->wfirst[kl[i*P->kmer_word]] += 1;
  orc_msd_sort(kl,nw,P->kmer_word,P->kmer_bytes);                      /* count.c:1539 */
  orc_count_sorted(P,kl,nw,cutoff,R);
  R->max_inst += ovf;                                                  /* count.c:1551 */
  free(kl);
  free(smers);
  return (0);
}

/* ---------------------------------------------------------------------------------------
 * the bucket scheme                   Determine_Scheme split.c:617-766 (loop), 116-270 (census),
 *                                     437-472 (refine_tree), 289-381 (assign_pieces)
 * Trained on the first block (Get_First_Block, io.c:2606-2630: the caller passes its reads).  The
 * trainer's super-mer rule is not Distribute_Block's: every byte that is not acgt counts as 'a',
 * the rescan after a forced cut takes the FIRST smallest value (strict <), a read's last super-mer
 * is not counted.  Leaves are dealt with drand48, which the reference never seeds: in glibc that is
 * X' = 0x5DEECE66D X + 0xB mod 2^48 from X = 0 (the library's state starts zeroed), value X'/2^48.
 */
static void scheme_census(const orc_params *P0, int pad, const char *bases, const int64_t *boff, int64_t nreads,
                          int bc_prefix, const int64_t *trie, int64_t *cnt)
{ const int K = P0->kmer, PL = 5+pad, PL1 = PL-1, MS = K-PL1;
  const uint64_t ptot = 1ull << (2*PL), pmsk = ptot-1;
  uint64_t fw[256], rc[256], *ring;
  int      rlen, rmsk, x;
  int64_t  r;

  for (x = 0; x < 256; x++)
    { fw[x] = (uint64_t) P0->tran[0];
      rc[x] = ((uint64_t) P0->tran[3]) << (2*PL1);
    }
  fw['c'] = fw['C'] = P0->tran[1];  rc['c'] = rc['C'] = ((uint64_t) P0->tran[2]) << (2*PL1);
  fw['g'] = fw['G'] = P0->tran[2];  rc['g'] = rc['G'] = ((uint64_t) P0->tran[1]) << (2*PL1);
  fw['t'] = fw['T'] = P0->tran[3];  rc['t'] = rc['T'] = ((uint64_t) P0->tran[0]) << (2*PL1);
  rlen = 1;
  while (rlen < K) rlen <<= 1;
  rlen <<= 1;
  rmsk = rlen-1;
  ring = malloc(sizeof(uint64_t)*rlen);
  for (r = 0; r < nreads; r++)
    { const char *s = bases + boff[r] + bc_prefix;
      int  q = (int) (boff[r+1] - boff[r]) - 1 - bc_prefix;
      uint64_t c = 0, u = 0, mp, mc = ptot;
      int  m = 0, p, last, n;

      if (q < K)
        continue;
      for (p = 0; p < K; p++)
        { x = (uint8_t) s[p];
          c = ((c << 2) | fw[x]) & pmsk;
          u = (u >> 2) | rc[x];
          if (p >= PL1)
            { mp = (u < c) ? u : c;
              ring[p&rmsk] = mp;
              if (mp < mc)
                { m = p; mc = mp; }
            }
        }
      last = K-1;
      for (p = K; p < q; p++)
        { int force;
          x = (uint8_t) s[p];
          c = ((c << 2) | fw[x]) & pmsk;
          u = (u >> 2) | rc[x];
          mp = (u < c) ? u : c;
          ring[p&rmsk] = mp;
          force = (p-m >= MS);
          if (force || mp < mc)
            { int64_t o = (int64_t) (mc >> (2*pad)), v = trie[o];
              int b = 2*pad-2;
              while (v < 0)
                { o = (int64_t) ((mc >> b) & 3) - v;
                  v = trie[o];
                  b -= 2;
                }
              cnt[o] += p-last;
              if (force)
                { m += 1;
                  mc = ring[m&rmsk];
                  for (n = m+1; n <= p; n++)
                    if (ring[n&rmsk] < mc)
                      { m = n; mc = ring[n&rmsk]; }
                }
              else
                { m = p; mc = mp; }
              last = p;
            }
        }
    }
  free(ring);
}

static void scheme_refine(int lev, int64_t i, int64_t kthresh, int64_t *count, int *states, int *pad)
{ lev += 1;
  if (count[i] >= 0)
    { if (count[i] > kthresh)
        { int a;
          count[i] = -(*states);
          for (a = 0; a < 4; a++)
            count[(*states)++] = 0;
          if (lev > *pad)
            *pad += 2;
        }
      else
        count[i] = 0;
    }
  else
    { int64_t j = -count[i];
      int a;
      for (a = 0; a < 4; a++)
        scheme_refine(lev,j+a,kthresh,count,states,pad);
    }
}

static const int64_t *Sort_Count;
static int by_count_desc(const void *l, const void *r)     /* ties keep their order: a merge sort, like glibc's qsort */
{ int x = *((const int *) l), y = *((const int *) r);
  if (Sort_Count[x] != Sort_Count[y])
    return (Sort_Count[x] > Sort_Count[y] ? -1 : 1);
  return (x < y ? -1 : (x > y));
}

int orc_scheme_train(const orc_params *P, const char *bases, const int64_t *boff, int64_t nreads,
                     int bc_prefix, int nparts, orc_scheme *S)
{ int      states = 1024, pad = 0, npieces = 2*nparts, i, o;
  int64_t *count = calloc(1024,sizeof(int64_t)), *trie, ktot = 0, kthresh, max_count, last_max = 0;
  uint64_t rnd = 0;       /* glibc's unseeded state: X = 0 (its static drand48 data is zeroed), not POSIX's 0x1234ABCD330E */

  for (;;)
    { trie = malloc(sizeof(int64_t)*states);
      for (i = 0; i < states; i++)
        { trie[i] = count[i] < 0 ? count[i] : 0;
          if (count[i] >= 0) count[i] = 0;
        }
      scheme_census(P,pad,bases,boff,nreads,bc_prefix,trie,count);
      free(trie);
      ktot = 0;
      for (i = 0; i < states; i++)
        if (count[i] >= 0)
          ktot += count[i];
      kthresh = ktot/npieces;
      max_count = 0;
      o = states;
      for (i = 0; i < states; i++)
        if (count[i] >= 0)
          { if (count[i] > kthresh) o += 4;
            if (count[i] > max_count) max_count = count[i];
          }
      if (o == states)
        break;
      if (pad > 0 && last_max < 1.02*max_count)
        { npieces = (int) (ktot/max_count+1);
          nparts  = npieces/2;
          break;
        }
      if (5+pad >= P->kmer-1)
        break;
      count = realloc(count,sizeof(int64_t)*o);
      for (i = states; i < o; i++)
        count[i] = 0;
      { int st = states;
        for (i = 0; i < 1024; i++)
          scheme_refine(0,i,kthresh,count,&st,&pad);
      }
      states = o;
      last_max = max_count;
    }
  if (nparts < 1) nparts = 1;
  { int64_t  pmer = ktot/nparts, *buck = calloc(nparts,sizeof(int64_t)), p, v, t;
    int     *perm = malloc(sizeof(int)*states), j, n, x;
    for (i = 0; i < states; i++) perm[i] = i;
    Sort_Count = count;
    qsort(perm,states,sizeof(int),by_count_desc);
    for (i = 0; i < states; i++)
      { x = perm[i];
        p = count[x];
        if (p < 0) continue;
        if (p == 0) { count[x] = nparts-1; continue; }
        v = 0;
        for (j = 0; j < nparts; j++)
          if (buck[j]+p <= pmer)
            v += pmer-buck[j];
        if (v == 0)
          { n = 0;
            for (j = 1; j < nparts; j++)
              if (buck[j] < buck[n]) n = j;
            buck[n] += p;
            count[x] = n;
          }
        else
          { rnd = (rnd*0x5DEECE66Dull + 0xBull) & ((1ull << 48)-1);
            t = (int64_t) ((double) v * ((double) rnd / 281474976710656.0));
            v = 0;
            for (j = 0; j < nparts; j++)
              if (buck[j]+p <= pmer)
                { v += pmer-buck[j];
                  if (v >= t)
                    { buck[j] += p;
                      count[x] = j;
                      break;
                    }
                }
          }
      }
    free(perm);
    free(buck);
  }
  S->pad = pad; S->states = states; S->nparts = nparts;
  S->part = malloc(sizeof(int)*states);
  for (i = 0; i < states; i++)
    S->part[i] = (int) count[i];
  free(count);
  return (0);
}

/* the whole path with the reference's buckets (count.c:1202 bucket loop): every bucket is sorted and counted on
   its own, histograms add, tables are disjoint and merged by key; wfirst is BUCKET 0's census -- Table_Split's
   input (count.c:1560-1565) */
static int Cmp_KB;
static int by_key(const void *l, const void *r) { return memcmp(l,r,Cmp_KB); }

int orc_fastk_parts(const orc_params *P, const orc_scheme *S, const char *bases, const int64_t *boff,
                    int64_t nreads, int bc_prefix, int cutoff, orc_result *R)
{ orc_params PP = *P;
  uint8_t *smers = NULL, *bucket = NULL, *part;
  int64_t  ns = 0, cap = 0, bcap = 0, i;
  int      b;

  orc_params_init(&PP,P->kmer,S->pad);                  /* PAD_LEN-base minimizers; the record widths stay P's */
  memcpy(PP.tran,P->tran,sizeof(PP.tran));
  PP.smer = P->smer; PP.slen_bits = P->slen_bits; PP.slen_bytes = P->slen_bytes;
  PP.smer_bytes = P->smer_bytes; PP.smer_word = P->smer_word;
  memset(R,0,sizeof(*R));
  Dist_Scheme = S; Dist_Bucket = &bucket; Dist_Bcap = &bcap;
  R->ninst = orc_distribute_block(&PP,bases,boff,nreads,bc_prefix,&smers,&ns,&cap);
  Dist_Scheme = NULL;
  R->nsuper = ns;
  part = malloc((size_t) (ns > 0 ? ns : 1)*P->smer_word);
  for (b = 0; b < S->nparts; b++)
    { orc_result Rb;
      uint8_t   *kl = NULL;
      int64_t    nb = 0, nw, ovf, nd;
      for (i = 0; i < ns; i++)
        if (bucket[i] == b)
          memcpy(part+(nb++)*P->smer_word,smers+i*P->smer_word,P->smer_word);
      memset(&Rb,0,sizeof(Rb));
      orc_msd_sort(part,nb,P->smer_word,P->smer_word);
      nw = orc_kmer_list(P,part,nb,&kl,&ovf,&nd);
      if (b == 0)
        for (i = 0; i < nw; i++)
          R->wfirst[kl[i*P->kmer_word]] += 1;
      orc_msd_sort(kl,nw,P->kmer_word,P->kmer_bytes);
      orc_count_sorted(P,kl,nw,cutoff,&Rb);
      R->ndistinct_super += nd;
      R->nweighted += nw;
      R->ndistinct += Rb.ndistinct;
      R->max_inst += Rb.max_inst + ovf;
      for (i = 1; i < 0x8000; i++)
        R->hist[i] += Rb.hist[i];
      if (Rb.ntable > 0)
        { R->table = realloc(R->table,(size_t) (R->ntable+Rb.ntable)*P->kmer_word);
          memcpy(R->table+R->ntable*P->kmer_word,Rb.table,(size_t) Rb.ntable*P->kmer_word);
          R->ntable += Rb.ntable;
        }
      free(Rb.table);
      free(kl);
    }
  Cmp_KB = P->kmer_bytes;
  if (R->ntable > 0)
    qsort(R->table,(size_t) R->ntable,(size_t) P->kmer_word,by_key);
  free(part);
  free(bucket);
  free(smers);
  return (0);
}

/* brute force: every valid window's canonical k-mer, sorted, run-length counted */
int orc_brute(int kmer, const char *bases, const int64_t *boff, int64_t nreads,
              int bc_prefix, int cutoff, orc_result *R)
{ const int8_t *code = code_table();
  orc_params P;
  int64_t    r, n, cap;
  uint8_t   *kl;
  uint8_t    fb[64], rb[64];
  int        KB, KW;

  orc_params_init(&P,kmer,0);
  KB = P.kmer_bytes; KW = P.kmer_word;
  memset(R,0,sizeof(*R));
  cap = 1024; n = 0;
  kl = malloc((size_t) cap*KW);
  for (r = 0; r < nreads; r++)
    { const char *s = bases + boff[r] + bc_prefix;
      int q = (int) (boff[r+1]-boff[r]) - 1 - bc_prefix;
      int i, b, bad;

      for (i = 0; i+kmer <= q; i++)
        { bad = 0;
          for (b = 0; b < kmer; b++)
            if (code[(uint8_t) s[i+b]] >= 4)
              { bad = 1; break; }
          if (bad)
            continue;
          memset(fb,0,KB); memset(rb,0,KB);
          for (b = 0; b < kmer; b++)
            { fb[b>>2] |= (uint8_t) (code[(uint8_t) s[i+b]] << (6-2*(b&3)));
              rb[b>>2] |= (uint8_t) ((3-code[(uint8_t) s[i+kmer-1-b]]) << (6-2*(b&3)));
            }
          if (n >= cap)
            { cap *= 2; kl = realloc(kl,(size_t) cap*KW); }
          memcpy(kl+n*KW,(memcmp(fb,rb,KB) < 0) ? fb : rb,KB);
          kl[n*KW+KB] = 1; kl[n*KW+KB+1] = 0;
          n += 1;
        }
    }
  R->ninst = n;
  R->nweighted = n;
  orc_msd_sort(kl,n,KW,KB);
  orc_count_sorted(&P,kl,n,cutoff,R);
  free(kl);
  return (0);
}

void orc_result_free(orc_result *R)
{ free(R->table);
  R->table = NULL;
}

/* ---------------------------------------------------------------------------------------
 * encodings                                         count.c:1893-1910, table.c:162-342,485-498
 */
int64_t orc_hist_bytes(int kmer, const orc_result *R, uint8_t *buf)
{ int32_t h[3];
  int64_t o = 0;

  h[0] = kmer; h[1] = 1; h[2] = 0x7fff;
  memcpy(buf+o,h,12);                     o += 12;
  memcpy(buf+o,&R->hist[1],8);            o += 8;
  memcpy(buf+o,&R->max_inst,8);           o += 8;
  memcpy(buf+o,&R->hist[1],8*0x7fff);     o += 8*0x7fff;
  return (o);
}

void orc_table_split(const orc_params *P, const orc_result *R, int nthreads, int *split)
{ int64_t asize, sum, thr;
  int     x, n, beg;

  asize = 0;                         /* the array msd_sort partitions is bucket 0's weighted k-mer list (count.c:1560) */
  for (x = 0; x < 256; x++)
    asize += R->wfirst[x] * P->kmer_word;
  thr   = asize / nthreads;
  n = 0; sum = 0; beg = 0;
  for (x = 0; x < 256; x++)
    { sum += R->wfirst[x] * P->kmer_word;
      if (sum >= thr && n < nthreads)
        { split[n++] = beg;
          thr = (asize * (n+1)) / nthreads;
          beg = x+1;
        }
    }
  while (n < nthreads)
    split[n++] = 256;
}

int orc_idx_bytes(int kmer, int64_t ntable)
{ if (ntable > 0x4000000ll && kmer >= 12)
    return (3);
  if (ntable >= 0x40000ll && kmer >= 8)
    return (2);
  return (1);
}

static int write_all(int fd, const void *p, size_t n)
{ const uint8_t *b = p;
  while (n > 0)
    { ssize_t w = write(fd,b,n);
      if (w < 0) return (-1);
      b += w; n -= (size_t) w;
    }
  return (0);
}

int orc_write_outputs(const orc_params *P, const orc_result *R, int cutoff, int nthreads,
                      const int *split, const char *dir, const char *root)
{ char     name[4096];
  uint8_t *hb;
  int      fd, t;

  hb = malloc(262164+16);
  orc_hist_bytes(P->kmer,R,hb);
  snprintf(name,sizeof(name),"%s/%s.hist",dir,root);
  fd = open(name,O_WRONLY|O_CREAT|O_TRUNC,0644);
  if (fd < 0 || write_all(fd,hb,262164) < 0) { free(hb); return (-1); }
  close(fd);
  free(hb);

  if (cutoff <= 0)
    return (0);

  { const int KB = P->kmer_bytes, KW = P->kmer_word;
    int       ib = orc_idx_bytes(P->kmer,R->ntable);
    int64_t   nidx = 1ll << (8*ib);
    int64_t  *idx = calloc((size_t) nidx,sizeof(int64_t));
    int      *sp = malloc(sizeof(int)*(nthreads+1));
    int64_t   i, lo;
    int32_t   h[4];

    if (split == NULL)
      orc_table_split(P,R,nthreads,sp);
    else
      memcpy(sp,split,sizeof(int)*nthreads);
    sp[nthreads] = 256;

    lo = 0;
    for (t = 0; t < nthreads; t++)
      { int64_t hi = lo, n;
        while (hi < R->ntable && R->table[hi*KW] < sp[t+1])
          hi += 1;
        n = hi-lo;
        snprintf(name,sizeof(name),"%s/.%s.ktab.%d",dir,root,t+1);
        fd = open(name,O_WRONLY|O_CREAT|O_TRUNC,0644);
        if (fd < 0) return (-1);
        write_all(fd,&P->kmer,4);
        write_all(fd,&n,8);
        for (i = lo; i < hi; i++)
          { const uint8_t *rec = R->table + i*KW;
            int64_t pre = 0;
            int     b;
            for (b = 0; b < ib; b++)
              pre = (pre << 8) | rec[b];
            idx[pre] += 1;
            write_all(fd,rec+ib,KW-ib);
          }
        close(fd);
        lo = hi;
      }
    for (i = 1; i < nidx; i++)
      idx[i] += idx[i-1];
    snprintf(name,sizeof(name),"%s/%s.ktab",dir,root);
    fd = open(name,O_WRONLY|O_CREAT|O_TRUNC,0644);
    if (fd < 0) return (-1);
    h[0] = P->kmer; h[1] = nthreads; h[2] = cutoff; h[3] = ib;
    write_all(fd,h,16);
    write_all(fd,idx,(size_t) nidx*8);
    close(fd);
    free(sp);
    free(idx);
    (void) KB;
  }
  return (0);
}

/* ---------------------------------------------------------------------------------------
 * inputs
 */
char *orc_synth_block(uint64_t seed, uint64_t genome_len, uint32_t read_len, uint32_t err_ppm,
                      uint64_t first_read, int64_t nreads, int64_t *boff)
{ static const char dna[4] = { 'a', 'c', 'g', 't' };
  fk_synth_spec sp;
  char   *bases;
  int64_t r, o;

  sp.seed = seed; sp.genome_len = genome_len; sp.read_len = read_len; sp.err_ppm = err_ppm;
  bases = malloc((size_t) nreads*(read_len+1)+1);
  o = 0;
  for (r = 0; r < nreads; r++)
    { uint64_t start; uint32_t strand, j;
      fk_synth_place(&sp,first_read+r,&start,&strand);
      boff[r] = o;
      for (j = 0; j < read_len; j++)
        bases[o++] = dna[fk_synth_base(&sp,first_read+r,j,start,strand)];
      bases[o++] = 0;
    }
  boff[nreads] = o;
  return (bases);
}

/* FASTA may be multi-line, FASTQ is strictly 4-line; every non-newline byte of a sequence
   line is a base (io.c:678-734).  File type from the extension (io.c:137-172). */
char *orc_load_fastx(const char *path, int64_t **boffp, int64_t *nreadsp)
{ FILE   *f = fopen(path,"rb");
  int64_t fsize, i, o, nr, rcap;
  char   *buf, *bases;
  int64_t *boff;
  int     fastq, state;
  size_t  pl = strlen(path);

  if (f == NULL)
    return (NULL);
  fastq = (pl >= 3 && strcmp(path+pl-3,".fq") == 0) || (pl >= 6 && strcmp(path+pl-6,".fastq") == 0);
  fseek(f,0,SEEK_END); fsize = ftell(f); fseek(f,0,SEEK_SET);
  buf = malloc((size_t) fsize+1);
  if (fread(buf,1,(size_t) fsize,f) != (size_t) fsize)
    { fclose(f); free(buf); return (NULL); }
  fclose(f);
  bases = malloc((size_t) fsize+2);
  rcap = 1024; nr = 0;
  boff = malloc(sizeof(int64_t)*(rcap+1));
  o = 0; boff[0] = 0;
  state = 0;    /* 0 at record start, 1 header, 2 fastq seq, 3 plus line, 4 qual line, 5 fasta seq, 6 fasta eol */
#define END_READ { bases[o++] = 0; nr += 1; if (nr >= rcap) { rcap *= 2; boff = realloc(boff,sizeof(int64_t)*(rcap+1)); } boff[nr] = o; }
  for (i = 0; i < fsize; i++)
    { int c = buf[i];
      switch (state)
      { case 0: state = 1; break;
        case 1: if (c == '\n') state = fastq ? 2 : 5; break;
        case 2: if (c != '\n') bases[o++] = (char) c; else { END_READ state = 3; } break;
        case 3: if (c == '\n') state = 4; break;
        case 4: if (c == '\n') state = 0; break;
        case 6: if (c == '>') { END_READ state = 1; }
                else if (c != '\n') { bases[o++] = (char) c; state = 5; }
                break;
        case 5: if (c == '\n') state = 6; else bases[o++] = (char) c; break;
      }
    }
  if (state == 6)
    END_READ
#undef END_READ
  free(buf);
  *boffp = boff;
  *nreadsp = nr;
  return (bases);
}


/* ---- profile decoder (libfastk.c:1657-1780, Fetch_Profile) ---------------------------------------
   First count in one byte (< 128) or two (top bit set, 15 bits); then per code byte x:
   00nnnnnn a run of n more copies of the current count; 01dddddd a 6-bit signed difference;
   1ddddddd dddddddd a 15-bit difference added modulo 2^15 (a set 0x40 bit keeps the sign bits,
   :1743-1749). */
int64_t orc_profile_decode_stream(const uint8_t *data, const int64_t *ends, int64_t n, uint8_t *out, int64_t cap)
{ int64_t need = 0, prev = 0, r;
  for (int pass = 0; pass < 2; pass++)
    { uint8_t *o = out;
      prev = 0;
      for (r = 0; r < n; r++)
        { const uint8_t *p = data + prev, *q = data + ends[r];
          int32_t len = 0;
          uint16_t d = 0, x;
          uint8_t *lenp = o;
          prev = ends[r];
          if (pass) o += 4;
          if (p < q)
            { x = *p++;
              if (x & 0x80)
                { if (p >= q) return -1;
                  d = (uint16_t) (((x & 0x7f) << 8) | *p++);
                }
              else
                d = x;
              if (pass) { memcpy(o, &d, 2); o += 2; }
              len = 1;
              while (p < q)
                { x = *p++;
                  if ((x & 0xc0) == 0)
                    { if (pass)
                        for (int i = 0; i < x; i++) { memcpy(o, &d, 2); o += 2; }
                      len += x;
                    }
                  else
                    { if (x & 0x80)
                        { if (p >= q) return -1;
                          if (x & 0x40) x = (uint16_t) (x << 8);
                          else          x = (uint16_t) ((x << 8) & 0x7fff);
                          x |= *p++;
                          d = (uint16_t) ((d + x) & 0x7fff);
                        }
                      else if (x & 0x20)
                        d = (uint16_t) (d + ((x & 0x1fu) | 0xffe0u));
                      else
                        d = (uint16_t) (d + (x & 0x1fu));
                      if (pass) { memcpy(o, &d, 2); o += 2; }
                      len += 1;
                    }
                }
            }
          if (pass) memcpy(lenp, &len, 4);
          else need += 4 + 2 * (int64_t) len;
        }
      if (pass == 0 && (out == NULL || cap < need))
        return need;
    }
  return need;
}
