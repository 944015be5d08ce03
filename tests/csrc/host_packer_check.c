/* Test harness (not shipped): the two-bit packer of FastK_amd's reader threads against a base-by-base restatement,
   on random reads with stretches of non-bases, at every bit offset, with and without AVX2; and the piece parser
   (pk_parse_piece) against the reference's scanner restated one byte at a time, on random odd texts. */
#define main FastK_amd_main
#include "../../fastk_amd/csrc/host/FastK_amd.c"
#undef main

static uint64_t rng = 88172645463325252ull;
static uint32_t rnd(void) { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return ((uint32_t) (rng >> 11)); }

static int run(int avx2)
{ int trial;
  PK_AVX2 = avx2;
  for (trial = 0; trial < 300; trial++)
    { Pk_Buf b;
      int    nreads = 1 + (int) (rnd() % 40), r;
      size_t total = 0, cap = 1 << 20;
      unsigned char *all = malloc(cap);
      int32_t *lens = malloc(sizeof(int32_t)*nreads);
      memset(&b,0,sizeof(b));
      b.codes = calloc(cap/4+64,1);
      for (r = 0; r < nreads; r++)
        { int len = (int) (rnd() % ((trial & 1) ? 300 : 5000)), i;
          unsigned char *s = all+total;
          int64_t start = b.nb;
          for (i = 0; i < len; i++)
            { uint32_t x = rnd();
              s[i] = (unsigned char) "acgtACGT"[x & 7];
              if ((x >> 8) % 400 == 0)
                { int run = 1 + (int) ((x >> 20) % 70), j;
                  for (j = 0; j < run && i < len; j++)
                    s[i++] = (unsigned char) "NnRY\r-"[(x >> 4) % 6];
                  i -= 1;
                }
            }
          pk_bases(&b,s,len);
          pk_end_read(&b,start);
          lens[r] = len;
          total += (size_t) len;
        }
      /* check */
      { size_t i;
        int    k = 0;
        int64_t pos, ninv = 0;
        if (b.nb != (int64_t) total || b.nreads != nreads) { fprintf(stderr,"counts\n"); return (1); }
        for (r = 0; r < nreads; r++)
          if (b.rlen[r] != lens[r]) { fprintf(stderr,"rlen\n"); return (1); }
        for (i = 0; i < total; i++)
          { int v = PK_ONE[all[i]];
            int got = (b.codes[i >> 2] >> (6-2*(i & 3))) & 3;
            int inside = 0;
            while (k < b.ninv && b.inv[2*k]+b.inv[2*k+1] <= (int64_t) i) k += 1;
            if (k < b.ninv && b.inv[2*k] <= (int64_t) i) inside = 1;
            if ((v < 0) != inside) { fprintf(stderr,"invalid stretch at %zu (trial %d, avx2 %d)\n",i,trial,avx2); return (1); }
            if (v >= 0 && got != v) { fprintf(stderr,"code at %zu: %d, not %d (trial %d, avx2 %d)\n",i,got,v,trial,avx2); return (1); }
            if (v < 0 && got != 0) { fprintf(stderr,"non-base not packed as 0 at %zu\n",i); return (1); }
          }
        for (pos = -1, k = 0; k < b.ninv; k++)          /* maximal, ordered, disjoint */
          { if (b.inv[2*k] <= pos) { fprintf(stderr,"stretches touch\n"); return (1); }
            pos = b.inv[2*k]+b.inv[2*k+1];
            ninv += b.inv[2*k+1];
          }
        (void) ninv;
      }
      free(all); free(lens); free(b.codes); free(b.rlen); free(b.inv);
    }
  return (0);
}

/* The reference's scanner (io.c:685-738), one byte at a time: reads as (start, length) pairs into the base string it
   builds.  FASTQ: header line, sequence line (a read ends at its newline), '+' line, quality line.  FASTA: after a
   header's newline every line is sequence until a line that FOLLOWS a sequence line begins with '>'; a read ends at
   that '>' or at the end of the text behind a newline. */
static int ref_scan(const unsigned char *t, size_t n, int fastq, unsigned char *bases, int64_t *rstart, int64_t *rlen)
{ enum { QAT, HSKP, QSEQ, QPLS, QSKP, AEOL, ASEQ } st = QAT;
  int64_t nb = 0, start = 0;
  int     nr = 0;
  size_t  i;
  for (i = 0; i < n; i++)
    { const unsigned char c = t[i];
      switch (st)
      { case QAT:  st = HSKP; break;
        case HSKP: if (c == '\n') { st = fastq ? QSEQ : ASEQ; start = nb; } break;
        case QSEQ: if (c != '\n') bases[nb++] = c; else { rstart[nr] = start; rlen[nr++] = nb-start; st = QPLS; } break;
        case QPLS: if (c == '\n') st = QSKP; break;
        case QSKP: if (c == '\n') st = QAT; break;
        case AEOL: if (c == '>') { rstart[nr] = start; rlen[nr++] = nb-start; st = HSKP; }
                   else if (c != '\n') { bases[nb++] = c; st = ASEQ; }
                   break;
        case ASEQ: if (c == '\n') st = AEOL; else bases[nb++] = c; break;
      }
    }
  if (st == AEOL)
    { rstart[nr] = start; rlen[nr++] = nb-start; }
  return (nr);
}

static int run_parse(int fastq, int avx2)
{ int trial;
  PK_AVX2 = avx2;
  for (trial = 0; trial < 400; trial++)
    { size_t cap = 1 << 18, n = 0;
      unsigned char *t = malloc(cap), *bases = malloc(cap);
      int64_t *rstart = malloc(sizeof(int64_t)*20000), *rlen = malloc(sizeof(int64_t)*20000);
      int nrec = 1 + (int) (rnd() % 30), r, nr, i;
      Pk_Buf b;
      memset(&b,0,sizeof(b));
      for (r = 0; r < nrec && n+6000 < cap; r++)
        { int len = (rnd() % 5 == 0) ? 0 : (int) (rnd() % 400), w = 1 + (int) (rnd() % 90), j;
          n += (size_t) sprintf((char *) t+n,"%cr%d > @ text\n",fastq ? '@' : '>',r);
          if (fastq)
            { for (j = 0; j < len; j++) t[n++] = (unsigned char) "acgtACGTNn"[rnd() % ((rnd() % 9) ? 8 : 10)];
              t[n++] = '\n'; t[n++] = '+'; t[n++] = '\n';
              for (j = 0; j < len; j++) t[n++] = (unsigned char) ((j == 0 && (rnd() & 1)) ? '@' : 'I');
              t[n++] = '\n';
            }
          else
            for (j = 0; j < len; j += w)
              { int k;
                for (k = j; k < j+w && k < len; k++) t[n++] = (unsigned char) "acgtACGTNnRY"[rnd() % ((rnd() % 9) ? 8 : 12)];
                t[n++] = '\n';
                if (rnd() % 40 == 0) t[n++] = '\n';                   /* an empty line inside a record */
              }
        }
      if (rnd() % 3 == 0 && n > 0) n -= 1;                              /* no newline at the very end */
      if (rnd() % 7 == 0 && n > 40) n -= rnd() % 40;                    /* a truncated file */
      nr = ref_scan(t,n,fastq,bases,rstart,rlen);
      b.codes = calloc(cap/4+64,1);
      pk_parse_piece(&b,t,t+n,fastq);
      if (b.nreads != nr)
        { fprintf(stderr,"%s trial %d: %d reads, the reference's scanner finds %d\n",fastq ? "FASTQ" : "FASTA",trial,b.nreads,nr); return (1); }
      { int64_t pos = 0;
        int     k = 0;
        for (r = 0; r < nr; r++)
          { if (b.rlen[r] != rlen[r]) { fprintf(stderr,"trial %d read %d: length %d, not %lld\n",trial,r,b.rlen[r],(long long) rlen[r]); return (1); }
            for (i = 0; i < (int) rlen[r]; i++, pos++)
              { const int v = PK_ONE[bases[rstart[r]+i]];
                const int got = (b.codes[pos >> 2] >> (6-2*(pos & 3))) & 3;
                int inside = 0;
                while (k < b.ninv && b.inv[2*k]+b.inv[2*k+1] <= pos) k += 1;
                if (k < b.ninv && b.inv[2*k] <= pos) inside = 1;
                if ((v < 0) != inside || (v >= 0 && got != v))
                  { fprintf(stderr,"trial %d read %d base %d differs\n",trial,r,i); return (1); }
              }
          }
        if (pos != b.nb) { fprintf(stderr,"trial %d: %lld bases, not %lld\n",trial,(long long) b.nb,(long long) pos); return (1); }
        while (k < b.ninv && b.inv[2*k]+b.inv[2*k+1] <= pos) k += 1;
        if (k < b.ninv) { fprintf(stderr,"trial %d: a stretch past the last base\n",trial); return (1); }
      }
      free(t); free(bases); free(rstart); free(rlen); free(b.codes); free(b.rlen); free(b.inv);
    }
  return (0);
}

int main(void)
{ pk_tables();
  if (run_parse(0,0) || run_parse(1,0)) return (1);
#if defined(__x86_64__)
  if (__builtin_cpu_supports("avx2") && (run_parse(0,1) || run_parse(1,1))) return (1);
#endif
  if (run(0)) return (1);
#if defined(__x86_64__)
  if (__builtin_cpu_supports("avx2") && run(1)) return (1);
#endif
  printf("packer OK\n");
  return (0);
}
