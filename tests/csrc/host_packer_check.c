/* Test harness (not shipped): the two-bit packer of FastK_amd's reader threads against a base-by-base restatement,
   on random reads with stretches of non-bases, at every bit offset, with and without AVX2. */
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

int main(void)
{ pk_tables();
  if (run(0)) return (1);
#if defined(__x86_64__)
  if (__builtin_cpu_supports("avx2") && run(1)) return (1);
#endif
  printf("packer OK\n");
  return (0);
}
