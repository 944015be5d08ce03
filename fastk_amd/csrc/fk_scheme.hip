// fk_scheme.hip -- the reference's bucket scheme, for exact_parts runs that the reference would cut into NPARTS > 1
// buckets: the padded-minimizer prefix trie (Determine_Scheme, split.c:617-766: leaf census by
// padded_minimizer_thread split.c:116-270, refine_tree split.c:437-472) and the deal of its leaves to the buckets
// (assign_pieces, split.c:289-381, with drand48 from its default seed).
//
// Nothing here changes a count: .hist and the .ktab stream are the same under any scheme.  What depends on it is where
// the hidden .ktab part files are cut -- Table_Split takes the first-byte ranges of the threads that sort BUCKET 0's
// weighted k-mers (count.c:1560-1565) -- so a run that shall reproduce every file of the reference at -M<small> has to
// put exactly the reference's super-mers into bucket 0.
//
// The census runs on the device, one thread per read of the training block (the reads Get_First_Block would hand
// over, io.c:2606-2630); the trie and the deal are a few thousand integers on the host.
#include "fk_common.h"
#include <algorithm>
#include <vector>

#define SC_THREADS 128
#define SC_RING    256          // >= 2 * nextpow2(K), like the exact splitter's

struct SchemeArgs
{ const unsigned char *bases;
  const int64_t *roff;          // [nreads+1]
  int64_t    nreads;
  int        kmer, bc_prefix;
  int        tran[4];
  int        pad_len;           // MIN_LEN + PAD
  int        pad2;              // 2 * PAD
  const int *trie;              // [states] < 0: children at -trie[x] .. -trie[x]+3, else leaf
  u64       *cnt;               // [states] k-mers per leaf
  u64       *nmin;              // super-mers closed
};

// padded_minimizer_thread (split.c:116-270): the super-mer rule of the trainer -- every byte that is not acgt counts
// as 'a' (Tran[] defaults to assn[0], split.c:563-568), strict < both on arrival and in the rescan after a forced cut,
// no flush at the end of a read -- and per closed super-mer its k-mers go to the trie leaf of its minimizer.
__global__ __launch_bounds__(SC_THREADS) void k_scheme_census(SchemeArgs a)
{ const int64_t r = (int64_t) blockIdx.x * SC_THREADS + threadIdx.x;
  if (r >= a.nreads)
    return;
  const int K = a.kmer;
  const unsigned char *s = a.bases + a.roff[r] + a.bc_prefix;
  const int q = (int) (a.roff[r + 1] - a.roff[r]) - 1 - a.bc_prefix;
  if (q < K)
    return;
  int rmsk = 1;
  while (rmsk < K) rmsk <<= 1;
  rmsk = 2 * rmsk - 1;
  const int PL1 = a.pad_len - 1;
  const int MS  = K - PL1;                                   // MAX_SUPER, split.c:628
  const u32 msk = (a.pad_len >= 16) ? 0xffffffffu : ((1u << (2 * a.pad_len)) - 1u);
  const u32 tot = msk;                                       // (any value above every minimizer)
  const int t0 = a.tran[0], t1 = a.tran[1], t2 = a.tran[2], t3 = a.tran[3];
  auto code_of = [&](unsigned ch) -> int
    { const unsigned u = ch & 0xDFu;
      return (u == 0x43u) ? 1 : (u == 0x47u) ? 2 : (u == 0x54u) ? 3 : 0;       // everything else is 'a'
    };
  auto fwv = [&](int code) -> u32 { return (u32) (code == 1 ? t1 : code == 2 ? t2 : code == 3 ? t3 : t0); };
  auto rcv = [&](int code) -> u32 { return (u32) (code == 1 ? t2 : code == 2 ? t1 : code == 3 ? t0 : t3) << (2 * PL1); };

  u32 ring[SC_RING];
  u32 c = 0, u = 0, mp = 0, mc = tot;
  bool mc_set = false;
  int m = 0, p;
  for (p = 0; p < K; p++)
    { const int code = code_of(s[p]);
      c = ((c << 2) | fwv(code)) & msk;
      u = (u >> 2) | rcv(code);
      if (p >= PL1)
        { mp = (u < c) ? u : c;
          ring[p & rmsk] = mp;
          if (!mc_set || mp < mc)                            // (mc starts at PAD_TOT, above every value)
            { m = p; mc = mp; mc_set = true; }
        }
    }
  int last = K - 1;
  u64 nm = 0;
  for (p = K; p < q; p++)
    { const int code = code_of(s[p]);
      c = ((c << 2) | fwv(code)) & msk;
      u = (u >> 2) | rcv(code);
      mp = (u < c) ? u : c;
      ring[p & rmsk] = mp;
      const bool force = (p - m >= MS);
      if (force || mp < mc)
        { int o = (int) (mc >> a.pad2);
          int v = a.trie[o];
          int b = a.pad2 - 2;
          while (v < 0)
            { o = (int) ((mc >> b) & 3u) - v;
              v = a.trie[o];
              b -= 2;
            }
          atomicAdd(&a.cnt[o], (u64) (p - last));
          nm += 1;
          if (force)
            { m += 1;
              mc = ring[m & rmsk];
              for (int n = m + 1; n <= p; n++)
                { const u32 x = ring[n & rmsk];
                  if (x < mc)
                    { m = n; mc = x; }
                }
            }
          else
            { m = p; mc = mp; }
          last = p;
        }
    }
  if (nm != 0)
    atomicAdd(a.nmin, nm);
}

// drand48 as the reference meets it: never seeded, and glibc's state starts zeroed -- X0 = 0, not POSIX's
// 0x1234ABCD330E --, X' = 0x5DEECE66D X + 0xB mod 2^48, value X' / 2^48: the same draws in every run
struct Drand48
{ uint64_t x = 0;
  double next()
  { x = (x * 0x5DEECE66Dull + 0xBull) & ((1ull << 48) - 1);
    return ((double) x / 281474976710656.0);
  }
};

/* Trains the scheme on reads [0, train) of d_bases (roff on both sides): on return ctx->min_part (host) and
   ctx->d_min_part hold the trie with bucket numbers at its leaves, ctx->scheme_pad the padding, ctx->scheme_states the
   trie's size and ctx->scheme_nparts the number of buckets (the reference lowers the request when an even split is
   not possible, split.c:740-746). */
int fkx_train_scheme(fk_ctx *ctx, const void *d_bases, const int64_t *d_roff, int64_t train, const int *tran,
                     int nparts_req)
{ hipStream_t s = ctx->stream;
  const int K = ctx->prm.kmer;
  const int MIN_LEN = 5, MIN_TOT = 1024;
  int NPARTS = nparts_req, npieces = 2 * nparts_req;
  int PAD = 0, states = MIN_TOT;
  std::vector<int64_t> count((size_t) MIN_TOT, 0);
  int64_t last_max = 0, ktot = 0;

  u64 *d_cnt = NULL;
  int *d_trie = NULL;
  int64_t cap = 0;
  int rc = FK_OK;
  for (;;)
    { const int PAD_LEN = MIN_LEN + PAD;
      if (PAD_LEN > 15)
        { fk_set_error(ctx, "exact_parts: the reference's scheme needs %d-base minimizers here; this engine follows it up to 15",
                       PAD_LEN);
          rc = FK_EUNSUPPORTED;
          break;
        }
      if (states > cap)
        { if (d_cnt) hipFree(d_cnt);
          if (d_trie) hipFree(d_trie);
          d_cnt = NULL; d_trie = NULL;
          cap = states * 2 + 4096;
          if (hipMalloc((void **) &d_cnt, (size_t) cap * 8 + 8) != hipSuccess || hipMalloc((void **) &d_trie, (size_t) cap * 4) != hipSuccess)
            { rc = FK_ENOMEM; break; }
        }
      std::vector<int> trie((size_t) states);
      for (int i = 0; i < states; i++)
        trie[i] = (count[i] < 0) ? (int) count[i] : 0;
      if (fkx_h2d_pageable(ctx, s, d_trie, trie.data(), (size_t) states * 4) != FK_OK
          || hipMemsetAsync(d_cnt, 0, (size_t) states * 8 + 8, s) != hipSuccess)
        { rc = FK_EHIP; break; }
      SchemeArgs a;
      a.bases = (const unsigned char *) d_bases;
      a.roff = d_roff;
      a.nreads = train;
      a.kmer = K;
      a.bc_prefix = ctx->prm.bc_prefix;
      for (int i = 0; i < 4; i++) a.tran[i] = tran[i];
      a.pad_len = PAD_LEN;
      a.pad2 = 2 * PAD;
      a.trie = d_trie;
      a.cnt = d_cnt;
      a.nmin = d_cnt + states;
      if (train > 0)
        hipLaunchKernelGGL(k_scheme_census, dim3((unsigned) ((train + SC_THREADS - 1) / SC_THREADS)), dim3(SC_THREADS), 0, s, a);
      std::vector<u64> hc((size_t) states + 1);
      if (hipGetLastError() != hipSuccess
          || fkx_d2h_pageable(ctx, s, hc.data(), d_cnt, (size_t) states * 8 + 8) != FK_OK)
        { rc = FK_EHIP; break; }
      ktot = 0;
      for (int i = 0; i < states; i++)
        if (count[i] >= 0)
          { count[i] = (int64_t) hc[(size_t) i];
            ktot += count[i];
          }
      const int64_t kthresh = ktot / npieces;                 // split.c:677
      int64_t max_count = 0;
      int o = states;
      for (int i = 0; i < states; i++)
        if (count[i] >= 0)
          { if (count[i] > kthresh) o += 4;
            if (count[i] > max_count) max_count = count[i];
          }
      if (o == states)                                        // split.c:737-750
        break;
      if (PAD > 0 && (double) last_max < 1.02 * (double) max_count)
        { npieces = (int) (ktot / max_count + 1);
          NPARTS = npieces / 2;
          break;
        }
      if (PAD_LEN >= K - 1)
        break;
      // refine_tree (split.c:437-472): a leaf above the threshold gets four children; the others start from zero
      count.resize((size_t) o, 0);
      int next = states;
      struct R
      { static void go(int lev, int i, int64_t kthresh, std::vector<int64_t> &count, int &next, int &PAD)
        { lev += 1;
          if (count[(size_t) i] >= 0)
            { if (count[(size_t) i] > kthresh)
                { count[(size_t) i] = -next;
                  for (int a = 0; a < 4; a++)
                    count[(size_t) next++] = 0;
                  if (lev > PAD)
                    PAD += 2;
                }
              else
                count[(size_t) i] = 0;
            }
          else
            { const int j = (int) -count[(size_t) i];
              for (int a = 0; a < 4; a++)
                go(lev, j + a, kthresh, count, next, PAD);
            }
        }
      };
      for (int i = 0; i < MIN_TOT; i++)
        R::go(0, i, kthresh, count, next, PAD);
      states = o;
      last_max = max_count;
    }
  if (d_cnt) hipFree(d_cnt);
  if (d_trie) hipFree(d_trie);
  if (rc != FK_OK)
    return (rc);
  if (NPARTS < 1) NPARTS = 1;
  if (NPARTS > FK_EXACT_MAXPARTS)
    { fk_set_error(ctx, "exact_parts: the reference would cut this input into %d buckets; this engine follows it up to %d",
                   NPARTS, FK_EXACT_MAXPARTS);
      return (FK_EUNSUPPORTED);
    }

  // assign_pieces (split.c:289-381): leaves by falling count (qsort is a stable merge sort in glibc 2.35, the
  // reference build's: equal counts keep their index order); a leaf goes to one of the buckets it still fits
  // in, drawn with weights "room left"; when it fits nowhere, to the emptiest bucket
  { const int64_t pmer = ktot / NPARTS;
    std::vector<int> perm((size_t) states);
    for (int i = 0; i < states; i++) perm[(size_t) i] = i;
    std::stable_sort(perm.begin(), perm.end(), [&](int x, int y) { return count[(size_t) x] > count[(size_t) y]; });
    std::vector<int64_t> buck((size_t) NPARTS, 0);
    Drand48 rnd;
    for (int i = 0; i < states; i++)
      { const int x = perm[(size_t) i];
        const int64_t p = count[(size_t) x];
        if (p < 0)
          continue;
        if (p == 0)
          { count[(size_t) x] = NPARTS - 1;
            continue;
          }
        int64_t v = 0;
        for (int j = 0; j < NPARTS; j++)
          if (buck[(size_t) j] + p <= pmer)
            v += pmer - buck[(size_t) j];
        if (v == 0)
          { int n = 0;
            for (int j = 1; j < NPARTS; j++)
              if (buck[(size_t) j] < buck[(size_t) n])
                n = j;
            buck[(size_t) n] += p;
            count[(size_t) x] = n;
          }
        else
          { const int64_t t = (int64_t) ((double) v * rnd.next());
            v = 0;
            for (int j = 0; j < NPARTS; j++)
              if (buck[(size_t) j] + p <= pmer)
                { v += pmer - buck[(size_t) j];
                  if (v >= t)
                    { buck[(size_t) j] += p;
                      count[(size_t) x] = j;
                      break;
                    }
                }
          }
      }
  }
  free(ctx->min_part);
  ctx->min_part = (int *) malloc(sizeof(int) * (size_t) states);
  if (ctx->min_part == NULL)
    return (FK_ENOMEM);
  for (int i = 0; i < states; i++)
    ctx->min_part[i] = (int) count[(size_t) i];
  if (ctx->d_min_part) hipFree(ctx->d_min_part);
  ctx->d_min_part = NULL;
  if (hipMalloc((void **) &ctx->d_min_part, (size_t) states * 4) != hipSuccess)
    return (FK_ENOMEM);
  if (fkx_h2d_pageable(ctx, s, ctx->d_min_part, ctx->min_part, (size_t) states * 4) != FK_OK)    // (min_part is malloc'ed)
    return (FK_EHIP);
  ctx->scheme_pad = PAD;
  ctx->scheme_states = states;
  ctx->scheme_nparts = NPARTS;
  return (FK_OK);
}
