// fk_count.hip -- sorted weighted k-mers -> count histogram + (k-mer,count) table entries.
//
// Replaces the leaf hist_kmers (MSDsort.c:491-509): per distinct k-mer cnt = sum of the uint16
// weights of its run; cnt >= 0x7fff -> hist[0x7fff]++, max_inst += cnt, cnt = 0x7fff, else
// hist[cnt]++; the head of the run receives cnt.  And table_write_thread (count.c:564-616):
// heads with cnt >= cutoff are compacted, in order, into the table.
//
// Pass A stages a tile (+ look-ahead) in LDS, walks each run from its head there (LDS-privatised
// low histogram bins, 64-bit global atomics for the tail) and counts table entries per tile; a
// single-workgroup scan turns tile counts into offsets; pass B repeats the walk and compacts.
#include "fk_common.h"

#define CT_THREADS 256
#define CT_ITEMS   8
#define CT_TILE    (CT_THREADS * CT_ITEMS)
#define CT_LOWBINS 4096
#define CT_AHEAD   255
#define CT_TPB     8       // tiles per workgroup in k_ct_fast (one histogram flush per workgroup)
#define CT_FIXCAP  1024    // heterogeneous prefix runs repaired per tile
#define CT_MAXFIX  192     // longest heterogeneous prefix run one thread insertion-sorts

// key comparison through per-word masks (all ones / partial last word / zero): branch-free
template <int KW> struct CtMask { u32 m[KW]; };

template <int KW>
__device__ __forceinline__ CtMask<KW> ct_make_mask(int kmer_bytes)
{ CtMask<KW> k;
  const int full = kmer_bytes >> 2;
  const u32 last = (kmer_bytes & 3) ? ((1u << (8 * (kmer_bytes & 3))) - 1u) : 0u;
#pragma unroll
  for (int w = 0; w < KW; w++)
    k.m[w] = (w < full) ? 0xffffffffu : (w == full) ? last : 0u;
  return k;
}

template <int KW>
__device__ __forceinline__ bool ct_same_key(const u32 *a, const u32 *b, const CtMask<KW> &k)
{ u32 diff = 0;
#pragma unroll
  for (int w = 0; w < KW; w++)
    diff |= (a[w] ^ b[w]) & k.m[w];
  return (diff == 0);
}

// lexicographic order of two keys (memory-order bytes): big-endian word compare under the masks
template <int KW>
__device__ __forceinline__ bool ct_key_less(const u32 *a, const u32 *b, const CtMask<KW> &k)
{ bool less = false;                                   // mask arithmetic, no short-circuit: no branches
#pragma unroll
  for (int w = KW - 1; w >= 0; w--)
    { const u32 x = __builtin_bswap32(a[w] & k.m[w]);
      const u32 y = __builtin_bswap32(b[w] & k.m[w]);
      less = (x < y) | ((x == y) & less);
    }
  return less;
}

// TABLE = false: histogram, max_inst, distinct count and the number of table entries per tile.
// TABLE = true : the same walk again (cheap: it runs in LDS), this time compacting the entries with
//                count >= cutoff into the table at the offsets the tile scan produced.
// PREFIX = true: the input is sorted on its first `prefix_bytes` key bytes only (four LSD passes
//                instead of KMER_BYTES).  A run of equal prefixes almost always holds ONE k-mer (its
//                duplicates); the few runs that hold several are insertion-sorted inside LDS by the
//                thread that owns their first record, so the walk below still sees sorted keys.  A
//                prefix run belongs to the tile that contains its first record, tail included.
//                Runs that do not fit the LDS window and are not homogeneous raise *unresolved and
//                the host falls back to the remaining digit passes.
// The input is never modified (the reference stores the total into the run head, MSDsort.c:507;
// on the GPU that would be 244 M scattered 4-byte writes into 13 GB).
template <int KW, bool TABLE, bool PREFIX>
__global__ __launch_bounds__(CT_THREADS) void k_ct_count(const u32 *__restrict__ km, int64_t n,
                                                         int kmer_bytes, int prefix_bytes, int cutoff,
                                                         u64 *__restrict__ hist,
                                                         u64 *__restrict__ scal,   // [0] max_inst [1] distinct [3] unresolved
                                                         u32 *__restrict__ tile_entries,
                                                         const u64 *__restrict__ tile_off,
                                                         u32 *__restrict__ table)
{ constexpr int WIN = CT_TILE + CT_AHEAD;                // records visible to a tile
  constexpr int NIT = PREFIX ? (WIN + CT_THREADS - 1) / CT_THREADS : CT_ITEMS;
  __shared__ u32 low[TABLE ? 1 : CT_LOWBINS];
  __shared__ u32 tmp[8];
  __shared__ u32 s_run;
  __shared__ int s_first, s_end;                         // PREFIX: owned range [s_first, s_end)
  __shared__ u32 s_nfix;
  __shared__ u32 fixq[PREFIX ? CT_FIXCAP : 1];           // heterogeneous runs: start << 12 | length
  __shared__ __attribute__((aligned(16))) u32 recs[(WIN + 1) * KW];  // tile + look-ahead
  __shared__ u32 prev[KW];                               // the record before the tile
  if (!TABLE)
    for (int i = threadIdx.x; i < CT_LOWBINS; i += CT_THREADS)
      low[i] = 0;
  if (threadIdx.x == 0)
    { s_run = 0;
      s_first = WIN + 1;
      s_end = WIN + 1;
      s_nfix = 0;
    }

  const CtMask<KW> kmask = ct_make_mask<KW>(kmer_bytes);
  const CtMask<KW> pmask = ct_make_mask<KW>(PREFIX ? prefix_bytes : kmer_bytes);
  const int cw  = (KW * 4 - 2) >> 2;
  const int csh = 8 * ((KW * 4 - 2) & 3);

  const int64_t t0 = (int64_t) blockIdx.x * CT_TILE;
  // stage records [t0, t0+CT_TILE+CT_AHEAD) so that run heads walk forward in LDS
  int64_t gend = t0 + WIN;
  if (gend > n) gend = n;
  const int nl = (int) (gend - t0);                      // records staged
  fk_stage16<(WIN * KW + 1023) / 1024, false>(recs, km + t0 * KW, nl * KW);
  if (t0 > 0 && threadIdx.x < KW)
    prev[threadIdx.x] = km[(t0 - 1) * KW + threadIdx.x];
  __syncthreads();

  int own_lo = 0, own_hi = (nl < CT_TILE) ? nl : CT_TILE;
  if (PREFIX)
    { // ---- fix-up --------------------------------------------------------------------------
      // per record, in parallel: does it open a prefix run (phead), or does it continue a prefix
      // run with a DIFFERENT k-mer than its predecessor (evidence that the run is heterogeneous)?
#pragma unroll 1
      for (int it = 0; it < NIT; it++)
        { const int l = it * CT_THREADS + threadIdx.x;
          bool phead = false, evid = false;
          if (l < nl)
            { const u32 *r  = recs + l * KW;
              const u32 *rp = (l > 0) ? r - KW : prev;
              if (t0 + l == 0)
                phead = true;
              else
                { phead = !ct_same_key<KW>(r, rp, pmask);
                  evid  = !phead && !ct_same_key<KW>(r, rp, kmask);
                }
            }
          // first prefix head inside the tile / first one in the look-ahead: one atomic per wave
          const u64 hm = __ballot(phead);
          if (hm != 0 && fk_lane() == 0)
            { const int first = (it * CT_THREADS + (int) (threadIdx.x & ~63u)) + (__ffsll((unsigned long long) hm) - 1);
              if (first < CT_TILE)
                atomicMin(&s_first, first);
              // the wave's lanes are consecutive records: find its first head at or past CT_TILE
              const int wbeg = it * CT_THREADS + (int) (threadIdx.x & ~63u);
              u64 hm2 = hm;
              if (wbeg < CT_TILE)
                hm2 = (CT_TILE - wbeg >= 64) ? 0ull : (hm >> (CT_TILE - wbeg)) << (CT_TILE - wbeg);
              if (hm2 != 0)
                atomicMin(&s_end, wbeg + (__ffsll((unsigned long long) hm2) - 1));
            }
          if (evid)
            { // rare: locate the run; only its FIRST evidence record queues the repair
              int  sidx = l - 1;
              bool first = true, found = false;
              while (sidx >= 0)
                { const u32 *q  = recs + sidx * KW;
                  const u32 *qp = (sidx > 0) ? q - KW : prev;
                  const bool ph = (t0 + sidx == 0) || !ct_same_key<KW>(q, qp, pmask);
                  if (ph)
                    { found = true;
                      break;
                    }
                  if (!ct_same_key<KW>(q, qp, kmask))
                    { first = false;
                      break;
                    }
                  sidx -= 1;
                }
              if (first && found && sidx < CT_TILE)     // run owned by this tile
                { int e = l + 1;
                  while (e < nl && ct_same_key<KW>(recs + e * KW, recs + l * KW, pmask))
                    e += 1;
                  bool bad = (e - sidx > CT_MAXFIX);
                  if (e == nl && t0 + nl < n
                      && ct_same_key<KW>(km + (t0 + nl) * KW, recs + l * KW, pmask))
                    bad = true;                           // run leaves the LDS window
                  if (!bad)
                    { const u32 slot = atomicAdd(&s_nfix, 1u);
                      if (slot < CT_FIXCAP)
                        fixq[slot] = ((u32) sidx << 12) | (u32) (e - sidx);
                      else
                        bad = true;
                    }
                  if (bad)
                    atomicAdd(&scal[3], 1ull);
                }
            }
        }
      __syncthreads();
      // insertion sort of the queued runs, one thread per run (runs are disjoint)
      const u32 nfix = (s_nfix < CT_FIXCAP) ? s_nfix : CT_FIXCAP;
      for (u32 f = threadIdx.x; f < nfix; f += CT_THREADS)
        { const int l = (int) (fixq[f] >> 12);
          const int m = (int) (fixq[f] & 0xfffu);
          for (int a = 1; a < m; a++)
            { u32 hold[KW];
#pragma unroll
              for (int w = 0; w < KW; w++)
                hold[w] = recs[(l + a) * KW + w];
              int b = a;
              while (b > 0 && ct_key_less<KW>(hold, recs + (l + b - 1) * KW, kmask))
                {
#pragma unroll
                  for (int w = 0; w < KW; w++)
                    recs[(l + b) * KW + w] = recs[(l + b - 1) * KW + w];
                  b -= 1;
                }
#pragma unroll
              for (int w = 0; w < KW; w++)
                recs[(l + b) * KW + w] = hold[w];
            }
        }
      __syncthreads();
      own_lo = s_first;
      own_hi = (s_end < nl) ? s_end : nl;
      if (own_lo > CT_TILE)
        own_lo = own_hi = 0;                   // no run starts in this tile: nothing is owned
    }

  const u64 base = TABLE ? tile_off[blockIdx.x] : 0ull;
  u32 entries = 0, distinct = 0;
  u64 maxi = 0;
#pragma unroll 1
  for (int it = 0; it < NIT; it++)
    { const int     l = it * CT_THREADS + threadIdx.x;
      const int64_t i = t0 + l;
      u32  mycnt = 0;                     // 0 = this lane holds no run head
      const u32 *r = recs + l * KW;
      if (l >= own_lo && l < own_hi)
        { const bool head = (i == 0) || !ct_same_key<KW>(r, (l > 0) ? r - KW : prev, kmask);
          if (head)
            { u64 cnt = (r[cw] >> csh) & 0xffffu;
              int lj = l + 1;
              int64_t j = i + 1;
              // two explicit loops: a pointer that may be LDS or global would compile to FLAT loads
              bool open = true;
              while (open && j < n && lj < nl)
                { const u32 *q = recs + lj * KW;
                  if (ct_same_key<KW>(r, q, kmask))
                    { cnt += (q[cw] >> csh) & 0xffffu;
                      j += 1; lj += 1;
                    }
                  else
                    open = false;
                }
              while (open && j < n)                    // run longer than the look-ahead: rare
                { const u32 *q = km + j * KW;
                  if (ct_same_key<KW>(r, q, kmask))
                    { cnt += (q[cw] >> csh) & 0xffffu;
                      j += 1;
                    }
                  else
                    open = false;
                }
              distinct += 1;
              if (cnt >= 0x7fff)                                     // MSDsort.c:498-506
                { maxi += cnt;
                  cnt = 0x7fff;
                }
              mycnt = (u32) cnt;
            }
        }
      const bool take = (cutoff > 0 && mycnt >= (u32) cutoff);
      if (!TABLE)
        { if (take)
            entries += 1;
          if (mycnt != 0)
            { if (mycnt < CT_LOWBINS)
                atomicAdd(&low[mycnt], 1u);
              else
                atomicAdd(&hist[mycnt], 1ull);
            }
        }
      else
        { u32 tot;
          const u32 ex  = fk_block_exscan_256<u32>(take ? 1u : 0u, tmp, &tot);
          const u32 run = s_run;
          __syncthreads();
          if (threadIdx.x == 0)
            s_run = run + tot;
          if (take)
            { u32 *dst = table + (base + run + ex) * KW;
#pragma unroll
              for (int w = 0; w < KW; w++)
                dst[w] = (w == cw) ? ((r[w] & ~(0xffffu << csh)) | (mycnt << csh)) : r[w];
            }
        }
    }
  if (TABLE)
    return;
  __syncthreads();
  for (int i = threadIdx.x; i < CT_LOWBINS; i += CT_THREADS)
    if (low[i] != 0)
      atomicAdd(&hist[i], (u64) low[i]);
  u32 te, td;
  (void) fk_block_exscan_256<u32>(entries, tmp, &te);
  (void) fk_block_exscan_256<u32>(distinct, tmp, &td);
  if (threadIdx.x == 0)
    { tile_entries[blockIdx.x] = te;
      if (td != 0)
        atomicAdd(&scal[1], (u64) td);
    }
  if (maxi != 0)
    atomicAdd(&scal[0], maxi);
}

// Fully sorted input, every lane busy: a wave looks at 64 consecutive records at a time; run heads
// come from one ballot, the weights are prefix-summed across the wave, and a head's total inside
// its 64-record segment is a difference of two prefix sums.  A run that reaches the end of its
// segment picks up the leading non-head weights of the following segments (kept in LDS), and only
// a run that leaves the tile's look-ahead window walks on in global memory.
template <int KW, bool TABLE>
__global__ __launch_bounds__(CT_THREADS) void k_ct_fast(const u32 *__restrict__ km, int64_t n,
                                                        int kmer_bytes, int cutoff,
                                                        u64 *__restrict__ hist,
                                                        u64 *__restrict__ scal,
                                                        u32 *__restrict__ tile_entries,
                                                        const u64 *__restrict__ tile_off,
                                                        u32 *__restrict__ table, int collapse)
{ // collapse != 0: the input is only GROUPED (equal k-mers adjacent); every run becomes one record
  // carrying its weight sum clipped to 0x7fff, the clipped remainder goes to scal[4] (the same
  // accounting as count.c:455-458); no histogram is taken -- that happens after the real sort.
  constexpr int WIN  = CT_TILE + CT_AHEAD + 1;           // 2304 records = 36 segments of 64
  constexpr int NSEG = WIN / 64;
  constexpr int NIT  = WIN / CT_THREADS;                 // 9
  __shared__ u32 low[TABLE ? 1 : CT_LOWBINS];
  __shared__ u32 tmp[8];
  __shared__ u32 s_run;
  __shared__ __attribute__((aligned(16))) u32 recs[(WIN + 1) * KW];
  __shared__ u32 prev[KW];
  __shared__ u32 part[WIN];                              // head: run sum inside its segment | open << 31
  __shared__ u32 lead[NSEG + 1];                         // weights before the first head of a segment
  __shared__ u32 nohead[NSEG + 1];
  if (!TABLE)
    for (int i = threadIdx.x; i < CT_LOWBINS; i += CT_THREADS)
      low[i] = 0;
  if (threadIdx.x == 0)
    s_run = 0;

  const CtMask<KW> kmask = ct_make_mask<KW>(kmer_bytes);
  const int cw  = (KW * 4 - 2) >> 2;
  const int csh = 8 * ((KW * 4 - 2) & 3);
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;

  u32 entries = 0, distinct = 0;
  u64 maxi = 0;
#pragma unroll 1
  for (int tb = 0; tb < CT_TPB; tb++)
  {
  const int64_t tileno = (int64_t) blockIdx.x * CT_TPB + tb;
  const int64_t t0 = tileno * CT_TILE;
  if (t0 >= n)
    break;
  __syncthreads();                                       // LDS of the previous tile is free
  if (threadIdx.x == 0)
    s_run = 0;
  int64_t gend = t0 + WIN;
  if (gend > n) gend = n;
  const int nl = (int) (gend - t0);
  fk_stage16<(WIN * KW + 1023) / 1024, false>(recs, km + t0 * KW, nl * KW);
  if (t0 > 0 && threadIdx.x < KW)
    prev[threadIdx.x] = km[(t0 - 1) * KW + threadIdx.x];
  __syncthreads();

  // ---- phase 1: per 64-record segment -----------------------------------------------------
#pragma unroll 1
  for (int it = 0; it < NIT; it++)
    { const int l = it * CT_THREADS + threadIdx.x;
      bool head = false;
      u32  wt = 0;
      if (l < nl)
        { const u32 *r = recs + l * KW;
          head = (t0 + l == 0) || !ct_same_key<KW>(r, (l > 0) ? r - KW : prev, kmask);
          wt = (r[cw] >> csh) & 0xffffu;
        }
      const u64 hm = __ballot(head);
      u32 S = wt;                                   // inclusive prefix sum over the wave
#pragma unroll
      for (int o = 1; o < 64; o <<= 1)
        { const u32 y = (u32) __shfl_up((int) S, o, 64);
          if (lane >= o) S += y;
        }
      const u64 above = (lane == 63) ? 0ull : (hm >> (lane + 1));
      const int e     = (above != 0) ? lane + 1 + (__ffsll((unsigned long long) above) - 1) : 64;
      const u32 Se    = (u32) __shfl((int) S, e - 1, 64);
      const int first = (hm != 0) ? (__ffsll((unsigned long long) hm) - 1) : 64;
      const u32 Sl    = (u32) __shfl((int) S, (first > 0 ? first : 1) - 1, 64);
      if (l < WIN)
        part[l] = head ? ((Se - S + wt) | ((e == 64) ? 0x80000000u : 0u)) : 0u;
      if (lane == 0)
        { const int sg = it * (CT_THREADS / 64) + wave;
          lead[sg]   = (first == 0) ? 0u : Sl;
          nohead[sg] = (hm == 0) ? 1u : 0u;
        }
    }
  __syncthreads();

  // ---- phase 2: totals, histogram, table -------------------------------------------------------
  const int nseg = (nl + 63) >> 6;
  const u64 base = TABLE ? tile_off[tileno] : 0ull;
  u32 tentries = 0;
#pragma unroll 1
  for (int it = 0; it < CT_ITEMS; it++)
    { const int l = it * CT_THREADS + threadIdx.x;
      const u32 *r = recs + l * KW;
      u32 mycnt = 0;
      const u32 v = (l < nl) ? part[l] : 0u;
      if (v != 0)
        { u64 cnt = v & 0x7fffffffu;
          if (v >> 31)
            { int sg = (l >> 6) + 1;
              bool open = true;
              while (open && sg < nseg)
                { cnt += lead[sg];
                  open = (nohead[sg] != 0);
                  sg += 1;
                }
              if (open)                               // run leaves the window: rare
                for (int64_t j = t0 + nl; j < n; j++)
                  { const u32 *q = km + j * KW;
                    if (!ct_same_key<KW>(r, q, kmask))
                      break;
                    cnt += (q[cw] >> csh) & 0xffffu;
                  }
            }
          distinct += 1;
          if (collapse)
            { if (cnt > 0x7fff)
                { maxi += cnt - 0x7fff;
                  cnt = 0x7fff;
                }
            }
          else if (cnt >= 0x7fff)                                // MSDsort.c:498-506
            { maxi += cnt;
              cnt = 0x7fff;
            }
          mycnt = (u32) cnt;
        }
      const bool take = collapse ? (mycnt != 0) : (cutoff > 0 && mycnt >= (u32) cutoff);
      if (!TABLE)
        { if (take)
            tentries += 1;
          if (mycnt != 0 && !collapse)
            { if (mycnt < CT_LOWBINS)
                atomicAdd(&low[mycnt], 1u);
              else
                atomicAdd(&hist[mycnt], 1ull);
            }
        }
      else
        { u32 tot;
          const u32 ex  = fk_block_exscan_256<u32>(take ? 1u : 0u, tmp, &tot);
          const u32 run = s_run;
          __syncthreads();
          if (threadIdx.x == 0)
            s_run = run + tot;
          if (take)
            { u32 *dst = table + (base + run + ex) * KW;
#pragma unroll
              for (int w = 0; w < KW; w++)
                dst[w] = (w == cw) ? ((r[w] & ~(0xffffu << csh)) | (mycnt << csh)) : r[w];
            }
        }
    }
  if (!TABLE)
    { u32 te;
      (void) fk_block_exscan_256<u32>(tentries, tmp, &te);
      if (threadIdx.x == 0)
        tile_entries[tileno] = te;
    }
  (void) entries;
  }   // tiles of this workgroup
  if (TABLE)
    { if (collapse && maxi != 0)
        atomicAdd(&scal[4], maxi);
      return;
    }
  if (collapse)
    return;
  __syncthreads();
  // one histogram flush per CT_TPB tiles (global 64-bit atomics are the expensive part)
  for (int i = threadIdx.x; i < CT_LOWBINS; i += CT_THREADS)
    if (low[i] != 0)
      atomicAdd(&hist[i], (u64) low[i]);
  u32 td;
  (void) fk_block_exscan_256<u32>(distinct, tmp, &td);
  if (threadIdx.x == 0 && td != 0)
    atomicAdd(&scal[1], (u64) td);
  if (maxi != 0)
    atomicAdd(&scal[0], maxi);
}

// sorted_bytes < KMER_BYTES: the records are ordered on their first sorted_bytes key bytes only;
// returns FK_ESTATE (nothing accumulated) when some prefix run could not be resolved inside LDS.
template <int KW>
static int count_t(fk_ctx *ctx, void *d_kmers, int64_t n, int cutoff, int sorted_bytes, int64_t *hist,
                   int64_t *max_inst, int64_t *ndistinct, void *d_table, int64_t cap,
                   int64_t *ntable)
{ hipStream_t s = ctx->stream;
  const int64_t ntiles = (n + CT_TILE - 1) / CT_TILE;

  if (ntable) *ntable = 0;
  if (ndistinct) *ndistinct = 0;
  if (n == 0)
    return (FK_OK);

  u32 *d_ent  = (u32 *) fk_slot(ctx, FK_SLOT_CT_ENT, ntiles * 4);
  u64 *d_off  = (u64 *) fk_slot(ctx, FK_SLOT_CT_OFF, ntiles * 8);
  u64 *d_hist = (u64 *) fk_slot(ctx, FK_SLOT_CT_HIST, (FK_HIST_BINS + 8) * 8);
  if (d_ent == NULL || d_off == NULL || d_hist == NULL)
    return (FK_ENOMEM);
  u64 *d_scal = d_hist + FK_HIST_BINS;     // [0] max_inst [1] distinct [2] total entries
  u64 *h = (u64 *) malloc((FK_HIST_BINS + 8) * 8);

  int rc = FK_OK;
  do
    { if (h == NULL) { rc = FK_ENOMEM; break; }
      if (hipMemsetAsync(d_hist, 0, (FK_HIST_BINS + 8) * 8, s) != hipSuccess) { rc = FK_EHIP; break; }
      const bool prefix = (sorted_bytes < ctx->wid.kmer_bytes);
      if (prefix)
        hipLaunchKernelGGL((k_ct_count<KW, false, true>), dim3((unsigned) ntiles), dim3(CT_THREADS), 0, s,
                           (const u32 *) d_kmers, n, ctx->wid.kmer_bytes, sorted_bytes, cutoff, d_hist,
                           d_scal, d_ent, (const u64 *) NULL, (u32 *) NULL);
      else
        hipLaunchKernelGGL((k_ct_fast<KW, false>), dim3((unsigned) ((ntiles + CT_TPB - 1) / CT_TPB)), dim3(CT_THREADS), 0, s,
                           (const u32 *) d_kmers, n, ctx->wid.kmer_bytes, cutoff, d_hist,
                           d_scal, d_ent, (const u64 *) NULL, (u32 *) NULL, 0);
      hipLaunchKernelGGL(k_exscan_tiles, dim3(1), dim3(256), 0, s, (const u32 *) d_ent, ntiles,
                         d_off, d_scal + 2);
      if (hipGetLastError() != hipSuccess) { rc = FK_EHIP; break; }
      if (fkx_d2h_pageable(ctx, s, h, d_hist, (FK_HIST_BINS + 8) * 8) != FK_OK)                // (h is malloc'ed)
        { rc = FK_EHIP; break; }
      if (h[FK_HIST_BINS + 3] != 0)
        { rc = FK_ESTATE;          // heterogeneous prefix run outside the LDS window: caller sorts on
          break;
        }
      for (int i = 1; i < FK_HIST_BINS; i++)
        hist[i] += (int64_t) h[i];
      *max_inst += (int64_t) h[FK_HIST_BINS + 0];
      if (ndistinct) *ndistinct = (int64_t) h[FK_HIST_BINS + 1];
      const int64_t nt = (int64_t) h[FK_HIST_BINS + 2];
      if (ntable) *ntable = nt;
      if (cutoff <= 0 || d_table == NULL)
        break;
      if (cap < nt)
        { fk_set_error(ctx, "table buffer too small: %lld entries needed, %lld given",
                       (long long) nt, (long long) cap);
          rc = FK_EINVAL;
          break;
        }
      if (prefix)
        hipLaunchKernelGGL((k_ct_count<KW, true, true>), dim3((unsigned) ntiles), dim3(CT_THREADS), 0, s,
                           (const u32 *) d_kmers, n, ctx->wid.kmer_bytes, sorted_bytes, cutoff, d_hist,
                           d_scal, d_ent, (const u64 *) d_off, (u32 *) d_table);
      else
        hipLaunchKernelGGL((k_ct_fast<KW, true>), dim3((unsigned) ((ntiles + CT_TPB - 1) / CT_TPB)), dim3(CT_THREADS), 0, s,
                           (const u32 *) d_kmers, n, ctx->wid.kmer_bytes, cutoff, d_hist,
                           d_scal, d_ent, (const u64 *) d_off, (u32 *) d_table, 0);
      if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
        { rc = FK_EHIP; break; }
    }
  while (0);
  if (rc == FK_EHIP)
    fk_set_error(ctx, "count: HIP failure: %s", hipGetErrorString(hipGetLastError()));
  free(h);
  return (rc);
}

int fkx_count(fk_ctx *ctx, const void *d_kmers, int64_t nweighted, int cutoff, int sorted_bytes,
              int64_t *hist, int64_t *max_inst, int64_t *ndistinct,
              void *d_table, int64_t cap, int64_t *ntable)
{ switch (ctx->wid.kmer_stride >> 2)
  { case 1: return count_t<1>(ctx, (void *) d_kmers, nweighted, cutoff, sorted_bytes, hist, max_inst, ndistinct, d_table, cap, ntable);
    case 2: return count_t<2>(ctx, (void *) d_kmers, nweighted, cutoff, sorted_bytes, hist, max_inst, ndistinct, d_table, cap, ntable);
    case 3: return count_t<3>(ctx, (void *) d_kmers, nweighted, cutoff, sorted_bytes, hist, max_inst, ndistinct, d_table, cap, ntable);
    case 4: return count_t<4>(ctx, (void *) d_kmers, nweighted, cutoff, sorted_bytes, hist, max_inst, ndistinct, d_table, cap, ntable);
    case 5: return count_t<5>(ctx, (void *) d_kmers, nweighted, cutoff, sorted_bytes, hist, max_inst, ndistinct, d_table, cap, ntable);
    default:
      fk_set_error(ctx, "k-mer stride %d not built", ctx->wid.kmer_stride);
      return (FK_EUNSUPPORTED);
  }
}

// Grouped weighted k-mers -> one record per run with the weight sum (clipped, remainder in *overflow).
template <int KW>
static int collapse_t(fk_ctx *ctx, const void *d_kmers, int64_t n, void *d_out, int64_t cap,
                      int64_t *nout, int64_t *overflow)
{ hipStream_t s = ctx->stream;
  const int64_t ntiles = (n + CT_TILE - 1) / CT_TILE;
  *nout = 0; *overflow = 0;
  if (n == 0)
    return (FK_OK);
  u32 *d_ent  = (u32 *) fk_slot(ctx, FK_SLOT_CT_ENT, ntiles * 4);
  u64 *d_off  = (u64 *) fk_slot(ctx, FK_SLOT_CT_OFF, ntiles * 8);
  u64 *d_hist = (u64 *) fk_slot(ctx, FK_SLOT_CT_HIST, (FK_HIST_BINS + 8) * 8);
  if (d_ent == NULL || d_off == NULL || d_hist == NULL)
    return (FK_ENOMEM);
  u64 *d_scal = d_hist + FK_HIST_BINS;
  const unsigned grid = (unsigned) ((ntiles + CT_TPB - 1) / CT_TPB);
  FK_HIP(ctx, hipMemsetAsync(d_scal, 0, 8 * 8, s));
  hipLaunchKernelGGL((k_ct_fast<KW, false>), dim3(grid), dim3(CT_THREADS), 0, s, (const u32 *) d_kmers, n,
                     ctx->wid.kmer_bytes, 1, d_hist, d_scal, d_ent, (const u64 *) NULL, (u32 *) NULL, 1);
  hipLaunchKernelGGL(k_exscan_tiles, dim3(1), dim3(256), 0, s, (const u32 *) d_ent, ntiles, d_off,
                     d_scal + 2);
  FK_LAUNCH_CHECK(ctx);
  FK_HIP(ctx, hipMemcpyAsync(ctx->h_scratch, d_scal, 8 * 8, hipMemcpyDeviceToHost, s));
  FK_HIP(ctx, hipStreamSynchronize(s));
  *nout = (int64_t) ctx->h_scratch[2];
  if (cap < *nout)
    { fk_set_error(ctx, "collapse buffer too small: %lld records needed, %lld given",
                   (long long) *nout, (long long) cap);
      return (FK_EINVAL);
    }
  hipLaunchKernelGGL((k_ct_fast<KW, true>), dim3(grid), dim3(CT_THREADS), 0, s, (const u32 *) d_kmers, n,
                     ctx->wid.kmer_bytes, 1, d_hist, d_scal, d_ent, (const u64 *) d_off, (u32 *) d_out, 1);
  FK_LAUNCH_CHECK(ctx);
  FK_HIP(ctx, hipMemcpyAsync(ctx->h_scratch, d_scal, 8 * 8, hipMemcpyDeviceToHost, s));
  FK_HIP(ctx, hipStreamSynchronize(s));
  *overflow = (int64_t) ctx->h_scratch[4];
  return (FK_OK);
}

int fkx_collapse(fk_ctx *ctx, const void *d_kmers, int64_t n, void *d_out, int64_t cap,
                 int64_t *nout, int64_t *overflow)
{ switch (ctx->wid.kmer_stride >> 2)
  { case 1: return collapse_t<1>(ctx, d_kmers, n, d_out, cap, nout, overflow);
    case 2: return collapse_t<2>(ctx, d_kmers, n, d_out, cap, nout, overflow);
    case 3: return collapse_t<3>(ctx, d_kmers, n, d_out, cap, nout, overflow);
    case 4: return collapse_t<4>(ctx, d_kmers, n, d_out, cap, nout, overflow);
    case 5: return collapse_t<5>(ctx, d_kmers, n, d_out, cap, nout, overflow);
    default:
      fk_set_error(ctx, "k-mer stride %d not built", ctx->wid.kmer_stride);
      return (FK_EUNSUPPORTED);
  }
}
