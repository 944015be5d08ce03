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

// TABLE = false: histogram, max_inst, distinct count and the number of table entries per tile.
// TABLE = true : the same walk again (cheap: it runs in LDS), this time compacting the entries with
//                count >= cutoff into the table at the offsets the tile scan produced.
// The input is never modified (the reference stores the total into the run head, MSDsort.c:507;
// on the GPU that would be 244 M scattered 4-byte writes into 13 GB).
template <int KW, bool TABLE>
__global__ __launch_bounds__(CT_THREADS) void k_ct_count(const u32 *__restrict__ km, int64_t n,
                                                         int kmer_bytes, int cutoff,
                                                         u64 *__restrict__ hist,
                                                         u64 *__restrict__ scal,   // [0] max_inst [1] distinct
                                                         u32 *__restrict__ tile_entries,
                                                         const u64 *__restrict__ tile_off,
                                                         u32 *__restrict__ table)
{ __shared__ u32 low[TABLE ? 1 : CT_LOWBINS];
  __shared__ u32 tmp[8];
  __shared__ u32 s_run;
  __shared__ __attribute__((aligned(16))) u32 recs[(CT_TILE + CT_AHEAD + 1) * KW];  // tile + look-ahead
  __shared__ u32 prev[KW];                               // the record before the tile
  if (!TABLE)
    for (int i = threadIdx.x; i < CT_LOWBINS; i += CT_THREADS)
      low[i] = 0;
  if (threadIdx.x == 0)
    s_run = 0;

  const CtMask<KW> kmask = ct_make_mask<KW>(kmer_bytes);
  const int cw  = (KW * 4 - 2) >> 2;
  const int csh = 8 * ((KW * 4 - 2) & 3);

  const int64_t t0 = (int64_t) blockIdx.x * CT_TILE;
  // stage records [t0, t0+CT_TILE+CT_AHEAD) so that run heads walk forward in LDS
  int64_t gend = t0 + CT_TILE + CT_AHEAD;
  if (gend > n) gend = n;
  const int nl = (int) (gend - t0);                      // records staged
  fk_stage16<((CT_TILE + CT_AHEAD) * KW + 1023) / 1024, false>(recs, km + t0 * KW, nl * KW);
  if (t0 > 0 && threadIdx.x < KW)
    prev[threadIdx.x] = km[(t0 - 1) * KW + threadIdx.x];
  __syncthreads();

  const u64 base = TABLE ? tile_off[blockIdx.x] : 0ull;
  u32 entries = 0, distinct = 0;
  u64 maxi = 0;
#pragma unroll 1
  for (int it = 0; it < CT_ITEMS; it++)
    { const int     l = it * CT_THREADS + threadIdx.x;
      const int64_t i = t0 + l;
      u32  mycnt = 0;                     // 0 = this lane holds no run head
      const u32 *r = recs + l * KW;
      if (i < n)
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

template <int KW>
static int count_t(fk_ctx *ctx, void *d_kmers, int64_t n, int cutoff, int64_t *hist,
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
      hipLaunchKernelGGL((k_ct_count<KW, false>), dim3((unsigned) ntiles), dim3(CT_THREADS), 0, s,
                         (const u32 *) d_kmers, n, ctx->wid.kmer_bytes, cutoff, d_hist, d_scal, d_ent,
                         (const u64 *) NULL, (u32 *) NULL);
      hipLaunchKernelGGL(k_exscan_tiles, dim3(1), dim3(256), 0, s, (const u32 *) d_ent, ntiles,
                         d_off, d_scal + 2);
      if (hipGetLastError() != hipSuccess) { rc = FK_EHIP; break; }
      if (hipMemcpyAsync(h, d_hist, (FK_HIST_BINS + 8) * 8, hipMemcpyDeviceToHost, s) != hipSuccess
          || hipStreamSynchronize(s) != hipSuccess)
        { rc = FK_EHIP; break; }
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
      hipLaunchKernelGGL((k_ct_count<KW, true>), dim3((unsigned) ntiles), dim3(CT_THREADS), 0, s,
                         (const u32 *) d_kmers, n, ctx->wid.kmer_bytes, cutoff, d_hist, d_scal, d_ent,
                         (const u64 *) d_off, (u32 *) d_table);
      if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
        { rc = FK_EHIP; break; }
    }
  while (0);
  if (rc == FK_EHIP)
    fk_set_error(ctx, "count: HIP failure: %s", hipGetErrorString(hipGetLastError()));
  free(h);
  return (rc);
}

int fkx_count(fk_ctx *ctx, const void *d_kmers, int64_t nweighted, int cutoff,
              int64_t *hist, int64_t *max_inst, int64_t *ndistinct,
              void *d_table, int64_t cap, int64_t *ntable)
{ switch (ctx->wid.kmer_stride >> 2)
  { case 1: return count_t<1>(ctx, (void *) d_kmers, nweighted, cutoff, hist, max_inst, ndistinct, d_table, cap, ntable);
    case 2: return count_t<2>(ctx, (void *) d_kmers, nweighted, cutoff, hist, max_inst, ndistinct, d_table, cap, ntable);
    case 3: return count_t<3>(ctx, (void *) d_kmers, nweighted, cutoff, hist, max_inst, ndistinct, d_table, cap, ntable);
    case 4: return count_t<4>(ctx, (void *) d_kmers, nweighted, cutoff, hist, max_inst, ndistinct, d_table, cap, ntable);
    case 5: return count_t<5>(ctx, (void *) d_kmers, nweighted, cutoff, hist, max_inst, ndistinct, d_table, cap, ntable);
    default:
      fk_set_error(ctx, "k-mer stride %d not built", ctx->wid.kmer_stride);
      return (FK_EUNSUPPORTED);
  }
}
