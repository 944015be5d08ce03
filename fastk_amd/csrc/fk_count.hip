// fk_count.hip -- sorted weighted k-mers -> count histogram + (k-mer,count) table entries.
//
// Replaces the leaf hist_kmers (MSDsort.c:491-509): per distinct k-mer cnt = sum of the uint16
// weights of its run; cnt >= 0x7fff -> hist[0x7fff]++, max_inst += cnt, cnt = 0x7fff, else
// hist[cnt]++; the head of the run receives cnt.  And table_write_thread (count.c:564-616):
// heads with cnt >= cutoff are compacted, in order, into the table.
//
// Pass A walks each run from its head (LDS-privatised low histogram bins, 64-bit global atomics
// for the tail), stores cnt into the head record like the reference does and counts table
// entries per tile; a single-workgroup scan turns tile counts into offsets; pass B compacts.
#include "fk_common.h"

#define CT_THREADS 256
#define CT_ITEMS   8
#define CT_TILE    (CT_THREADS * CT_ITEMS)
#define CT_LOWBINS 4096

template <int KW>
__device__ __forceinline__ bool ct_same_key(const u32 *a, const u32 *b, int full, u32 lastmask)
{ bool same = true;
#pragma unroll
  for (int w = 0; w < KW; w++)
    if (w < full)
      same &= (a[w] == b[w]);
    else if (w == full)
      same &= (((a[w] ^ b[w]) & lastmask) == 0);
  return same;
}

template <int KW>
__global__ __launch_bounds__(CT_THREADS) void k_ct_count(u32 *__restrict__ km, int64_t n,
                                                         int kmer_bytes, int cutoff,
                                                         u64 *__restrict__ hist,
                                                         u64 *__restrict__ scal,   // [0] max_inst [1] distinct
                                                         u32 *__restrict__ tile_entries)
{ __shared__ u32 low[CT_LOWBINS];
  __shared__ u32 tmp[8];
  for (int i = threadIdx.x; i < CT_LOWBINS; i += CT_THREADS)
    low[i] = 0;
  __syncthreads();

  const int full = kmer_bytes >> 2;                                   // complete key words
  const u32 lastmask = (kmer_bytes & 3) ? ((1u << (8 * (kmer_bytes & 3))) - 1u) : 0u;
  const int cw  = (KW * 4 - 2) >> 2;
  const int csh = 8 * ((KW * 4 - 2) & 3);

  const int64_t t0 = (int64_t) blockIdx.x * CT_TILE;
  u32 entries = 0, distinct = 0;
  u64 maxi = 0;
#pragma unroll 1
  for (int it = 0; it < CT_ITEMS; it++)
    { const int64_t i = t0 + it * CT_THREADS + threadIdx.x;
      if (i >= n)
        continue;
      u32 *r = km + i * KW;
      const bool head = (i == 0) || !ct_same_key<KW>(r, r - KW, full, lastmask);
      if (!head)
        continue;
      u64 cnt = (r[cw] >> csh) & 0xffffu;
      for (int64_t j = i + 1; j < n; j++)
        { const u32 *q = km + j * KW;
          if (!ct_same_key<KW>(r, q, full, lastmask))
            break;
          cnt += (q[cw] >> csh) & 0xffffu;
        }
      distinct += 1;
      if (cnt >= 0x7fff)
        { maxi += cnt;
          cnt = 0x7fff;
        }
      if (cnt < CT_LOWBINS)
        atomicAdd(&low[cnt], 1u);
      else
        atomicAdd(&hist[cnt], 1ull);
      r[cw] = (r[cw] & ~(0xffffu << csh)) | (((u32) cnt) << csh);
      if (cutoff > 0 && cnt >= (u64) cutoff)
        entries += 1;
    }
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
__global__ __launch_bounds__(CT_THREADS) void k_ct_table(const u32 *__restrict__ km, int64_t n,
                                                         int kmer_bytes, int cutoff,
                                                         const u64 *__restrict__ tile_off,
                                                         u32 *__restrict__ table)
{ __shared__ u32 tmp[8];
  __shared__ u32 s_run;
  const int full = kmer_bytes >> 2;
  const u32 lastmask = (kmer_bytes & 3) ? ((1u << (8 * (kmer_bytes & 3))) - 1u) : 0u;
  const int cw  = (KW * 4 - 2) >> 2;
  const int csh = 8 * ((KW * 4 - 2) & 3);
  const int64_t t0 = (int64_t) blockIdx.x * CT_TILE;
  const u64 base = tile_off[blockIdx.x];
  if (threadIdx.x == 0)
    s_run = 0;
  __syncthreads();
#pragma unroll 1
  for (int it = 0; it < CT_ITEMS; it++)
    { const int64_t i = t0 + it * CT_THREADS + threadIdx.x;
      bool take = false;
      const u32 *r = km + i * KW;
      if (i < n)
        { const bool head = (i == 0) || !ct_same_key<KW>(r, r - KW, full, lastmask);
          take = head && (((r[cw] >> csh) & 0xffffu) >= (u32) cutoff);
        }
      u32 tot;
      const u32 ex = fk_block_exscan_256<u32>(take ? 1u : 0u, tmp, &tot);
      const u32 run = s_run;
      __syncthreads();
      if (threadIdx.x == 0)
        s_run = run + tot;
      if (take)
        { u32 *dst = table + (base + run + ex) * KW;
#pragma unroll
          for (int w = 0; w < KW; w++)
            dst[w] = r[w];
        }
    }
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
      hipLaunchKernelGGL(k_ct_count<KW>, dim3((unsigned) ntiles), dim3(CT_THREADS), 0, s,
                         (u32 *) d_kmers, n, ctx->wid.kmer_bytes, cutoff, d_hist, d_scal, d_ent);
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
      hipLaunchKernelGGL(k_ct_table<KW>, dim3((unsigned) ntiles), dim3(CT_THREADS), 0, s,
                         (const u32 *) d_kmers, n, ctx->wid.kmer_bytes, cutoff, (const u64 *) d_off,
                         (u32 *) d_table);
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
