// fk_tsort.hip -- the MSD engine: records ordered on their first `ksize` bytes, most significant byte first.
//
// What the reference's radix_sort does (MSDsort.c:129-261): partition on the leading byte, recurse into every
// part on the next byte, stop recursing where a part holds <= 15 records and finish those with shell_sort on the
// remaining bytes (:105-127), pass over a digit that is the same for all records of a part (:144-158).  The same
// three ideas here, shaped for a device whose passes must be single coalesced launches over the whole array:
//   * DEPTH.  n records seldom agree in their leading P = ceil(log256 n) non-constant key bytes, so only those P
//     digits are radix-partitioned -- P array passes whatever ksize is (a 19-byte super-mer key: 4 passes, not 19).
//     The P levels are executed as stable 8-bit scatter passes, last level first: every level is then ONE launch
//     of the stream engine (fk_radix.hip) over all records, and the array that leaves the last pass is exactly the
//     array an MSD recursion of depth P leaves -- parts of equal P-byte prefix, in prefix order.
//   * SMALL PARTS IN LDS.  The parts are short (under one record on average); a workgroup loads a tile plus a halo,
//     finds the parts that begin in its tile and insertion-sorts each on the remaining key bytes in LDS
//     (k_ts_fix_tile: the role of shell_sort), and writes back the stretch it owns.  One more array pass.
//   * CONSTANT DIGITS are passed over: the level histograms are made first (one read of the array), a digit whose
//     histogram has one bin is not a level (fk_radix.hip, lsd_sort_stream_t).
// A part too long for a tile's halo (repetitive data) sends all tied records through the general route: they are
// pulled out (stream compaction, in position order), sorted on the full key among themselves, and written back
// into the slots they came from -- runs of ties are contiguous and in run order both before and after, so the i-th
// sorted tie belongs in the i-th tie slot.  The result is the same total order whatever the data looks like.
// Used for the table (distinct k-mers, KMER_BYTES: the reference's table comes out of Weighted_Kmer_Sort sorted,
// MSDsort.c:536-544; here the table records leave the aggregation in no order) and exported as
// fk_msd_sort_records for any record width up to 32 bytes.
#include "fk_common.h"

#define TS_THREADS 256
#define TS_ITEMS   8
#define TS_TILE    (TS_THREADS * TS_ITEMS)
#define TS_FIX_ITEMS(KW) ((KW) <= 5 ? 8 : 4)

template <int KW>
__device__ __forceinline__ bool ts_same(const u32 *a, const u32 *b, int full, u32 lastm)
{ u32 diff = 0;                                         // no short-circuit: straight-line code
#pragma unroll
  for (int w = 0; w < KW; w++)
    { const u32 m = (w < full) ? 0xffffffffu : (w == full) ? lastm : 0u;
      diff |= (a[w] ^ b[w]) & m;
    }
  return (diff == 0u);
}

// EMIT = false: tile_count[t] = records of tile t that agree with a neighbour on the first pbytes bytes
// EMIT = true : those records go to sub[] and their positions to pos[], from offset tile_off[t]
template <int KW, bool EMIT>
__global__ __launch_bounds__(TS_THREADS) void k_ts_ties(const u32 *__restrict__ recs, int64_t n, int pbytes,
                                                        u32 *__restrict__ tile_count,
                                                        const u64 *__restrict__ tile_off,
                                                        u32 *__restrict__ sub, u64 *__restrict__ pos)
{ __shared__ u32 tmp[8];
  const int  full  = pbytes >> 2;
  const u32  lastm = (pbytes & 3) ? ((1u << (8 * (pbytes & 3))) - 1u) : 0u;
  const int64_t t0 = (int64_t) blockIdx.x * TS_TILE + (int64_t) threadIdx.x * TS_ITEMS;
  u32 r[TS_ITEMS + 2][KW];
#pragma unroll
  for (int j = 0; j < TS_ITEMS + 2; j++)
    { const int64_t i = t0 + j - 1;
#pragma unroll
      for (int w = 0; w < KW; w++)
        r[j][w] = (i >= 0 && i < n) ? recs[i * KW + w] : 0u;
    }
  u32 flags = 0, cnt = 0;
#pragma unroll
  for (int j = 1; j <= TS_ITEMS; j++)
    { const int64_t i = t0 + j - 1;
      if (i < n)
        { const bool tie = (i > 0 && ts_same<KW>(r[j], r[j - 1], full, lastm))
                        || (i + 1 < n && ts_same<KW>(r[j], r[j + 1], full, lastm));
          if (tie)
            { flags |= 1u << j;
              cnt += 1;
            }
        }
    }
  u32 tot;
  const u32 ex = fk_block_exscan_256<u32>(cnt, tmp, &tot);
  if (!EMIT)
    { if (threadIdx.x == 0)
        tile_count[blockIdx.x] = tot;
      return;
    }
  u64 o = tile_off[blockIdx.x] + ex;
#pragma unroll
  for (int j = 1; j <= TS_ITEMS; j++)
    if ((flags >> j) & 1u)
      {
#pragma unroll
        for (int w = 0; w < KW; w++)
          sub[o * KW + w] = r[j][w];
        pos[o] = (u64) (t0 + j - 1);
        o += 1;
      }
}

// records as byte strings: a < b on the first kbytes bytes
template <int KW>
__device__ __forceinline__ bool ts_less(const u32 *a, const u32 *b, int kfull, u32 klast)
{ bool less = false;                                    // from the last word up, mask arithmetic only
#pragma unroll
  for (int w = KW - 1; w >= 0; w--)
    { const u32 m = (w < kfull) ? 0xffffffffu : (w == kfull) ? klast : 0u;
      const u32 x = __builtin_bswap32(a[w] & m), y = __builtin_bswap32(b[w] & m);
      less = (x < y) | ((x == y) & less);
    }
  return (less);
}

// In-place repair of short tie runs: the thread of a run's first record insertion-sorts the run on
// the full key (runs are disjoint, so no other thread touches these records).  Read data makes such
// runs common and short: a k-mer and its single-substitution error variants agree on every prefix
// that does not contain the substituted base.  A run longer than `limit` raises *long_runs and is
// left to the compaction path.
template <int KW>
__global__ __launch_bounds__(TS_THREADS) void k_ts_fix(u32 *__restrict__ recs, int64_t n, int pbytes,
                                                       int kbytes, int limit, u32 *__restrict__ long_runs)
{ const int  full  = pbytes >> 2;
  const u32  lastm = (pbytes & 3) ? ((1u << (8 * (pbytes & 3))) - 1u) : 0u;
  const int  kfull = kbytes >> 2;
  const u32  klast = (kbytes & 3) ? ((1u << (8 * (kbytes & 3))) - 1u) : 0u;
  const int64_t i = (int64_t) blockIdx.x * TS_THREADS + threadIdx.x;
  if (i >= n)
    return;
  u32 me[KW], nb[KW];
#pragma unroll
  for (int w = 0; w < KW; w++)
    me[w] = recs[i * KW + w];
  if (i > 0)
    {
#pragma unroll
      for (int w = 0; w < KW; w++)
        nb[w] = recs[(i - 1) * KW + w];
      if (ts_same<KW>(me, nb, full, lastm))
        return;                                    // not the first record of its run
    }
  for (int64_t j = i + 1; j < n; j++)
    { u32 x[KW];
#pragma unroll
      for (int w = 0; w < KW; w++)
        x[w] = recs[j * KW + w];
      if (!ts_same<KW>(me, x, full, lastm))
        break;
      if (j - i >= limit)
        { *long_runs = 1;
          return;
        }
      int64_t k = j - 1;                            // insert x into the sorted recs[i..j-1]
      while (k >= i)
        { u32 y[KW];
#pragma unroll
          for (int w = 0; w < KW; w++)
            y[w] = recs[k * KW + w];
          if (!ts_less<KW>(x, y, kfull, klast))
            break;
#pragma unroll
          for (int w = 0; w < KW; w++)
            recs[(k + 1) * KW + w] = y[w];
          k -= 1;
        }
      if (k + 1 != j)
        {
#pragma unroll
          for (int w = 0; w < KW; w++)
            recs[(k + 1) * KW + w] = x[w];
        }
    }
}

// The same repair through LDS: a workgroup loads a tile of TS_TILE records plus TS_HALO more, sorts
// the runs that START in its tile (they may reach into the halo) and writes back the stretch from its
// first run start to the end of its last run -- the stretch before belongs to a run of the previous
// tile, whose workgroup writes it.  Records read while a neighbour rewrites them may be torn, but
// only their first pbytes bytes are looked at, and those are the same for every record of a run.
#define TS_HALO 64

// (ITEMS records per thread: 8 up to 20-byte records, 4 above -- the tile and its halo must fit 64 KB of LDS)
template <int KW, int ITEMS>
__global__ __launch_bounds__(TS_THREADS) void k_ts_fix_tile(u32 *__restrict__ recs, int64_t n, int pbytes,
                                                            int kbytes, u32 *__restrict__ long_runs)
{ constexpr int FT_ITEMS = ITEMS, FT_TILE = TS_THREADS * ITEMS;
  __shared__ __attribute__((aligned(16))) u32 t[(FT_TILE + TS_HALO) * KW];
  __shared__ u32 prev[KW];                                // the record in front of the tile
  __shared__ int s_lo, s_hi;
  const int  full  = pbytes >> 2;
  const u32  lastm = (pbytes & 3) ? ((1u << (8 * (pbytes & 3))) - 1u) : 0u;
  const int  kfull = kbytes >> 2;
  const u32  klast = (kbytes & 3) ? ((1u << (8 * (kbytes & 3))) - 1u) : 0u;
  const int64_t t0 = (int64_t) blockIdx.x * FT_TILE;      // global index of LDS slot 0
  const int  tn = (int) ((n - t0 < FT_TILE + TS_HALO) ? (n - t0) : (FT_TILE + TS_HALO));   // slots 0..tn-1
  const int  own = (tn < FT_TILE) ? tn : FT_TILE;         // run starts this workgroup owns: slots 0..own-1
  fk_stage16<((FT_TILE + TS_HALO) * KW + 1023) / 1024, false>(t, recs + t0 * KW, tn * KW);
  if (threadIdx.x < KW)
    prev[threadIdx.x] = (t0 > 0) ? recs[(t0 - 1) * KW + threadIdx.x] : 0u;
  if (threadIdx.x == 0) { s_lo = FT_TILE + TS_HALO + 1; s_hi = 0; }
  __syncthreads();
  int lo = FT_TILE + TS_HALO + 1, hi = 0;
  for (int q = 0; q < FT_ITEMS; q++)
    { const int i = q * TS_THREADS + threadIdx.x;          // LDS slot
      if (i >= own)
        continue;
      const bool head = (t0 + i == 0) || !ts_same<KW>(t + i * KW, (i > 0) ? t + (i - 1) * KW : prev, full, lastm);
      if (!head)
        continue;
      lo = min(lo, i);
      int j = i + 1;
      for (; j < tn; j++)
        { u32 x[KW];
#pragma unroll
          for (int w = 0; w < KW; w++)
            x[w] = t[j * KW + w];
          if (!ts_same<KW>(t + i * KW, x, full, lastm))
            break;
          int k = j - 1;
          while (k >= i && ts_less<KW>(x, t + k * KW, kfull, klast))
            {
#pragma unroll
              for (int w = 0; w < KW; w++)
                t[(k + 1) * KW + w] = t[k * KW + w];
              k -= 1;
            }
          if (k + 1 != j)
            {
#pragma unroll
              for (int w = 0; w < KW; w++)
                t[(k + 1) * KW + w] = x[w];
            }
        }
      // the run may go on past what was loaded
      if (j >= tn && t0 + tn < n)
        *long_runs = 1;
      hi = max(hi, j);
    }
  if (lo < FT_TILE + TS_HALO + 1) atomicMin(&s_lo, lo);
  if (hi > 0) atomicMax(&s_hi, hi);
  __syncthreads();
  const int wlo = s_lo, whi = s_hi;                        // write back slots [wlo, whi)
  for (int j = wlo * KW + threadIdx.x; j < whi * KW; j += TS_THREADS)
    recs[t0 * KW + j] = t[j];
}

template <int KW>
__global__ __launch_bounds__(TS_THREADS) void k_ts_putback(const u32 *__restrict__ sub,
                                                           const u64 *__restrict__ pos, int64_t m,
                                                           u32 *__restrict__ recs)
{ const int64_t j = (int64_t) blockIdx.x * TS_THREADS + threadIdx.x;
  if (j >= m)
    return;
  const u64 p = pos[j];
#pragma unroll
  for (int w = 0; w < KW; w++)
    recs[p * KW + w] = sub[j * KW + w];
}

// top > 0 (fk_radix.hip): of the non-constant bytes among bytes[] only the `top` most significant are passes
int fkx_lsd_sort_top(fk_ctx *ctx, int64_t nelem, void *d_src, void *d_trg, int rsize, const int *bytes, int nbytes,
                     int top, void **result);

// table != 0: the caller is the table sort (debug knobs of the tests apply; leading bytes are the levels)
template <int KW>
static int msd_sort_t(fk_ctx *ctx, int64_t n, void *d_tab, void *d_tmp, int kb, void **result, int64_t *wfirst, int table)
{ hipStream_t s = ctx->stream;
  const int stride = KW * 4;
  int bytes[64];
  *result = d_tab;
  ctx->sort_stats.passes = 0;
  ctx->sort_stats.nelem = n;
  ctx->sort_stats.rsize = stride;
  ctx->sort_stats.pass_ms_total = ctx->sort_stats.scatter_ms_total = ctx->sort_stats.hist_ms = 0.;
  if (n <= 0 || kb <= 0)
    return (FK_OK);
  int P = 1;
  while (P < 8 && (n >> (8 * P)) > 0)
    P += 1;                                         // ceil(log256 n): at most one record per prefix value on average
  // (one byte more leaves hardly any ties, but the pass it costs is dearer than the longer tie repair: 3.0 G
  //  records, prefix 4 / 5 / 6 bytes: 106 / 114 / 132 ms)
  if (table && ctx->dbg_table_prefix >= 2 && ctx->dbg_table_prefix < kb)
    P = ctx->dbg_table_prefix;
  if (table && ctx->dbg_table_sort >= 2 && ctx->dbg_table_sort < kb)   // tests: a short prefix makes many ties
    P = ctx->dbg_table_sort;
  else if (P >= kb || n < (1 << 20) || (table && ctx->dbg_table_sort == 1))
    P = kb;
  if (P >= kb)
    { // few records (or a key no longer than the depth): every non-constant key byte is a level
      for (int i = 0; i < kb; i++)
        bytes[i] = kb - 1 - i;
      int rc = fkx_lsd_sort(ctx, n, d_tab, d_tmp, stride, bytes, kb, result);
      if (rc == FK_OK && wfirst != NULL)
        for (int x = 0; x < 256; x++)
          wfirst[x] = (int64_t) ctx->h_scratch[x];
      return (rc);
    }
  // the levels: the first P non-constant ones of the leading P + 2 key bytes (the table's leading bytes are never
  // constant; a unit caller's may be)
  const int lead = table ? P : ((P + 2 < kb) ? P + 2 : kb);
  for (int i = 0; i < lead; i++)
    bytes[i] = lead - 1 - i;
  void *sorted = d_tab;
  int rc = fkx_lsd_sort_top(ctx, n, d_tab, d_tmp, stride, bytes, lead, P, &sorted);
  if (rc != FK_OK)
    return (rc);
  const int pb = ctx->rx_top_pbytes;               // the records are in order on their first pb bytes
  const int passes = ctx->sort_stats.passes;
  const double pass_ms = ctx->sort_stats.pass_ms_total, scat_ms = ctx->sort_stats.scatter_ms_total;
  if (wfirst != NULL)
    for (int x = 0; x < 256; x++)
      wfirst[x] = (int64_t) ctx->h_scratch[x];
  *result = sorted;
  if (pb >= kb)
    return (FK_OK);

  // short parts are sorted where they are; only if some part is long does the compaction path run
  u64 *d_tot = ctx->d_scratch + 3072;
  FK_HIP(ctx, hipMemsetAsync(d_tot + 1, 0, 8, s));
  if (((uintptr_t) sorted & 15) == 0)
    { constexpr int IT = TS_FIX_ITEMS(KW);
      hipLaunchKernelGGL((k_ts_fix_tile<KW, IT>), dim3((unsigned) ((n + TS_THREADS * IT - 1) / (TS_THREADS * IT))), dim3(TS_THREADS), 0, s,
                         (u32 *) sorted, n, pb, kb, (u32 *) (d_tot + 1));
    }
  else
    hipLaunchKernelGGL(k_ts_fix<KW>, dim3((unsigned) ((n + TS_THREADS - 1) / TS_THREADS)), dim3(TS_THREADS), 0, s,
                       (u32 *) sorted, n, pb, kb, 48, (u32 *) (d_tot + 1));
  FK_LAUNCH_CHECK(ctx);
  FK_HIP(ctx, hipMemcpyAsync(ctx->h_scratch + 4096, d_tot + 1, 8, hipMemcpyDeviceToHost, s));
  FK_HIP(ctx, hipStreamSynchronize(s));
  ctx->tsort_ties = 0;
  if ((ctx->h_scratch[4096] & 0xffffffffull) == 0 && !(table && ctx->dbg_table_sort >= 2))
    { ctx->sort_stats.passes = passes;
      return (FK_OK);
    }

  const int64_t ntiles = (n + TS_TILE - 1) / TS_TILE;
  u32 *d_cnt = (u32 *) fk_slot(ctx, FK_SLOT_CT_ENT, ntiles * 4);
  u64 *d_off = (u64 *) fk_slot(ctx, FK_SLOT_CT_OFF, ntiles * 8);
  if (d_cnt == NULL || d_off == NULL)
    return (FK_ENOMEM);
  hipLaunchKernelGGL((k_ts_ties<KW, false>), dim3((unsigned) ntiles), dim3(TS_THREADS), 0, s,
                     (const u32 *) sorted, n, pb, d_cnt, (const u64 *) NULL, (u32 *) NULL, (u64 *) NULL);
  hipLaunchKernelGGL(k_exscan_tiles, dim3(1), dim3(256), 0, s, (const u32 *) d_cnt, ntiles, d_off, d_tot);
  FK_LAUNCH_CHECK(ctx);
  FK_HIP(ctx, hipMemcpyAsync(ctx->h_scratch + 4096, d_tot, 8, hipMemcpyDeviceToHost, s));
  FK_HIP(ctx, hipStreamSynchronize(s));
  const int64_t m = (int64_t) ctx->h_scratch[4096];
  ctx->tsort_ties = m;
  if (m == 0)
    return (FK_OK);
  u32 *d_sa = (u32 *) fk_slot(ctx, FK_SLOT_TIE_A, m * stride);
  u32 *d_sb = (u32 *) fk_slot(ctx, FK_SLOT_TIE_B, m * stride);
  u64 *d_ps = (u64 *) fk_slot(ctx, FK_SLOT_TIE_POS, m * 8);
  if (d_sa == NULL || d_sb == NULL || d_ps == NULL)
    return (FK_ENOMEM);
  hipLaunchKernelGGL((k_ts_ties<KW, true>), dim3((unsigned) ntiles), dim3(TS_THREADS), 0, s,
                     (const u32 *) sorted, n, pb, d_cnt, (const u64 *) d_off, d_sa, d_ps);
  FK_LAUNCH_CHECK(ctx);
  for (int i = 0; i < kb; i++)
    bytes[i] = kb - 1 - i;
  void *subsorted = d_sa;
  if ((rc = fkx_lsd_sort(ctx, m, d_sa, d_sb, stride, bytes, kb, &subsorted)) != FK_OK)
    return (rc);
  hipLaunchKernelGGL(k_ts_putback<KW>, dim3((unsigned) ((m + TS_THREADS - 1) / TS_THREADS)), dim3(TS_THREADS), 0, s,
                     (const u32 *) subsorted, (const u64 *) d_ps, m, (u32 *) sorted);
  FK_LAUNCH_CHECK(ctx);
  FK_HIP(ctx, hipStreamSynchronize(s));
  // report the passes over the whole array (what bench.py prices), not the small repair sort
  ctx->sort_stats.passes = passes;
  ctx->sort_stats.pass_ms_total = pass_ms;
  ctx->sort_stats.scatter_ms_total = scat_ms;
  ctx->sort_stats.nelem = n;
  return (FK_OK);
}

/* The MSD engine on records of rsize bytes (a multiple of 4, at most 32): order on the first ksize bytes; d_a and
   d_b are the ping-pong pair, *result points at the sorted one.  1 + P + 1 array passes, P = ceil(log256 n). */
int fkx_msd_sort(fk_ctx *ctx, int64_t n, void *d_a, void *d_b, int rsize, int ksize, void **result)
{ if (rsize <= 0 || (rsize & 3) != 0 || rsize > 32 || ksize < 0 || ksize > rsize)
    { fk_set_error(ctx, "MSD sort: record size %d (multiple of 4, <= 32), key bytes %d", rsize, ksize);
      return (FK_EUNSUPPORTED);
    }
  switch (rsize >> 2)
  { case 1: return msd_sort_t<1>(ctx, n, d_a, d_b, ksize, result, NULL, 0);
    case 2: return msd_sort_t<2>(ctx, n, d_a, d_b, ksize, result, NULL, 0);
    case 3: return msd_sort_t<3>(ctx, n, d_a, d_b, ksize, result, NULL, 0);
    case 4: return msd_sort_t<4>(ctx, n, d_a, d_b, ksize, result, NULL, 0);
    case 5: return msd_sort_t<5>(ctx, n, d_a, d_b, ksize, result, NULL, 0);
    case 6: return msd_sort_t<6>(ctx, n, d_a, d_b, ksize, result, NULL, 0);
    case 7: return msd_sort_t<7>(ctx, n, d_a, d_b, ksize, result, NULL, 0);
    default: return msd_sort_t<8>(ctx, n, d_a, d_b, ksize, result, NULL, 0);
  }
}

/* Sort n table records (kmer_stride bytes each, distinct keys or not) on their KMER_BYTES key.
   d_tab and d_tmp are the ping-pong pair; *result points at the sorted one.  wfirst (may be NULL)
   receives the first-byte census of the records. */
int fkx_sort_table(fk_ctx *ctx, int64_t n, void *d_tab, void *d_tmp, void **result, int64_t *wfirst)
{ const int kb = ctx->wid.kmer_bytes;
  switch (ctx->wid.kmer_stride >> 2)
  { case 1: return msd_sort_t<1>(ctx, n, d_tab, d_tmp, kb, result, wfirst, 1);
    case 2: return msd_sort_t<2>(ctx, n, d_tab, d_tmp, kb, result, wfirst, 1);
    case 3: return msd_sort_t<3>(ctx, n, d_tab, d_tmp, kb, result, wfirst, 1);
    case 4: return msd_sort_t<4>(ctx, n, d_tab, d_tmp, kb, result, wfirst, 1);
    case 5: return msd_sort_t<5>(ctx, n, d_tab, d_tmp, kb, result, wfirst, 1);
    default:
      fk_set_error(ctx, "k-mer stride %d not built", ctx->wid.kmer_stride);
      return (FK_EUNSUPPORTED);
  }
}
