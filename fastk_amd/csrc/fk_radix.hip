// fk_radix.hip -- in-HBM byte radix sort over fixed-width packed records (gfx950).
//
// Replaces the reference's sort engines on the k-mer counting path:
//   LSD_Sort            LSDsort.c:115-271 (lex_thread :55-94)        -> fkx_lsd_sort (same contract)
//   Supermer_Sort /     MSDsort.c:458-489, 536-544 (radix_sort :129-261)
//   Weighted_Kmer_Sort                                                -> same engine, key bytes MSB..LSB
//
// One digit pass = ONE kernel (k_radix_pass): every workgroup takes a tile of records by ticket,
// stages it in LDS with 16-byte coalesced loads, ranks the 8-bit digit of its records with
// wave64 ballots (match-any) into per-wave LDS histograms, obtains the tile's global bin offsets by
// decoupled look-back over 8-byte {epoch,flag,count} status words (relaxed agent-scope atomics, the
// count travels inside the word so no other fence is needed), reorders the tile by digit inside
// LDS and writes every bin's run to HBM as consecutive dwords.  HBM traffic per pass is one read
// and one write of the records (2*n*R bytes) plus 2 KB of status per tile.  Digit histograms for all
// key bytes are taken by one extra read of the records (k_digit_hist) before the first pass.
// The pass is stable, so LSD order over a byte list reproduces LSD_Sort bit for bit.
#include "fk_common.h"

#define RX_THREADS 256
#define RX_WAVES   4

template <int RW> struct RxCfg
{ // tile of ~48 KB so that three workgroups share a CU's 160 KB of LDS
  static constexpr int ITEMS = (RW <= 3) ? 15 : (RW == 4) ? 11 : (RW == 5) ? 9 : (RW == 6) ? 7
                             : (RW == 7) ? 6 : 5;
  static constexpr int TILE  = RX_THREADS * ITEMS;
};

#define ST_AGG  1ull
#define ST_PFX  2ull
#define ST_VAL  ((1ull << 54) - 1)

__device__ __forceinline__ u64 st_pack(u32 epoch, u64 flag, u64 val)
{ return (((u64) epoch) << 56) | (flag << 54) | val; }

// ---------------------------------------------------------------------------------------------
// digit histograms for every byte of the record selected in `want`
template <int RW>
__global__ __launch_bounds__(RX_THREADS) void k_digit_hist(const u32 *__restrict__ src, int64_t n,
                                                           u32 want, u64 *__restrict__ out)
{ __shared__ u32 h[RW * 4 * 256];
  for (int i = threadIdx.x; i < RW * 4 * 256; i += RX_THREADS)
    h[i] = 0;
  __syncthreads();
  for (int64_t i = (int64_t) blockIdx.x * RX_THREADS + threadIdx.x; i < n;
       i += (int64_t) gridDim.x * RX_THREADS)
    { u32 r[RW];
#pragma unroll
      for (int w = 0; w < RW; w++)
        r[w] = src[i * RW + w];
#pragma unroll
      for (int w = 0; w < RW; w++)
#pragma unroll
        for (int b = 0; b < 4; b++)
          if (want & (1u << (w * 4 + b)))
            atomicAdd(&h[(w * 4 + b) * 256 + ((r[w] >> (8 * b)) & 0xffu)], 1u);
    }
  __syncthreads();
  for (int i = threadIdx.x; i < RW * 4 * 256; i += RX_THREADS)
    if (h[i] != 0)
      atomicAdd(&out[i], (u64) h[i]);
}

// ---------------------------------------------------------------------------------------------
// one stable 8-bit digit pass
template <int RW>
__global__ __launch_bounds__(RX_THREADS) void k_radix_pass(const u32 *__restrict__ src,
                                                           u32 *__restrict__ dst, int64_t n,
                                                           int byte_idx,
                                                           const u64 *__restrict__ ghist,
                                                           u64 *status, u32 *ticket, u32 epoch)
{ constexpr int ITEMS = RxCfg<RW>::ITEMS;
  constexpr int TILE  = RxCfg<RW>::TILE;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32      *recs     = (u32 *) smem;                                   // TILE*RW
  int64_t  *goff     = (int64_t *) (smem + (size_t) TILE * RW * 4);    // 256
  u64      *tmp64    = (u64 *) (goff + 256);                           // 8
  volatile u32 *whist = (volatile u32 *) (tmp64 + 8);                  // 4*256
  u32      *binstart = (u32 *) (whist + RX_WAVES * 256);               // 256
  u32      *tmp32    = binstart + 256;                                 // 8
  u32      *s_tile   = tmp32 + 8;                                      // 1

  const int tid  = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;

  if (tid == 0)
    *s_tile = atomicAdd(ticket, 1u);
  for (int i = tid; i < RX_WAVES * 256; i += RX_THREADS)
    whist[i] = 0;
  __syncthreads();

  const u32     tile   = *s_tile;
  const int64_t tstart = (int64_t) tile * TILE;
  const int     tn     = (n - tstart < TILE) ? (int) (n - tstart) : TILE;
  const int     ndw    = tn * RW;

  { const u32   *gsrc = src + tstart * RW;
    const uint4 *g4   = (const uint4 *) gsrc;
    uint4       *l4   = (uint4 *) recs;
    const int    n4   = ndw >> 2;
    for (int i = tid; i < n4; i += RX_THREADS)
      l4[i] = g4[i];
    for (int i = (n4 << 2) + tid; i < ndw; i += RX_THREADS)
      recs[i] = gsrc[i];
  }
  __syncthreads();

  u32 rec[ITEMS][RW];
  u32 dig[ITEMS];
  u32 rnk[ITEMS];
  const int  wbase = wave * 64 * ITEMS;
  const u64  lt    = fk_lanemask_lt();
  const unsigned char *lbytes = (const unsigned char *) smem;

#pragma unroll
  for (int it = 0; it < ITEMS; it++)
    { const int  r     = wbase + it * 64 + lane;
      const bool valid = (r < tn);
      u32 d = 0;
      if (valid)
        {
#pragma unroll
          for (int w = 0; w < RW; w++)
            rec[it][w] = recs[r * RW + w];
          d = lbytes[r * RW * 4 + byte_idx];
        }
      u64 mask = __ballot(valid);
#pragma unroll
      for (int b = 0; b < 8; b++)
        { const bool bit = (d >> b) & 1u;
          const u64  bm  = __ballot(bit);
          mask &= bit ? bm : ~bm;
        }
      const u32 below = (u32) __popcll(mask & lt);
      const u32 cnt   = (u32) __popcll(mask);
      u32 base = 0;
      if (valid)
        base = whist[wave * 256 + d];
      if (valid && below == 0)
        whist[wave * 256 + d] = base + cnt;
      dig[it] = d;
      rnk[it] = base + below;
    }
  __syncthreads();

  // digit `tid`: exclusive prefix over the waves, tile total, bin start, global offset
  { u32 run = 0;
#pragma unroll
    for (int w = 0; w < RX_WAVES; w++)
      { const u32 t = whist[w * 256 + tid];
        whist[w * 256 + tid] = run;
        run += t;
      }
    const u32 total = run;
    u32 tsum;
    const u32 bstart = fk_block_exscan_256<u32>(total, tmp32, &tsum);
    u64 gsum;
    const u64 gbase  = fk_block_exscan_256<u64>(ghist[tid], tmp64, &gsum);
    binstart[tid] = bstart;

    u64 excl = 0;
    if (tile == 0)
      __hip_atomic_store(&status[tid], st_pack(epoch, ST_PFX, total), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    else
      { u64 *mine = status + (size_t) tile * 256 + tid;
        __hip_atomic_store(mine, st_pack(epoch, ST_AGG, total), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        int64_t look = (int64_t) tile - 1;
        while (true)
          { const u64 s = __hip_atomic_load(status + (size_t) look * 256 + tid, __ATOMIC_RELAXED,
                                            __HIP_MEMORY_SCOPE_AGENT);
            const u64 flag = (s >> 54) & 3ull;
            if ((u32) (s >> 56) != epoch || flag == 0)
              { __builtin_amdgcn_s_sleep(1);
                continue;
              }
            excl += (s & ST_VAL);
            if (flag == ST_PFX)
              break;
            look -= 1;
          }
        __hip_atomic_store(mine, st_pack(epoch, ST_PFX, excl + total), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      }
    goff[tid] = (int64_t) (gbase + excl) - (int64_t) bstart;
  }
  __syncthreads();

  // reorder the tile by digit inside LDS (all records are in registers by now)
#pragma unroll
  for (int it = 0; it < ITEMS; it++)
    { const int r = wbase + it * 64 + lane;
      if (r < tn)
        { const u32 d   = dig[it];
          const u32 pos = binstart[d] + whist[wave * 256 + d] + rnk[it];
#pragma unroll
          for (int w = 0; w < RW; w++)
            recs[pos * RW + w] = rec[it][w];
        }
    }
  __syncthreads();

  // every bin's run leaves as consecutive dwords
  for (int j = tid; j < ndw; j += RX_THREADS)
    { const int p = j / RW;
      const int w = j - p * RW;
      const u32 d = lbytes[p * RW * 4 + byte_idx];
      const int64_t g = (goff[d] + p) * RW + w;
      dst[g] = recs[j];
    }
}

template <int RW> static size_t rx_lds_bytes()
{ return ((size_t) RxCfg<RW>::TILE * RW * 4 + 256 * 8 + 8 * 8 + RX_WAVES * 256 * 4 + 256 * 4 + 8 * 4
          + 16);
}

template <int RW>
static int lsd_sort_t(fk_ctx *ctx, int64_t n, void *d_src, void *d_trg, const int *bytes,
                      int nbytes, void **result)
{ constexpr int TILE = RxCfg<RW>::TILE;
  const int64_t ntiles = (n + TILE - 1) / TILE;
  hipStream_t   s = ctx->stream;
  u32 want = 0;

  ctx->sort_stats.passes = 0;
  ctx->sort_stats.nelem  = n;
  ctx->sort_stats.rsize  = RW * 4;
  ctx->sort_stats.pass_ms_total = 0.;
  ctx->sort_stats.hist_ms = 0.;
  *result = d_src;
  if (n == 0 || nbytes == 0)
    return (FK_OK);
  if (nbytes > 60)
    { fk_set_error(ctx, "too many key bytes (%d)", nbytes);
      return (FK_EINVAL);
    }
  for (int i = 0; i < nbytes; i++)
    { if (bytes[i] < 0 || bytes[i] >= RW * 4)
        { fk_set_error(ctx, "key byte %d outside record of %d bytes", bytes[i], RW * 4);
          return (FK_EINVAL);
        }
      want |= (1u << bytes[i]);
    }

  if (ntiles * 256 > ctx->status_cap)
    { if (ctx->d_status != NULL)
        FK_HIP(ctx, hipFree(ctx->d_status));
      ctx->d_status = NULL;
      ctx->status_cap = 0;
      FK_HIP(ctx, hipMalloc((void **) &ctx->d_status, (size_t) ntiles * 256 * 8));
      ctx->status_cap = ntiles * 256;
    }
  FK_HIP(ctx, hipMemsetAsync(ctx->d_status, 0, (size_t) ntiles * 256 * 8, s));
  FK_HIP(ctx, hipMemsetAsync(ctx->d_ticket, 0, 64 * sizeof(u32), s));
  FK_HIP(ctx, hipMemsetAsync(ctx->d_digit_hist, 0, 32 * 256 * sizeof(u64), s));

  FK_HIP(ctx, hipEventRecord(ctx->ev0, s));
  { int64_t nb = (n + RX_THREADS - 1) / RX_THREADS;
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(k_digit_hist<RW>, dim3((unsigned) nb), dim3(RX_THREADS), 0, s,
                       (const u32 *) d_src, n, want, ctx->d_digit_hist);
    FK_LAUNCH_CHECK(ctx);
  }
  FK_HIP(ctx, hipEventRecord(ctx->ev1, s));
  FK_HIP(ctx, hipMemcpyAsync(ctx->h_scratch, ctx->d_digit_hist, (size_t) RW * 4 * 256 * 8,
                             hipMemcpyDeviceToHost, s));
  FK_HIP(ctx, hipStreamSynchronize(s));
  { float ms = 0.f;
    FK_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    ctx->sort_stats.hist_ms = ms;
  }

  static bool attr_set = false;
  if (!attr_set)
    { FK_HIP(ctx, hipFuncSetAttribute((const void *) k_radix_pass<RW>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int) rx_lds_bytes<RW>()));
      attr_set = true;
    }

  u32 *src = (u32 *) d_src, *trg = (u32 *) d_trg;
  int  passes = 0;
  FK_HIP(ctx, hipEventRecord(ctx->ev0, s));
  for (int i = 0; i < nbytes; i++)
    { const u64 *h = ctx->h_scratch + (size_t) bytes[i] * 256;
      bool constant = false;
      for (int x = 0; x < 256; x++)
        if (h[x] == (u64) n)
          constant = true;
      if (constant)
        continue;        // every record carries the same digit: the pass is the identity
      hipLaunchKernelGGL(k_radix_pass<RW>, dim3((unsigned) ntiles), dim3(RX_THREADS),
                         rx_lds_bytes<RW>(), s, (const u32 *) src, trg, n, bytes[i],
                         (const u64 *) (ctx->d_digit_hist + (size_t) bytes[i] * 256),
                         ctx->d_status, ctx->d_ticket + passes, (u32) (passes + 1));
      FK_LAUNCH_CHECK(ctx);
      passes += 1;
      u32 *t = src; src = trg; trg = t;
    }
  FK_HIP(ctx, hipEventRecord(ctx->ev1, s));
  FK_HIP(ctx, hipStreamSynchronize(s));
  { float ms = 0.f;
    FK_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    ctx->sort_stats.pass_ms_total = ms;
    ctx->sort_stats.passes = passes;
  }
  *result = (void *) src;
  return (FK_OK);
}

int fkx_lsd_sort(fk_ctx *ctx, int64_t nelem, void *d_src, void *d_trg, int rsize,
                 const int *bytes, int nbytes, void **result)
{ if (rsize <= 0 || (rsize & 3) != 0 || rsize > 32)
    { fk_set_error(ctx, "record size %d not supported (multiple of 4, <= 32)", rsize);
      return (FK_EUNSUPPORTED);
    }
  switch (rsize >> 2)
  { case 1: return lsd_sort_t<1>(ctx, nelem, d_src, d_trg, bytes, nbytes, result);
    case 2: return lsd_sort_t<2>(ctx, nelem, d_src, d_trg, bytes, nbytes, result);
    case 3: return lsd_sort_t<3>(ctx, nelem, d_src, d_trg, bytes, nbytes, result);
    case 4: return lsd_sort_t<4>(ctx, nelem, d_src, d_trg, bytes, nbytes, result);
    case 5: return lsd_sort_t<5>(ctx, nelem, d_src, d_trg, bytes, nbytes, result);
    case 6: return lsd_sort_t<6>(ctx, nelem, d_src, d_trg, bytes, nbytes, result);
    case 7: return lsd_sort_t<7>(ctx, nelem, d_src, d_trg, bytes, nbytes, result);
    default: return lsd_sort_t<8>(ctx, nelem, d_src, d_trg, bytes, nbytes, result);
  }
}
