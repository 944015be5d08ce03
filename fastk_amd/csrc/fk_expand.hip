// fk_expand.hip -- sorted super-mers -> weighted canonical k-mer records.
//
// Replaces the run detection of count.c:421-426, count_smers (MSDsort.c:381-456, which sizes the
// list) and kmer_list_thread (count.c:339-542): every DISTINCT super-mer with multiplicity ct
// emits its sln+1 k-mers once, each in canonical orientation, carrying weight min(ct,0x7fff);
// the clipped remainder goes to `overflow` exactly as count.c:455-458.
//
// Three kernels: (1) per-tile number of run heads and of k-mers they emit, (2) a single-workgroup
// exclusive scan of the per-tile pairs, (3) the expansion itself -- the tile's records are staged
// in LDS as big-endian words, the head of each run walks forward to get its multiplicity, then
// slides a 2K-bit forward window and its reverse complement over the super-mer, two bits per step.
#include "fk_common.h"

#define EX_THREADS 256
#define EX_ITEMS   4
#define EX_TILE    (EX_THREADS * EX_ITEMS)

template <int RW>
__device__ __forceinline__ bool ex_same(const u32 *a, const u32 *b)
{ bool same = true;
#pragma unroll
  for (int w = 0; w < RW; w++)
    same &= (a[w] == b[w]);
  return same;
}

// ---- (1) per tile: heads and k-mers ----------------------------------------------------------
template <int RW>
__global__ __launch_bounds__(EX_THREADS) void k_ex_count(const u32 *__restrict__ sm, int64_t n,
                                                         int len_byte, u32 *__restrict__ tile_heads,
                                                         u32 *__restrict__ tile_kmers)
{ __shared__ u32 tmp[8];
  const int64_t t0 = (int64_t) blockIdx.x * EX_TILE;
  u32 heads = 0, kmers = 0;
#pragma unroll
  for (int it = 0; it < EX_ITEMS; it++)
    { const int64_t i = t0 + it * EX_THREADS + threadIdx.x;
      if (i < n)
        { const u32 *r = sm + i * RW;
          const bool head = (i == 0) || !ex_same<RW>(r, r - RW);
          if (head)
            { heads += 1;
              kmers += ((r[len_byte >> 2] >> (8 * (len_byte & 3))) & 0xffu) + 1u;
            }
        }
    }
  u32 th, tk;
  (void) fk_block_exscan_256<u32>(heads, tmp, &th);
  (void) fk_block_exscan_256<u32>(kmers, tmp, &tk);
  if (threadIdx.x == 0)
    { tile_heads[blockIdx.x] = th;
      tile_kmers[blockIdx.x] = tk;
    }
}

// ---- (3) expansion ----------------------------------------------------------------------------
__device__ __forceinline__ u32 ex_revpairs(u32 x)
{ const u32 y = __builtin_bitreverse32(x);
  return ((y >> 1) & 0x55555555u) | ((y & 0x55555555u) << 1);
}

// 32 bits starting at bit offset `bit` of a record stored as big-endian value words in LDS
__device__ __forceinline__ u32 ex_bits(const u32 *rec, int bit)
{ const u32 hi = rec[bit >> 5];
  const u32 lo = rec[(bit >> 5) + 1];
  const int sh = bit & 31;
  return (sh == 0) ? hi : ((hi << sh) | (lo >> (32 - sh)));
}

// Phase 1 (one thread per record): run heads, their multiplicity (walk forward in LDS, count.c:421-426)
// and the offset of their first k-mer inside the tile.  Phase 2 (one thread per OUTPUT k-mer, so every
// lane works and stores are consecutive 12-byte records): find the head by binary search over the
// offsets, cut the 2K-bit window out of the super-mer, build its reverse complement, keep the smaller.
template <int RW, int KN>     // KN = words holding a k-mer
__global__ __launch_bounds__(EX_THREADS) void k_ex_expand(const u32 *__restrict__ sm, int64_t n,
                                                          int kmer, int len_byte, int ow,
                                                          const u64 *__restrict__ tile_koff,
                                                          u32 *__restrict__ out,
                                                          u64 *__restrict__ overflow)
{ __shared__ __attribute__((aligned(16))) u32 recs[(EX_TILE + 1) * RW];   // big-endian value words (+1 guard)
  __shared__ u32 hoff[EX_TILE + 1];      // first k-mer of head h inside the tile
  __shared__ uint16_t hrec[EX_TILE];     // record index of head h
  __shared__ uint16_t hct[EX_TILE];      // its clipped multiplicity
  __shared__ u32 tmp[8];
  __shared__ u32 s_runk, s_runh;

  const int64_t t0 = (int64_t) blockIdx.x * EX_TILE;
  const int     tn = (n - t0 < EX_TILE) ? (int) (n - t0) : EX_TILE;
  fk_stage16<(EX_TILE * RW + 1023) / 1024, true>(recs, sm + t0 * RW, tn * RW);
  if (threadIdx.x < RW)
    recs[tn * RW + threadIdx.x] = 0;               // guard word read by ex_bits at the last record
  if (threadIdx.x == 0)
    { s_runk = 0; s_runh = 0; }
  __syncthreads();

#pragma unroll 1
  for (int it = 0; it < EX_ITEMS; it++)
    { const int  l = it * EX_THREADS + threadIdx.x;        // record within the tile
      bool head = false;
      u32  nk = 0;
      if (l < tn)
        { const int64_t i = t0 + l;
          head = (i == 0);
          if (!head)
            { if (l > 0)
                head = !ex_same<RW>(recs + l * RW, recs + (l - 1) * RW);
              else
                { bool same = true;
#pragma unroll
                  for (int w = 0; w < RW; w++)
                    same &= (__builtin_bswap32(sm[(i - 1) * RW + w]) == recs[w]);
                  head = !same;
                }
            }
          if (head)
            nk = ((recs[l * RW + (len_byte >> 2)] >> (24 - 8 * (len_byte & 3))) & 0xffu) + 1u;
        }
      u32 totk, toth;
      const u32 exk = fk_block_exscan_256<u32>(nk, tmp, &totk);
      const u32 exh = fk_block_exscan_256<u32>(head ? 1u : 0u, tmp, &toth);
      const u32 runk = s_runk, runh = s_runh;
      __syncthreads();
      if (threadIdx.x == 0)
        { s_runk = runk + totk; s_runh = runh + toth; }

      if (head)
        { // multiplicity: walk forward while the next record is identical
          const int64_t i = t0 + l;
          u32 mine[RW];
#pragma unroll
          for (int w = 0; w < RW; w++)
            mine[w] = recs[l * RW + w];
          int64_t ct = 1;
          int64_t j = i + 1;
          int     lj = l + 1;
          bool open = true;
          while (open && j < n && lj < tn)
            { bool same = true;
#pragma unroll
              for (int w = 0; w < RW; w++)
                same &= (recs[lj * RW + w] == mine[w]);
              if (same) { ct += 1; j += 1; lj += 1; }
              else open = false;
            }
          while (open && j < n)
            { bool same = true;
#pragma unroll
              for (int w = 0; w < RW; w++)
                same &= (__builtin_bswap32(sm[j * RW + w]) == mine[w]);
              if (same) { ct += 1; j += 1; }
              else open = false;
            }
          if (ct >= 0x8000)                          // count.c:455-458
            { atomicAdd(overflow, (u64) (ct - 0x7fff) * (u64) nk);
              ct = 0x7fff;
            }
          hoff[runh + exh] = runk + exk;
          hrec[runh + exh] = (uint16_t) l;
          hct[runh + exh]  = (uint16_t) ct;
        }
    }
  __syncthreads();

  const u32 nh    = s_runh;
  const u32 ktile = s_runk;
  const int pad   = 32 * KN - 2 * kmer;           // unused low bits of the last k-mer word
  const u32 lastm = (pad == 0) ? 0xffffffffu : ~((1u << pad) - 1u);
  const int cw    = (ow * 4 - 2) >> 2;            // word and shift of the uint16 weight
  const int csh   = 8 * ((ow * 4 - 2) & 3);
  u32 *gout = out + tile_koff[blockIdx.x] * (u64) ow;

  for (u32 j = threadIdx.x; j < ktile; j += EX_THREADS)
    { // head of output k-mer j: last h with hoff[h] <= j
      u32 lo = 0, hi = nh;
      while (hi - lo > 1)
        { const u32 mid = (lo + hi) >> 1;
          if (hoff[mid] <= j) lo = mid; else hi = mid;
        }
      const u32  o   = j - hoff[lo];
      const u32 *rec = recs + (u32) hrec[lo] * RW;
      const u32  ct  = hct[lo];

      u32 f[KN], r[KN];
#pragma unroll
      for (int q = 0; q < KN; q++)
        f[q] = ex_bits(rec, 2 * (int) o + 32 * q);
      f[KN - 1] &= lastm;
      { u32 t[KN];
#pragma unroll
        for (int q = 0; q < KN; q++)
          { u32 c = ~f[KN - 1 - q];
            if (q == 0)
              c &= lastm;
            const u32 y = __builtin_bitreverse32(c);
            t[q] = ((y >> 1) & 0x55555555u) | ((y & 0x55555555u) << 1);
          }
#pragma unroll
        for (int q = 0; q < KN; q++)
          { const u32 h2 = t[q];
            const u32 l2 = (q + 1 < KN) ? t[q + 1] : 0u;
            r[q] = (pad == 0) ? h2 : ((h2 << pad) | (l2 >> (32 - pad)));
          }
      }
      bool use_f = false, decided = false;         // count.c:484-495: forward iff strictly smaller
#pragma unroll
      for (int q = 0; q < KN; q++)
        if (!decided && f[q] != r[q])
          { use_f = (f[q] < r[q]);
            decided = true;
          }
      u32 *dst = gout + j * (u32) ow;
#pragma unroll 1
      for (int q = 0; q < ow; q++)
        { u32 x = 0;
#pragma unroll
          for (int z = 0; z < KN; z++)
            if (q == z)
              x = __builtin_bswap32(use_f ? f[z] : r[z]);
          if (q == cw)
            x |= ct << csh;
          dst[q] = x;
        }
    }
}

// ---------------------------------------------------------------------------------------------
template <int RW>
static int expand_t(fk_ctx *ctx, const void *d_smers, int64_t n, void *d_out, int64_t cap,
                    int64_t *nweighted, int64_t *ndistinct, int64_t *overflow)
{ hipStream_t s = ctx->stream;
  const int   K = ctx->prm.kmer;
  const int   kn = (2 * K + 31) / 32;
  const int   ow = ctx->wid.kmer_stride / 4;
  const int   len_byte = ctx->wid.smer_bytes;        // SLEN_BYTES == 1 for k <= 128
  const int64_t ntiles = (n + EX_TILE - 1) / EX_TILE;

  *nweighted = 0; *ndistinct = 0; *overflow = 0;
  if (n == 0)
    return (FK_OK);

  u32 *d_heads = (u32 *) fk_slot(ctx, FK_SLOT_EX_HEADS, ntiles * 4);
  u32 *d_kmers = (u32 *) fk_slot(ctx, FK_SLOT_EX_KMERS, ntiles * 4);
  u64 *d_koff  = (u64 *) fk_slot(ctx, FK_SLOT_EX_KOFF, ntiles * 8);
  if (d_heads == NULL || d_kmers == NULL || d_koff == NULL)
    return (FK_ENOMEM);
  u64 *d_tot = ctx->d_scratch;     // [0] k-mers, [1] heads (via second scan), [2] overflow

  int rc = FK_OK;
  do
    { hipLaunchKernelGGL(k_ex_count<RW>, dim3((unsigned) ntiles), dim3(EX_THREADS), 0, s,
                         (const u32 *) d_smers, n, len_byte, d_heads, d_kmers);
      // heads total: reuse the scan with d_koff as a throw-away output
      hipLaunchKernelGGL(k_exscan_tiles, dim3(1), dim3(256), 0, s, (const u32 *) d_heads, ntiles,
                         d_koff, d_tot + 1);
      hipLaunchKernelGGL(k_exscan_tiles, dim3(1), dim3(256), 0, s, (const u32 *) d_kmers, ntiles,
                         d_koff, d_tot + 0);
      if (hipGetLastError() != hipSuccess) { rc = FK_EHIP; break; }
      if (hipMemsetAsync(d_tot + 2, 0, 8, s) != hipSuccess) { rc = FK_EHIP; break; }
      if (hipMemcpyAsync(ctx->h_scratch, d_tot, 16, hipMemcpyDeviceToHost, s) != hipSuccess
          || hipStreamSynchronize(s) != hipSuccess)
        { rc = FK_EHIP; break; }
      *nweighted = (int64_t) ctx->h_scratch[0];
      *ndistinct = (int64_t) ctx->h_scratch[1];
      if (d_out == NULL)
        break;
      if (cap < *nweighted)
        { fk_set_error(ctx, "k-mer buffer too small: %lld records needed, %lld given",
                       (long long) *nweighted, (long long) cap);
          rc = FK_EINVAL;
          break;
        }
      switch (kn)
      { case 1: hipLaunchKernelGGL((k_ex_expand<RW, 1>), dim3((unsigned) ntiles), dim3(EX_THREADS), 0, s,
                                   (const u32 *) d_smers, n, K, len_byte, ow, (const u64 *) d_koff,
                                   (u32 *) d_out, d_tot + 2); break;
        case 2: hipLaunchKernelGGL((k_ex_expand<RW, 2>), dim3((unsigned) ntiles), dim3(EX_THREADS), 0, s,
                                   (const u32 *) d_smers, n, K, len_byte, ow, (const u64 *) d_koff,
                                   (u32 *) d_out, d_tot + 2); break;
        case 3: hipLaunchKernelGGL((k_ex_expand<RW, 3>), dim3((unsigned) ntiles), dim3(EX_THREADS), 0, s,
                                   (const u32 *) d_smers, n, K, len_byte, ow, (const u64 *) d_koff,
                                   (u32 *) d_out, d_tot + 2); break;
        case 4: hipLaunchKernelGGL((k_ex_expand<RW, 4>), dim3((unsigned) ntiles), dim3(EX_THREADS), 0, s,
                                   (const u32 *) d_smers, n, K, len_byte, ow, (const u64 *) d_koff,
                                   (u32 *) d_out, d_tot + 2); break;
        default:
          fk_set_error(ctx, "k = %d needs %d k-mer words; only k <= 64 is built", K, kn);
          rc = FK_EUNSUPPORTED;
      }
      if (rc != FK_OK) break;
      if (hipGetLastError() != hipSuccess) { rc = FK_EHIP; break; }
      if (hipMemcpyAsync(ctx->h_scratch, d_tot + 2, 8, hipMemcpyDeviceToHost, s) != hipSuccess
          || hipStreamSynchronize(s) != hipSuccess)
        { rc = FK_EHIP; break; }
      *overflow = (int64_t) ctx->h_scratch[0];
    }
  while (0);
  if (rc == FK_EHIP)
    fk_set_error(ctx, "expand: HIP failure: %s", hipGetErrorString(hipGetLastError()));
  return (rc);
}

int fkx_expand(fk_ctx *ctx, const void *d_smers, int64_t nsuper, void *d_out, int64_t cap,
               int64_t *nweighted, int64_t *ndistinct, int64_t *overflow)
{ switch (ctx->wid.smer_stride >> 2)
  { case 2: return expand_t<2>(ctx, d_smers, nsuper, d_out, cap, nweighted, ndistinct, overflow);
    case 3: return expand_t<3>(ctx, d_smers, nsuper, d_out, cap, nweighted, ndistinct, overflow);
    case 4: return expand_t<4>(ctx, d_smers, nsuper, d_out, cap, nweighted, ndistinct, overflow);
    case 5: return expand_t<5>(ctx, d_smers, nsuper, d_out, cap, nweighted, ndistinct, overflow);
    case 6: return expand_t<6>(ctx, d_smers, nsuper, d_out, cap, nweighted, ndistinct, overflow);
    case 7: return expand_t<7>(ctx, d_smers, nsuper, d_out, cap, nweighted, ndistinct, overflow);
    case 8: return expand_t<8>(ctx, d_smers, nsuper, d_out, cap, nweighted, ndistinct, overflow);
    default:
      fk_set_error(ctx, "super-mer stride %d not built", ctx->wid.smer_stride);
      return (FK_EUNSUPPORTED);
  }
}
