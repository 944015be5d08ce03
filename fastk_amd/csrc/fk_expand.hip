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
#define EX_ITEMS   2
#define EX_TILE    (EX_THREADS * EX_ITEMS)
#define EX_G       6                     // consecutive output k-mers per thread

template <int RW>
__device__ __forceinline__ bool ex_same(const u32 *a, const u32 *b)
{ bool same = true;
#pragma unroll
  for (int w = 0; w < RW; w++)
    same &= (a[w] == b[w]);
  return same;
}

// ---- (1) per tile: heads and k-mers ----------------------------------------------------------
// DD: the records are already de-duplicated (fkx_dedup_supermers): RW dwords + one dword with the
// multiplicity, every record is a run of its own
template <int RW, bool DD>
__global__ __launch_bounds__(EX_THREADS) void k_ex_count(const u32 *__restrict__ sm, int64_t n,
                                                         int len_byte, u32 *__restrict__ tile_heads,
                                                         u32 *__restrict__ tile_kmers)
{ constexpr int RS = RW + (DD ? 1 : 0);
  __shared__ u32 tmp[8];
  const int64_t t0 = (int64_t) blockIdx.x * EX_TILE;
  u32 heads = 0, kmers = 0;
#pragma unroll
  for (int it = 0; it < EX_ITEMS; it++)
    { const int64_t i = t0 + it * EX_THREADS + threadIdx.x;
      if (i < n)
        { const u32 *r = sm + i * RS;
          const bool head = DD || (i == 0) || !ex_same<RW>(r, r - RS);
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
  // one 64-bit shift: with a conditional on sh == 0 the compiler sinks the second load into a
  // divergent branch (an exec-mask round trip and a second LDS wait per word)
  return ((u32) (((((u64) hi) << 32) | (u64) lo) >> (32 - (bit & 31))));
}

// Phase 1 (one thread per record): run heads, their multiplicity (walk forward in LDS, count.c:421-426)
// and the offset of their first k-mer inside the tile.  Phase 2 (one thread per OUTPUT k-mer, so every
// lane works and stores are consecutive 12-byte records): find the head by binary search over the
// offsets, cut the 2K-bit window out of the super-mer, build its reverse complement, keep the smaller.
template <int OW> struct __attribute__((packed, aligned(4))) ex_out { u32 w[OW]; };

// REF (round 6, fk_recut.hip): the tile is EX_TILE REFERENCES to pieces of de-duplicated super-mers, in key order; every
// reference is a run of its own whose record is fetched from sm[] by index, whose k-mers begin `off` k-mers into that
// record and number `n` -- the k-mers of one minimizer domain, so that the W records leave grouped by minimizer key.
template <int RW, int KN, int OW, bool DD, bool REF = false>   // KN = words holding a k-mer, OW = words of an output record
__global__ __launch_bounds__(EX_THREADS) void k_ex_expand(const u32 *__restrict__ sm, int64_t n,
                                                          int kmer, int len_byte,
                                                          const u64 *__restrict__ tile_koff,
                                                          u32 *__restrict__ out,
                                                          u64 *__restrict__ overflow, int64_t ntiles,
                                                          uint8_t *__restrict__ dig, int kbytes,
                                                          const u64 *__restrict__ refs = NULL)
{ static_assert(!REF || DD, "references point at de-duplicated records");
  constexpr int RS = RW + (DD ? 1 : 0);             // dwords per input record
  __shared__ u64     sref[REF ? EX_TILE : 1];       // the tile's references
  __shared__ uint8_t hofs[REF ? EX_TILE : 4];       // first k-mer of head h inside its record
  __shared__ __attribute__((aligned(16))) u32 recs[(EX_TILE + 1) * RS];   // big-endian value words (+1 guard)
  __shared__ uint16_t hoff[EX_TILE + 2]; // first k-mer of head h inside the tile (a tile holds < 2^16 k-mers: 1024 x (k - 6))
  __shared__ uint16_t hrec[EX_TILE];     // record index of head h
  __shared__ uint16_t hct[EX_TILE];      // its clipped multiplicity
  __shared__ u32 tmp[8];
  __shared__ u32 s_runk, s_runh;
  // digit bytes of the k-mers a wave emits in one step (64 * EX_G consecutive ones), staged so that the
  // stream is written as whole dwords: a byte store per k-mer made every 32-byte sector of the stream
  // go to memory EX_G times (WRITE_SIZE 1.43 x the algorithmic W * 13 bytes, profiles/r02_a)
  __shared__ __attribute__((aligned(16))) u32 sdig[EX_THREADS / 64][(64 * EX_G) / 4 + 2];
  // the records of one step of a wave, staged so that they are written as whole lines (records of up to 12 bytes:
  // wider ones would not leave room for two workgroups per CU)
  constexpr bool STAGE = (OW <= 3);
  __shared__ u32 sout[STAGE ? EX_THREADS / 64 : 1][STAGE ? 64 * EX_G * OW : 1];

  // persistent workgroups
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x)
  {
  __syncthreads();
  const int64_t t0 = tile * EX_TILE;
  const int     tn = (n - t0 < EX_TILE) ? (int) (n - t0) : EX_TILE;
  if (REF)
    { for (int l = threadIdx.x; l < tn; l += EX_THREADS)
        sref[l] = refs[t0 + l];
      __syncthreads();
      // every load is issued before the first LDS write (a loop of load / store pairs waits for memory once per turn)
      constexpr int NG = (EX_TILE * RS) / EX_THREADS;
      static_assert((EX_TILE * RS) % EX_THREADS == 0, "whole turns");
      u32 g[NG];
#pragma unroll
      for (int k = 0; k < NG; k++)
        { const int slot = (int) threadIdx.x + k * EX_THREADS;
          const int l = min(slot / RS, tn - 1), wd = slot % RS;
          g[k] = sm[(u64) fk_ref_idx(sref[l]) * RS + wd];
        }
#pragma unroll
      for (int k = 0; k < NG; k++)
        { const int slot = (int) threadIdx.x + k * EX_THREADS;
          if (slot < tn * RS)
            recs[slot] = __builtin_bswap32(g[k]);
        }
    }
  else
  fk_stage16<(EX_TILE * RS + 1023) / 1024, true>(recs, sm + t0 * RS, tn * RS);
  if (threadIdx.x < RS)
    recs[tn * RS + threadIdx.x] = 0;               // guard word read by ex_bits at the last record
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
          head = DD || (i == 0);
          if (!head)
            { if (l > 0)
                head = !ex_same<RW>(recs + l * RS, recs + (l - 1) * RS);
              else
                { bool same = true;
#pragma unroll
                  for (int w = 0; w < RW; w++)
                    same &= (__builtin_bswap32(sm[(i - 1) * RS + w]) == recs[w]);
                  head = !same;
                }
            }
          if (head)
            nk = REF ? fk_ref_n(sref[REF ? l : 0])
                     : ((recs[l * RS + (len_byte >> 2)] >> (24 - 8 * (len_byte & 3))) & 0xffu) + 1u;
        }
      u32 totk, toth;
      const u32 exk = fk_block_exscan_256<u32>(nk, tmp, &totk);
      const u32 exh = fk_block_exscan_256<u32>(head ? 1u : 0u, tmp, &toth);
      const u32 runk = s_runk, runh = s_runh;
      __syncthreads();
      if (threadIdx.x == 0)
        { s_runk = runk + totk; s_runh = runh + toth; }

      if (head)
        { hoff[runh + exh] = (uint16_t) (runk + exk);
          hrec[runh + exh] = (uint16_t) l;
          if (REF)
            hofs[REF ? runh + exh : 0] = (uint8_t) fk_ref_off(sref[REF ? l : 0]);
        }
    }
  __syncthreads();

  // multiplicity of a head = distance to the next head (the records are grouped); only the last run
  // of the tile may go on in the next tiles and is followed there
  { const u32 nh1 = s_runh;
    for (u32 h = threadIdx.x; h < nh1; h += EX_THREADS)
      { const int l = hrec[h];
        int64_t ct = (h + 1 < nh1) ? (int64_t) hrec[h + 1] - l : (int64_t) tn - l;
        if (DD)
          ct = (int64_t) __builtin_bswap32(recs[l * RS + RW]);
        else if (h + 1 == nh1)
          { int64_t j = t0 + tn;
            bool open = true;
            while (open && j < n)
              { bool same = true;
#pragma unroll
                for (int w = 0; w < RW; w++)
                  same &= (__builtin_bswap32(sm[j * RS + w]) == recs[l * RS + w]);
                if (same) { ct += 1; j += 1; }
                else open = false;
              }
          }
        if (ct >= 0x8000)                          // count.c:455-458
          { const u32 nk = REF ? fk_ref_n(sref[REF ? l : 0])        // (a piece accounts for its own k-mers)
                               : ((recs[l * RS + (len_byte >> 2)] >> (24 - 8 * (len_byte & 3))) & 0xffu) + 1u;
            atomicAdd(overflow, (u64) (ct - 0x7fff) * (u64) nk);
            ct = 0x7fff;
          }
        hct[h] = (uint16_t) ct;
      }
  }
  __syncthreads();

  const u32 nh    = s_runh;
  const u32 ktile = s_runk;
  if (threadIdx.x == 0)
    hoff[nh] = (uint16_t) ktile;                  // sentinel: end of the last head's k-mers
  __syncthreads();
  const int padb  = 32 * KN - 2 * kmer;           // unused low bits of the last k-mer word
  const u32 lastm = (padb == 0) ? 0xffffffffu : ~((1u << padb) - 1u);
  const u64 kbase = tile_koff[tile];
  u32 *gout = out + kbase * (u64) OW;

  // every thread emits EX_G CONSECUTIVE k-mers: one binary search for the first, after that the
  // next k-mer costs a fresh forward window (ex_bits) and its reverse complement; moving on to the
  // next head only changes the record the window is cut from
  const u32 lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  for (u32 jw = (threadIdx.x & ~63u) * EX_G; jw < ktile; jw += EX_THREADS * EX_G)     // uniform in a wave
    { const u32 j0 = jw + lane * EX_G;
      u64 dgs = 0;                                  // this thread's digit bytes, first k-mer lowest
      if (j0 < ktile)
      {
      u32 lo = 0, hi = nh;
      while (hi - lo > 1)
        { const u32 mid = (lo + hi) >> 1;
          if ((u32) hoff[mid] <= j0) lo = mid; else hi = mid;
        }
      u32        o   = j0 - hoff[lo] + (REF ? (u32) hofs[REF ? lo : 0] : 0u);
      u32        nxt = hoff[lo + 1];
      const u32 *rec = recs + (u32) hrec[lo] * RS;
      u32        ct  = hct[lo];
      u32 f[KN], r[KN];
#pragma unroll
      for (int g = 0; g < EX_G; g++)
        { const u32 j = j0 + g;
          if (j >= ktile)
            break;
          if (g > 0)
            { if (j == nxt)
                { lo += 1;
                  o = REF ? (u32) hofs[REF ? lo : 0] : 0u;
                  nxt = hoff[lo + 1];
                  rec = recs + (u32) hrec[lo] * RS;
                  ct  = hct[lo];
                }
              else
                o += 1;
            }
#pragma unroll
          for (int q = 0; q < KN; q++)
            f[q] = ex_bits(rec, 2 * (int) o + 32 * q);
          f[KN - 1] &= lastm;
          // reverse complement straight from the forward window, every time: rolling the previous one
          // by a base is cheaper per k-mer, but some lane of a wave enters a new super-mer at almost
          // every step, so the wave would run both forms
          { u32 t[KN];
#pragma unroll
            for (int q = 0; q < KN; q++)
              { u32 c = ~f[KN - 1 - q];
                if (q == 0)
                  c &= lastm;
                t[q] = ex_revpairs(c);
              }
#pragma unroll
            for (int q = 0; q < KN; q++)
              { const u32 h2 = t[q];
                const u32 l2 = (q + 1 < KN) ? t[q + 1] : 0u;
                r[q] = (padb == 0) ? h2 : ((h2 << padb) | (l2 >> (32 - padb)));
              }
          }
          bool use_f = (f[KN - 1] < r[KN - 1]);        // count.c:484-495: forward iff strictly smaller
#pragma unroll
          for (int q = KN - 2; q >= 0; q--)            // (no short-circuit: mask arithmetic, no branches)
            use_f = (f[q] < r[q]) | ((f[q] == r[q]) & use_f);
          ex_out<OW> x;
#pragma unroll
          for (int q = 0; q < OW; q++)
            x.w[q] = (q < KN) ? __builtin_bswap32(use_f ? f[q < KN ? q : 0] : r[q < KN ? q : 0]) : 0u;
          if (dig != NULL)
            { // the k-mer stage groups these records by a hash of their key bytes next: hand it the first
              // digit stream (saves a pass over W).  No histograms beside it: k_rx_superscan sums the digit totals
              // from the tile histograms (round 4) -- the two LDS atomics per k-mer that kept them here were dead work
              u32 ha, hb;
              fk_rec_hash<OW>(x.w, kbytes, ha, hb);
              dgs |= (u64) (hb & 0xffu) << (8 * g);
            }
          x.w[OW - 1] |= ct << 16;                     // uint16 weight in the record's last two bytes
          if (STAGE)
            {
#pragma unroll
              for (int q = 0; q < OW; q++)
                sout[wv][lane * (EX_G * OW) + g * OW + q] = x.w[q];
            }
          else
            *(ex_out<OW> *) (gout + (u64) j * OW) = x;
        }
      }
      if (STAGE)
        { // the wave's 64 * EX_G records leave as whole lines: lane l stores dwords l, l + 64, ... of the wave's
          // stretch.  (A 12-byte store per record and lane, 72 bytes apart, is 64 separate write requests per
          // instruction: the kernel ran 281 us per launch at 1/10 of configs[2] and 197 us with the same bytes sent
          // to coalesced addresses.)
          __builtin_amdgcn_wave_barrier();             // (LDS operations of a wave execute in order)
          const u32 nvd = ((ktile - jw < 64u * EX_G) ? (ktile - jw) : 64u * EX_G) * OW;
          u32 *gw = gout + (u64) jw * OW;
#pragma unroll
          for (int t = 0; t < EX_G * OW; t++)
            { const u32 idx = (u32) t * 64u + lane;
              if (idx < nvd)
                gw[idx] = sout[wv][idx];
            }
          __builtin_amdgcn_wave_barrier();
        }
      if (dig != NULL)
        { // the wave's digits: bytes jw .. jw + nv - 1 of the tile's stretch of the stream
          static_assert(EX_G == 6, "three 16-bit stores per thread");
          uint16_t *s16 = (uint16_t *) sdig[wv] + lane * (EX_G / 2);
          s16[0] = (uint16_t) dgs; s16[1] = (uint16_t) (dgs >> 16); s16[2] = (uint16_t) (dgs >> 32);
          __builtin_amdgcn_wave_barrier();             // (LDS operations of a wave execute in order)
          const u32 nv = (ktile - jw < 64u * EX_G) ? (ktile - jw) : 64u * EX_G;
          uint8_t  *gd = dig + kbase + jw;
          const uint8_t *sb = (const uint8_t *) sdig[wv];
          u32 head = (u32) ((4u - ((u32) (uintptr_t) gd & 3u)) & 3u);
          if (head > nv) head = nv;
          if (lane < head)
            gd[lane] = sb[lane];
          const u32 ndw = (nv - head) >> 2;
          for (u32 t = lane; t < ndw; t += 64)
            { const u32 x0 = sdig[wv][t + (head >> 2)], x1 = sdig[wv][t + (head >> 2) + 1];
              const u32 sh = 8u * (head & 3u);
              *(u32 *) (gd + head + 4 * t) = sh ? ((x0 >> sh) | (x1 << (32 - sh))) : x0;
            }
          const u32 done = head + 4 * ndw;
          if (lane < nv - done)
            gd[done + lane] = sb[done + lane];
          __builtin_amdgcn_wave_barrier();
        }
    }
  }
}

#if !defined(FK_HOST_EMU) || defined(FK_EMU_FULL)      // (the CPU tests run the kernels above through tests/csrc/hip_emu.h; what follows talks to the HIP runtime)
// ---------------------------------------------------------------------------------------------
template <int RW>
static int expand_t(fk_ctx *ctx, const void *d_smers, int64_t n, void *d_out, int64_t cap,
                    int64_t *nweighted, int64_t *ndistinct, int64_t *overflow, bool reuse_counts,
                    bool hash_stream, bool dedup)
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
    { if (reuse_counts && d_out != NULL)
        { // the sizing call just before this one left the per-tile offsets in d_koff
          *nweighted = ctx->ex_nweighted;
          *ndistinct = ctx->ex_ndistinct;
          if (hipMemsetAsync(d_tot + 2, 0, 8, s) != hipSuccess) { rc = FK_EHIP; break; }
        }
      else
      {
      if (dedup)
        hipLaunchKernelGGL((k_ex_count<RW, true>), dim3((unsigned) ntiles), dim3(EX_THREADS), 0, s,
                           (const u32 *) d_smers, n, len_byte, d_heads, d_kmers);
      else
        hipLaunchKernelGGL((k_ex_count<RW, false>), dim3((unsigned) ntiles), dim3(EX_THREADS), 0, s,
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
      ctx->ex_nweighted = *nweighted;
      ctx->ex_ndistinct = *ndistinct;
      }
      if (d_out == NULL)
        break;
      if (cap < *nweighted)
        { fk_set_error(ctx, "k-mer buffer too small: %lld records needed, %lld given",
                       (long long) *nweighted, (long long) cap);
          rc = FK_EINVAL;
          break;
        }
#define EX_LAUNCH(KN, OW)                                                                          \
    do {                                                                                             \
      if (dedup)                                                                                       \
        hipLaunchKernelGGL((k_ex_expand<RW, KN, OW, true>), dim3((unsigned) std::min<int64_t>(ntiles, egrid)), \
                           dim3(EX_THREADS), 0, s,                                                      \
                           (const u32 *) d_smers, n, K, len_byte, (const u64 *) d_koff, (u32 *) d_out, \
                           d_tot + 2, ntiles, d_dig, ctx->wid.kmer_bytes);                             \
      else                                                                                             \
        hipLaunchKernelGGL((k_ex_expand<RW, KN, OW, false>), dim3((unsigned) std::min<int64_t>(ntiles, egrid)), \
                           dim3(EX_THREADS), 0, s,                                                      \
                           (const u32 *) d_smers, n, K, len_byte, (const u64 *) d_koff, (u32 *) d_out, \
                           d_tot + 2, ntiles, d_dig, ctx->wid.kmer_bytes);                             \
    } while (0)
      // hash_stream: also emit what the hashed grouping of the k-mers needs (the stream of hash digit 0)
      uint8_t *d_dig = NULL;
      const int64_t egrid = 8ll * (ctx->num_cus > 0 ? ctx->num_cus : 256);
      if (hash_stream)
        { d_dig = (uint8_t *) fk_slot(ctx, FK_SLOT_DIG_A, *nweighted + 64);
          if (d_dig == NULL) { rc = FK_ENOMEM; break; }
        }
      if (ow != kn && ow != kn + 1)
        { fk_set_error(ctx, "k = %d: %d k-mer words do not fit records of %d words", K, kn, ow);
          rc = FK_EUNSUPPORTED;
          break;
        }
      switch (kn)
      { case 1: if (ow == 1) EX_LAUNCH(1, 1); else EX_LAUNCH(1, 2); break;
        case 2: if (ow == 2) EX_LAUNCH(2, 2); else EX_LAUNCH(2, 3); break;
        case 3: if (ow == 3) EX_LAUNCH(3, 3); else EX_LAUNCH(3, 4); break;
        case 4: if (ow == 4) EX_LAUNCH(4, 4); else EX_LAUNCH(4, 5); break;
        default:
          fk_set_error(ctx, "k = %d needs %d k-mer words; only k <= 64 is built", K, kn);
          rc = FK_EUNSUPPORTED;
      }
#undef EX_LAUNCH
      if (rc != FK_OK) break;
      if (hipGetLastError() != hipSuccess) { rc = FK_EHIP; break; }
      if (hipMemcpyAsync(ctx->h_scratch, d_tot + 2, 8, hipMemcpyDeviceToHost, s) != hipSuccess
          || hipStreamSynchronize(s) != hipSuccess)
        { rc = FK_EHIP; break; }
      *overflow = (int64_t) ctx->h_scratch[0];
      if (hash_stream)
        ctx->pre_hist_n = *nweighted;
    }
  while (0);
  if (rc == FK_EHIP)
    fk_set_error(ctx, "expand: HIP failure: %s", hipGetErrorString(hipGetLastError()));
  return (rc);
}

// ---- expansion in the order of sorted references (fk_recut.hip) --------------------------------------------------
template <int RW>
static int expand_refs_t(fk_ctx *ctx, const void *d_dd, const u64 *d_refs, int64_t nref, void *d_out, int64_t cap,
                         int64_t *nweighted, int64_t *overflow, const u64 **koff_out)
{ hipStream_t s = ctx->stream;
  const int   K = ctx->prm.kmer;
  const int   kn = (2 * K + 31) / 32;
  const int   ow = ctx->wid.kmer_stride / 4;
  const int   len_byte = ctx->wid.smer_bytes;
  const int64_t ntiles = (nref + EX_TILE - 1) / EX_TILE;
  *nweighted = 0; *overflow = 0;
  u32 *d_kmers = (u32 *) fk_slot(ctx, FK_SLOT_EX_KMERS, ntiles * 4);
  u64 *d_koff  = (u64 *) fk_slot(ctx, FK_SLOT_EX_KOFF, ntiles * 8);
  if (d_kmers == NULL || d_koff == NULL)
    return (FK_ENOMEM);
  u64 *d_tot = ctx->d_scratch;     // [0] k-mers, [2] overflow
  int rc = fkx_ref_count(ctx, d_refs, nref, d_kmers);
  if (rc != FK_OK) return (rc);
  hipLaunchKernelGGL(k_exscan_tiles, dim3(1), dim3(256), 0, s, (const u32 *) d_kmers, ntiles, d_koff, d_tot + 0);
  FK_LAUNCH_CHECK(ctx);
  FK_HIP(ctx, hipMemsetAsync(d_tot + 2, 0, 8, s));
  FK_HIP(ctx, hipMemcpyAsync(ctx->h_scratch, d_tot, 8, hipMemcpyDeviceToHost, s));
  FK_HIP(ctx, hipStreamSynchronize(s));
  *nweighted = (int64_t) ctx->h_scratch[0];
  *koff_out = d_koff;
  if (cap < *nweighted)
    { fk_set_error(ctx, "k-mer buffer too small: %lld records needed, %lld given", (long long) *nweighted, (long long) cap);
      return (FK_EINVAL);
    }
  if (ow != kn && ow != kn + 1)
    { fk_set_error(ctx, "k = %d: %d k-mer words do not fit records of %d words", K, kn, ow);
      return (FK_EUNSUPPORTED);
    }
  const int64_t egrid = 8ll * (ctx->num_cus > 0 ? ctx->num_cus : 256);
#define EXR_LAUNCH(KN, OW)                                                                           \
    hipLaunchKernelGGL((k_ex_expand<RW, KN, OW, true, true>), dim3((unsigned) std::min<int64_t>(ntiles, egrid)), \
                       dim3(EX_THREADS), 0, s, (const u32 *) d_dd, nref, K, len_byte, (const u64 *) d_koff, \
                       (u32 *) d_out, d_tot + 2, ntiles, (uint8_t *) NULL, ctx->wid.kmer_bytes, d_refs)
  switch (kn)
  { case 2: if (ow == 2) EXR_LAUNCH(2, 2); else EXR_LAUNCH(2, 3); break;
    case 3: if (ow == 3) EXR_LAUNCH(3, 3); else EXR_LAUNCH(3, 4); break;
    case 4: if (ow == 4) EXR_LAUNCH(4, 4); else EXR_LAUNCH(4, 5); break;
    default:
      fk_set_error(ctx, "k = %d needs %d k-mer words; the reference expansion is built for k = 32 ... 64", K, kn);
      return (FK_EUNSUPPORTED);
  }
#undef EXR_LAUNCH
  FK_LAUNCH_CHECK(ctx);
  FK_HIP(ctx, hipMemcpyAsync(ctx->h_scratch, d_tot + 2, 8, hipMemcpyDeviceToHost, s));
  FK_HIP(ctx, hipStreamSynchronize(s));
  *overflow = (int64_t) ctx->h_scratch[0];
  return (FK_OK);
}

/* The W weighted k-mer records of the nref pieces d_refs points at (sorted by key, fkx_recut) inside the de-duplicated
   records d_dd, written in that order; *d_koff = records in front of every tile of EX_TILE references (for
   fkx_ref_bounds). */
int fkx_expand_refs(fk_ctx *ctx, const void *d_dd, int64_t nsx, const u64 *d_refs, int64_t nref, void *d_out, int64_t cap,
                    int64_t *nweighted, int64_t *overflow, const u64 **d_koff)
{ (void) nsx;
  static_assert(EX_TILE == 512, "fk_recut.hip counts k-mers per RF_TILE = 512 references");
  switch (ctx->wid.smer_stride >> 2)
  { case 4: return expand_refs_t<4>(ctx, d_dd, d_refs, nref, d_out, cap, nweighted, overflow, d_koff);
    case 5: return expand_refs_t<5>(ctx, d_dd, d_refs, nref, d_out, cap, nweighted, overflow, d_koff);
    case 6: return expand_refs_t<6>(ctx, d_dd, d_refs, nref, d_out, cap, nweighted, overflow, d_koff);
    case 7: return expand_refs_t<7>(ctx, d_dd, d_refs, nref, d_out, cap, nweighted, overflow, d_koff);
    default:
      fk_set_error(ctx, "super-mer stride %d: no reference expansion", ctx->wid.smer_stride);
      return (FK_EUNSUPPORTED);
  }
}

int fkx_expand(fk_ctx *ctx, const void *d_smers, int64_t nsuper, void *d_out, int64_t cap,
               int64_t *nweighted, int64_t *ndistinct, int64_t *overflow, bool reuse_counts,
               bool hash_stream, bool dedup)
{ switch (ctx->wid.smer_stride >> 2)
  { case 1: return expand_t<1>(ctx, d_smers, nsuper, d_out, cap, nweighted, ndistinct, overflow, reuse_counts, hash_stream, dedup);   // (k = 8)
    case 2: return expand_t<2>(ctx, d_smers, nsuper, d_out, cap, nweighted, ndistinct, overflow, reuse_counts, hash_stream, dedup);
    case 3: return expand_t<3>(ctx, d_smers, nsuper, d_out, cap, nweighted, ndistinct, overflow, reuse_counts, hash_stream, dedup);
    case 4: return expand_t<4>(ctx, d_smers, nsuper, d_out, cap, nweighted, ndistinct, overflow, reuse_counts, hash_stream, dedup);
    case 5: return expand_t<5>(ctx, d_smers, nsuper, d_out, cap, nweighted, ndistinct, overflow, reuse_counts, hash_stream, dedup);
    case 6: return expand_t<6>(ctx, d_smers, nsuper, d_out, cap, nweighted, ndistinct, overflow, reuse_counts, hash_stream, dedup);
    case 7: return expand_t<7>(ctx, d_smers, nsuper, d_out, cap, nweighted, ndistinct, overflow, reuse_counts, hash_stream, dedup);
    case 8: return expand_t<8>(ctx, d_smers, nsuper, d_out, cap, nweighted, ndistinct, overflow, reuse_counts, hash_stream, dedup);
    default:
      fk_set_error(ctx, "super-mer stride %d not built", ctx->wid.smer_stride);
      return (FK_EUNSUPPORTED);
  }
}
#endif   // FK_HOST_EMU
