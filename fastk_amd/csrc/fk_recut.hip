// fk_recut.hip -- the weighted k-mer list grouped WITHOUT moving it: distinct super-mers are cut into the domains of a
// second, longer minimizer and only 8-byte references to the pieces are sorted (round 6).
//
// What this replaces.  kmer_list_thread (count.c:339-542) writes every k-mer of every distinct super-mer to the slot
// its first byte selects (cursors from count_smers' khist, MSDsort.c:381-456), so that Weighted_Kmer_Sort
// (MSDsort.c:536-544) starts one level down.  Until round 5 the device wrote the W weighted records in super-mer
// order and then moved all of them twice (two hashed 8-bit passes, 39 % of the pipeline's algorithmic bytes at BASELINE
// configs[2]) only to bring equal k-mers into one LDS fill of the aggregation (fk_aggr.hip).
//
// Here equal k-mers meet because of WHERE they are written.  Every k-mer x has M(x) = the smallest hash of the canonical
// 16-mers inside it: a function of the k-mer alone, the same for x and its reverse complement, and constant over runs of
// ~13 consecutive k-mers of a read (the domain of a minimizer).  A distinct super-mer (at most k - 4 k-mers, 24 bytes
// with its multiplicity) is cut where M changes -- 2.1 pieces on average -- and each piece becomes a REFERENCE
//     [ 22 key bits = a mix of M | 28 bits: which super-mer | 7 bits: first k-mer | 7 bits: k-mers ]
// of 8 bytes.  The references are sorted on the key (three 8-bit passes over 8-byte records: a twelfth of the bytes the
// two passes over W moved), the expansion walks them in that order and fetches each piece's super-mer (fk_expand.hip,
// REF), and the W records come out with all copies of a k-mer inside one key group.  Fills of the aggregation are then
// cut at group boundaries, packed to 15/16 of the LDS table whatever the groups' sizes (k_ref_bounds): a census on
// BASELINE configs[2]'s data shape (profiles/r06_minbin_census.txt) shows single 16-bit minimizer bins 19 % larger than a
// fill one time in five -- why the bins are not used as they come but packed from 4 M fine groups.
//
// Result-invariance: the multiset of (canonical k-mer, weight) records is the one the linear expansion writes; only
// their order differs, and the aggregation sums per k-mer (MSDsort.c:491-509 semantics, unchanged).
#include "fk_common.h"

#include <algorithm>

#define RC_THREADS 256
#define RC_STAGE   2048                  // references a workgroup collects in LDS before it reserves room for them

template <int RW> struct RcCfg
{ static constexpr int LMAX = (RW * 4 - 1) * 4;      // bases a record of RW words holds in front of its length byte
  static constexpr int PMAX = LMAX - FK_REF_MLEN + 1; // 16-mer starts
  static constexpr int NMAX = (LMAX + 5) / 2 - 4;     // k-mers of a super-mer: at most k - 4, and 2 k - 5 <= LMAX bases
};

__device__ __forceinline__ u32 rc_revpairs(u32 x)
{ const u32 y = __builtin_bitreverse32(x);
  return (((y >> 1) & 0x55555555u) | ((y & 0x55555555u) << 1));
}

// order of the canonical 16-mers: a bijective mix of the 32-bit code (xor, odd multiply, xor-shift -- ONE multiply: a
// 32-bit integer multiply costs four issue slots and the kernel spends one per 16-mer start), so that poly-A (code 0)
// is not the smallest and low-complexity sequence does not flood one key.  Two different 16-mers never tie.
__device__ __forceinline__ u32 rc_rank(u32 c)
{ u32 x = (c ^ 0x5bd1e995u) * 0x9E3779B1u;
  return (x ^ (x >> 15));
}

template <int N> struct __attribute__((packed, aligned(4))) rc_rec { u32 w[N]; };

// One thread per distinct super-mer (RW dwords in the reference's byte order + a dword with its multiplicity).
//   hash of every 16-mer start, rolled: the forward code is a funnel shift of two record words, the reverse complement
//   takes the complement of the entering base at its top;
//   window minimum over w = k - 15 starts by doubling in registers (1, 2, 4, ... starts) and one look d = w - 2^j further
//   along the thread's own column in LDS;
//   a piece begins where the minimum changes; its reference goes to the workgroup's stage and leaves in whole lines.
// scal: [0] write cursor (references)  [1] overflow of `cap`  [2] references made (statistics)
template <int RW>
__global__ __launch_bounds__(RC_THREADS) void k_recut(const u32 *__restrict__ dd, int64_t n, int kmer, int len_byte,
                                                      u64 *__restrict__ out, u64 cap, u64 *__restrict__ scal,
                                                      int64_t ntiles)
{ constexpr int RS   = RW + 1;
  constexpr int PMAX = RcCfg<RW>::PMAX;
  constexpr int NMAX = RcCfg<RW>::NMAX;
  FK_DYN_LDS(u32, rc_col);                           // [rows][RC_THREADS]: the thread's window minima, by start
  __shared__ u64 stage[RC_STAGE];
  __shared__ u32 s_cnt;
  __shared__ u64 s_base;
  const int tid = threadIdx.x;
  const int w   = kmer - FK_REF_MLEN + 1;            // 16-mer starts inside a k-mer
  const int lg  = 31 - __builtin_clz((unsigned) w);
  const int d   = w - (1 << lg);
  u32 *col = rc_col + tid;
  if (tid == 0)
    s_cnt = 0;
  __syncthreads();

  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x)
    { const int64_t i = tile * RC_THREADS + tid;
      u32 v[RW + 1];
      int nk = 0;
      if (i < n)
        { const rc_rec<RW> r = *(const rc_rec<RW> *) (dd + i * RS);
#pragma unroll
          for (int q = 0; q < RW; q++)
            v[q] = __builtin_bswap32(r.w[q]);
          nk = (int) ((v[len_byte >> 2] >> (24 - 8 * (len_byte & 3))) & 0xffu) + 1;
        }
      else
        {
#pragma unroll
          for (int q = 0; q < RW; q++)
            v[q] = 0;
        }
      v[RW] = 0;

      // ---- ranks of the 16-mer starts.  Starts behind the super-mer's last one hold whatever the padding gives: they lie
      // in no window of a k-mer that exists (window j ends at start j + w - 1 <= nk + w - 2).
      u32 A[PMAX];
      { u32 f = v[0];
        u32 rc = rc_revpairs(~f);
        A[0] = rc_rank(min(f, rc));
#pragma unroll
        for (int p = 1; p < PMAX; p++)
          { const int q = p >> 4, s = p & 15;
            f  = (s == 0) ? v[q] : __funnelshift_l(v[q + 1], v[q], 2 * s);
            rc = __funnelshift_r(rc, ~f, 2);         // the complement of the entering base in front
            A[p] = rc_rank(min(f, rc));
          }
      }
      // ---- minima over 2^lg consecutive starts, in place (ascending p: A[p + s] is still the previous level's)
#pragma unroll
      for (int s = 1; s <= 32; s <<= 1)
        if (s < (1 << lg))                           // (uniform)
          {
#pragma unroll
            for (int p = 0; p + s < PMAX; p++)
              A[p] = min(A[p], A[p + s]);
          }
      // ---- the thread's column: window j = min(A[j], A[j + d]); a piece begins where that value changes
#pragma unroll
      for (int p = 0; p < PMAX; p++)
        col[p * RC_THREADS] = A[p];
      u64 starts = 0;
      { u32 prev = 0;
#pragma unroll
        for (int j = 0; j < NMAX; j++)
            { const u32 far = col[(min(j + d, PMAX - 1)) * RC_THREADS];
              const u32 m = min(A[j], far);
              col[j * RC_THREADS] = m;               // (rows <= j are no longer read as A)
              if (j < nk && (j == 0 || m != prev))
                starts |= (1ull << j);
              prev = m;
            }
      }
      // ---- the pieces
      { const u32 idx = (u32) (i & ((1ll << FK_REF_IDX_BITS) - 1));
        u64 m = starts;
        while (m != 0)
          { const int j = __builtin_ctzll(m);
            m &= m - 1;
            const int e = (m != 0) ? __builtin_ctzll(m) : nk;
            const u32 key = (col[j * RC_THREADS] * 0x9E3779B1u) >> (32 - FK_REF_KEY_BITS);
            const u64 ref = fk_ref_pack(key, idx, (u32) j, (u32) (e - j));
            const u32 slot = atomicAdd(&s_cnt, 1u);
            if (slot < RC_STAGE)
              stage[slot] = ref;
            else
              { const u64 g = atomicAdd((unsigned long long *) &scal[0], 1ull);      // (a tile of very short pieces)
                if (g < cap) out[g] = ref;
                else         scal[1] = 1;
              }
          }
      }
      __syncthreads();
      const u32 cnt = s_cnt;
      __syncthreads();                               // (everybody has read it before the next tile adds to it)
      const bool last = (tile + gridDim.x >= ntiles);
      if (cnt + 1024 > RC_STAGE || last)             // (uniform) not enough room for a usual tile: the stage leaves
        { const u32 T = (cnt < RC_STAGE) ? cnt : RC_STAGE;
          if (tid == 0)
            { s_base = atomicAdd((unsigned long long *) &scal[0], (unsigned long long) T);
              atomicAdd((unsigned long long *) &scal[2], (unsigned long long) cnt);
            }
          __syncthreads();
          const u64 base = s_base;
          for (u32 t = (u32) tid; t < T; t += RC_THREADS)
            { if (base + t < cap) out[base + t] = stage[t];
              else                scal[1] = 1;
            }
          __syncthreads();
          if (tid == 0)
            s_cnt = 0;
          __syncthreads();
        }
    }
}

// ---- k-mers per tile of EX_TILE references (the expansion's offsets) -----------------------------------------------
#define RF_TILE 512
__global__ __launch_bounds__(256) void k_ref_count(const u64 *__restrict__ refs, int64_t n, u32 *__restrict__ tile_kmers)
{ __shared__ u32 tmp[8];
  const int64_t t0 = (int64_t) blockIdx.x * RF_TILE;
  u32 km = 0;
#pragma unroll
  for (int it = 0; it < RF_TILE / 256; it++)
    { const int64_t i = t0 + it * 256 + threadIdx.x;
      if (i < n)
        km += fk_ref_n(refs[i]);
    }
  u32 tot;
  (void) fk_block_exscan_256<u32>(km, tmp, &tot);
  if (threadIdx.x == 0)
    tile_kmers[blockIdx.x] = tot;
}

// ---- the fills of the aggregation ----------------------------------------------------------------------------------
// bounds[f] (f = 0 .. nfills) = the first record of fill f among the W weighted k-mers the expansion writes in reference
// order: the first KEY GROUP that begins at or behind record f * target (bounds[nfills] = W).  All copies of a k-mer
// share a key, so no k-mer is cut by a bound; a fill holds target + one group's records at most, and a group beyond the
// LDS table (a minimizer of repetitive sequence) is taken in chunks by the aggregation like any bin that is too large.
__device__ __forceinline__ u64 rb_prefix(const u64 *refs, const u64 *koff, int64_t r)
{ // records in front of reference r
  const int64_t t = r / RF_TILE;
  u64 s = koff[t];
  for (int64_t i = t * RF_TILE; i < r; i++)
    s += fk_ref_n(refs[i]);
  return (s);
}

__global__ __launch_bounds__(256) void k_ref_bounds(const u64 *__restrict__ refs, int64_t nref, const u64 *__restrict__ koff,
                                                    int64_t ntiles, u64 W, u32 target, int64_t nfills,
                                                    u64 *__restrict__ bounds)
{ const int64_t f = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (f > nfills) return;
  if (f == nfills) { bounds[f] = W; return; }
  if (f == 0)      { bounds[f] = 0; return; }
  const u64 g = (u64) f * target;                    // (< W: nfills = ceil(W / target))
  // the tile that holds record g, then the reference
  int64_t lo = 0, hi = ntiles;                       // koff[lo] <= g < koff[hi]  (koff[ntiles] = W, not stored)
  while (hi - lo > 1)
    { const int64_t mid = (lo + hi) >> 1;
      if (koff[mid] <= g) lo = mid; else hi = mid;
    }
  int64_t r = lo * RF_TILE;
  u64 s = koff[lo];
  for (;;)
    { if (r >= nref) { bounds[f] = W; return; }      // (cannot happen while the offsets describe these references)
      const u64 nk = fk_ref_n(refs[r]);
      if (s + nk > g) break;
      s += nk;
      r += 1;
    }
  // reference r holds record g (s = records in front of r).  A group that begins exactly here is the answer.
  const u32 key = fk_ref_key(refs[r]);
  if (s == g && (r == 0 || fk_ref_key(refs[r - 1]) != key))
    { bounds[f] = g; return; }
  // else the next group: the first reference behind r with another key (the references are in key order: gallop, bisect)
  int64_t a = r, step = 1, b;
  for (;;)
    { b = a + step;
      if (b >= nref) { b = nref; break; }
      if (fk_ref_key(refs[b]) != key) break;
      a = b; step <<= 1;
    }
  while (b - a > 1)                                  // key(a) == key, (b == nref or key(b) != key)
    { const int64_t mid = (a + b) >> 1;
      if (fk_ref_key(refs[mid]) == key) a = mid; else b = mid;
    }
  bounds[f] = (b >= nref) ? W : rb_prefix(refs, koff, b);
}

#if !defined(FK_HOST_EMU) || defined(FK_EMU_FULL)      // (the CPU tests run the kernels above through tests/csrc/hip_emu.h; what follows talks to the HIP runtime)
// ---------------------------------------------------------------------------------------------------------------------
bool fkx_recut_applies(const fk_ctx *ctx, int64_t nsx)
{ const int rw = ctx->wid.smer_stride >> 2;
  return (ctx->prm.kmer >= 32 && ctx->prm.kmer <= 64 && rw >= 4 && rw <= 7 && nsx > 0 && nsx < (1ll << FK_REF_IDX_BITS)
          && ctx->dbg_kmer_stage == 0 && ctx->dbg_aggr_engine != 1 && ctx->dbg_aggr_variant == 0);
}

template <int RW>
static int recut_t(fk_ctx *ctx, const void *d_dd, int64_t nsx, u64 *d_out, int64_t cap, int64_t *nref)
{ hipStream_t s = ctx->stream;
  const int K = ctx->prm.kmer;
  u64 *d_scal = ctx->d_scratch + 6000;
  FK_HIP(ctx, hipMemsetAsync(d_scal, 0, 4 * sizeof(u64), s));
  const int64_t ntiles = (nsx + RC_THREADS - 1) / RC_THREADS;
  const size_t lds = (size_t) RcCfg<RW>::PMAX * RC_THREADS * sizeof(u32);
  static bool attr_set[16] = { false };
  if (!attr_set[RW])
    { auto kern = k_recut<RW>;
      FK_HIP(ctx, hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
      attr_set[RW] = true;
    }
  const int cus = ctx->num_cus > 0 ? ctx->num_cus : 256;
  const int64_t grid = std::min<int64_t>(ntiles, 2ll * cus);
  hipLaunchKernelGGL(k_recut<RW>, dim3((unsigned) grid), dim3(RC_THREADS), lds, s, (const u32 *) d_dd, nsx, K,
                     (int) ctx->wid.smer_bytes, d_out, (u64) cap, d_scal, ntiles);
  FK_LAUNCH_CHECK(ctx);
  FK_HIP(ctx, hipMemcpyAsync(ctx->h_scratch + 6000, d_scal, 4 * sizeof(u64), hipMemcpyDeviceToHost, s));
  FK_HIP(ctx, hipStreamSynchronize(s));
  if (ctx->h_scratch[6001] != 0 || (int64_t) ctx->h_scratch[6000] > cap)
    return (FK_ESTATE);                              // more pieces than room: the caller groups the W records instead
  *nref = (int64_t) ctx->h_scratch[6000];
  return (FK_OK);
}

/* d_dd: nsx de-duplicated super-mer records (fkx_dedup_supermers).  Leaves the references to their pieces, sorted by
   key, in *d_refs (a slot of the context) and their number in *nref.  FK_ESTATE: the pieces outran the room given to
   them (three per super-mer + slack) -- nothing is lost, the caller takes the hashed grouping of the W records. */
int fkx_recut(fk_ctx *ctx, const void *d_dd, int64_t nsx, u64 **d_refs, int64_t *nref)
{ *d_refs = NULL; *nref = 0;
  const int64_t cap = 3 * nsx + 65536;
  u64 *ra = (u64 *) fk_slot(ctx, FK_SLOT_REF_A, cap * 8);
  u64 *rb = (u64 *) fk_slot(ctx, FK_SLOT_REF_B, cap * 8);
  if (ra == NULL || rb == NULL)
    return (FK_ENOMEM);
  int rc;
  switch (ctx->wid.smer_stride >> 2)
  { case 4: rc = recut_t<4>(ctx, d_dd, nsx, ra, cap, nref); break;
    case 5: rc = recut_t<5>(ctx, d_dd, nsx, ra, cap, nref); break;
    case 6: rc = recut_t<6>(ctx, d_dd, nsx, ra, cap, nref); break;
    case 7: rc = recut_t<7>(ctx, d_dd, nsx, ra, cap, nref); break;
    default: return (FK_EUNSUPPORTED);
  }
  if (rc != FK_OK)
    return (rc);
  static const int bytes[3] = { 5, 6, 7 };           // the key's 22 bits are the top of the 64-bit word
  void *sorted = ra;
  if ((rc = fkx_lsd_sort(ctx, *nref, ra, rb, 8, bytes, 3, &sorted)) != FK_OK)
    return (rc);
  *d_refs = (u64 *) sorted;
  return (FK_OK);
}

/* The fills of the aggregation for W records written in the order of the nref sorted references: d_koff = the
   expansion's per-tile record offsets (RF_TILE references per tile).  *d_bounds (slot FK_SLOT_AG_BOUNDS) gets
   *nfills + 1 record positions. */
int fkx_ref_bounds(fk_ctx *ctx, const u64 *d_refs, int64_t nref, const u64 *d_koff, int64_t W, int target,
                   u64 **d_bounds, int64_t *nfills)
{ hipStream_t s = ctx->stream;
  const int64_t nf = (W + target - 1) / target;
  u64 *b = (u64 *) fk_slot(ctx, FK_SLOT_AG_BOUNDS, std::max<int64_t>(nf + 1, 65537) * 8);
  if (b == NULL)
    return (FK_ENOMEM);
  const int64_t ntiles = (nref + RF_TILE - 1) / RF_TILE;
  hipLaunchKernelGGL(k_ref_bounds, dim3((unsigned) ((nf + 1 + 255) / 256)), dim3(256), 0, s, d_refs, nref, d_koff, ntiles,
                     (u64) W, (u32) target, nf, b);
  FK_LAUNCH_CHECK(ctx);
  *d_bounds = b;
  *nfills = nf;
  return (FK_OK);
}

int fkx_ref_count(fk_ctx *ctx, const u64 *d_refs, int64_t nref, u32 *d_tile_kmers)
{ const int64_t ntiles = (nref + RF_TILE - 1) / RF_TILE;
  if (ntiles > 0)
    hipLaunchKernelGGL(k_ref_count, dim3((unsigned) ntiles), dim3(256), 0, ctx->stream, d_refs, nref, d_tile_kmers);
  FK_LAUNCH_CHECK(ctx);
  return (FK_OK);
}
#endif   // FK_HOST_EMU
