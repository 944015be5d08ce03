// fk_radix.hip -- in-HBM byte radix sort over fixed-width packed records (gfx950).
//
// Replaces the reference's sort engines on the k-mer counting path:
//   LSD_Sort            LSDsort.c:115-271 (lex_thread :55-94)        -> fkx_lsd_sort (same contract)
//   Supermer_Sort /     MSDsort.c:458-489, 536-544 (radix_sort :129-261)
//   Weighted_Kmer_Sort                                                -> same engine, key bytes MSB..LSB
//
// The product engine is the dependency-free "stream" digit pass further down (k_rx_tilehist /
// k_rx_chunkscan / k_rx_superscan / k_rx_scatter): every pass also writes the digit the NEXT pass sorts
// on next to each record, so a pass gets its per-tile bin offsets from an n-byte stream without touching
// the records and without any inter-workgroup dependency.  HBM traffic per pass is one read and one write
// of the records (2*n*R bytes) plus 2n bytes of digit stream.  Every pass is stable, so LSD order over a
// byte list reproduces LSD_Sort bit for bit.
//
// (The first engine -- ONE kernel per pass with decoupled look-back over status words, 2.3 TB/s against this one's
// 4.5, and its ablated variants -- was deleted in round 6; DESIGN.md section 4 keeps what was learnt from it.  -DFK_ABLATION
// (make ABLATION=1) still builds the scatter kernels' own ablation bits, RX_ABL_*, for tools/scatter_ablation.py.)
#include "fk_common.h"
#include <type_traits>

#define RX_THREADS 256
#define RX_WAVES   4

template <int RW> struct RxCfg
{ // default records per thread: a tile of ~36 KB plus its permutation and histograms lets three
  // workgroups share a CU's 160 KB of LDS
  static constexpr int ITEMS = (RW <= 2) ? 16 : (RW == 3) ? 12 : (RW == 4) ? 8 : (RW == 5) ? 8 : 4;
};

// a record as one object of RW dwords, 4-byte aligned: copies compile to dwordxN loads/stores
template <int RW> struct __attribute__((packed, aligned(4))) rx_rec { u32 w[RW]; };

#define ST_AGG  1ull
#define ST_PFX  2ull
#define ST_VAL  ((1ull << 54) - 1)

__device__ __forceinline__ u64 st_pack(u32 epoch, u64 flag, u64 val)
{ return (((u64) epoch) << 56) | (flag << 54) | val; }

// ---------------------------------------------------------------------------------------------
// 64-bit mix of a whole record, for GROUPING identical records (fkx_group): the digit of pass p is
// byte p of the hash, so five passes make identical records adjacent whatever their width.
template <int RW>
__device__ __forceinline__ u32 rx_hash_digit(const u32 *r, int byte_idx, int hbytes)
{ u32 a, b;
  fk_rec_hash<RW>(r, hbytes, a, b);
  const u32 x = (byte_idx < 4) ? b : a;
  return (x >> (8 * (byte_idx & 3))) & 0xffu;
}

template <int RW>
__global__ __launch_bounds__(RX_THREADS) void k_hash_hist(const u32 *__restrict__ src, int64_t n,
                                                          int nbytes, u64 *__restrict__ out,
                                                          uint8_t *__restrict__ dig, int dig_byte,
                                                          int hbytes)
{ __shared__ u32 h[8 * 256];
  for (int i = threadIdx.x; i < 8 * 256; i += RX_THREADS)
    h[i] = 0;
  __syncthreads();
  for (int64_t i = (int64_t) blockIdx.x * RX_THREADS + threadIdx.x; i < n;
       i += (int64_t) gridDim.x * RX_THREADS)
    { u32 r[RW];
      { const rx_rec<RW> rr = *(const rx_rec<RW> *) (src + i * RW);      // one load per record, not RW
#pragma unroll
        for (int w = 0; w < RW; w++)
          r[w] = rr.w[w];
      }
      for (int b = 0; b < nbytes; b++)
        atomicAdd(&h[b * 256 + rx_hash_digit<RW>(r, b, hbytes)], 1u);
      if (dig != NULL)
        dig[i] = (uint8_t) rx_hash_digit<RW>(r, dig_byte, hbytes);
    }
  __syncthreads();
  for (int i = threadIdx.x; i < 8 * 256; i += RX_THREADS)
    if (h[i] != 0)
      atomicAdd(&out[i], (u64) h[i]);
}

// ---------------------------------------------------------------------------------------------
// digit histograms for every byte of the record selected in `want`
template <int RW>
__global__ __launch_bounds__(RX_THREADS) void k_digit_hist(const u32 *__restrict__ src, int64_t n,
                                                           u32 want, u64 *__restrict__ out,
                                                           uint8_t *__restrict__ dig, int dig_byte)
{ __shared__ u32 h[RW * 4 * 256];
  for (int i = threadIdx.x; i < RW * 4 * 256; i += RX_THREADS)
    h[i] = 0;
  __syncthreads();
  for (int64_t i = (int64_t) blockIdx.x * RX_THREADS + threadIdx.x; i < n;
       i += (int64_t) gridDim.x * RX_THREADS)
    { u32 r[RW];
      { const rx_rec<RW> rr = *(const rx_rec<RW> *) (src + i * RW);
#pragma unroll
        for (int w = 0; w < RW; w++)
          r[w] = rr.w[w];
      }
#pragma unroll
      for (int w = 0; w < RW; w++)
#pragma unroll
        for (int b = 0; b < 4; b++)
          if (want & (1u << (w * 4 + b)))
            atomicAdd(&h[(w * 4 + b) * 256 + ((r[w] >> (8 * b)) & 0xffu)], 1u);
      if (dig != NULL)
        {
#pragma unroll
          for (int w = 0; w < RW; w++)
            if ((dig_byte >> 2) == w)
              dig[i] = (uint8_t) ((r[w] >> (8 * (dig_byte & 3))) & 0xffu);
        }
    }
  __syncthreads();
  for (int i = threadIdx.x; i < RW * 4 * 256; i += RX_THREADS)
    if (h[i] != 0)
      atomicAdd(&out[i], (u64) h[i]);
}

// =============================================================================================
// Dependency-free digit pass ("stream" engine, the default).
//
// The look-back above costs ~30 % of a pass on this chip (status words cross eight non-coherent
// XCD L2s) and forces tiles to be taken in ticket order.  Here every pass ALSO writes, next to each
// record's new position, the digit the NEXT pass will sort on (one byte per record).  The next pass
// then gets its per-tile bin offsets without touching the records and without any inter-workgroup
// dependency:
//   k_rx_tilehist   reads the n-byte digit stream; per chunk of RX_CH tiles it writes, per tile and
//                   bin, the running count inside the chunk (u16) and the chunk totals (u32)
//   k_rx_chunkscan  running totals of chunks inside a super-chunk of RX_SC chunks (u32) + super totals
//   k_rx_superscan  exclusive scan over super-chunks and bins -> absolute bases (u64)
//   k_rx_scatter    load tile, rank (as above), offset = super + chunk + tile prefix, scatter, and
//                   emit the next digit stream.  Workgroup b works on tile (b%8)*ceil(T/8) + b/8,
//                   so each XCD (b % 8 under the observed dispatch) streams one contiguous range of
//                   tiles and neighbouring bin runs meet in the same L2.
// HBM bytes per pass: 2*n*R (records) + 2*n (digit stream) + ~1.4 % tables.
#define RX_CH 16
#define RX_SC 256
// -DFK_ABLATION builds (WRONG output; tools/scatter_ablation.py): bits of the scatter kernels' `unstable` argument, set
// from fk_debug_set("scatter_abl", bits) -- what a pass would cost without one of its parts
#define RX_ABL_NOHASH 0x100     // hashed passes: the next pass's digit is a byte of the record (another one every pass), not a hash of it
#define RX_ABL_LINEAR 0x200     // the records leave in tile order (whole lines, no scatter): the bound of any write combining
#define RX_ABL_NOPERM 0x400     // sorted slot p takes record p (no LDS gather)
#define RX_ABL_NORANK 0x800     // no ranking: every record gets rank 0 of its wave's bin (no ballots, no LDS atomics)
#ifdef FK_ABLATION
#define RX_ABL_BITS(ctx) ((ctx)->dbg_scatter_abl & 0xf00)
#else
#define RX_ABL_BITS(ctx) 0
#endif

template <int ITEMS>
__global__ __launch_bounds__(RX_THREADS) void k_rx_tilehist(const uint8_t *__restrict__ dig, int64_t n,
                                                            uint16_t *__restrict__ tilepfx,
                                                            u32 *__restrict__ chunktot)
{ constexpr int TILE = RX_THREADS * ITEMS;
  constexpr int NW   = ITEMS / 4;
  __shared__ u32 h[256];
  const int     tid   = threadIdx.x;
  const int64_t tile0 = (int64_t) blockIdx.x * RX_CH;
  u32 run = 0;
  h[tid] = 0;
  __syncthreads();

  u32 cur[NW], nxt[NW];
  auto load = [&](int64_t tile, u32 *v)
    { const int64_t base = tile * TILE + (int64_t) tid * ITEMS;
#pragma unroll
      for (int k = 0; k < NW; k++)
        { const int64_t o = base + 4 * k;
          u32 x = 0;
          if (o + 4 <= n)
            x = *(const u32 *) (dig + o);
          else
            for (int b = 0; b < 4; b++)
              if (o + b < n)
                x |= ((u32) dig[o + b]) << (8 * b);
          v[k] = x;
        }
    };
  load(tile0, cur);
  for (int t = 0; t < RX_CH; t++)
    { const int64_t tile = tile0 + t;
      if (tile * TILE >= n)
        break;
      if (t + 1 < RX_CH && (tile + 1) * TILE < n)
        load(tile + 1, nxt);
      const int64_t base = tile * TILE + (int64_t) tid * ITEMS;
#pragma unroll
      for (int k = 0; k < NW; k++)
#pragma unroll
        for (int b = 0; b < 4; b++)
          if (base + 4 * k + b < n)
            atomicAdd(&h[(cur[k] >> (8 * b)) & 0xffu], 1u);
      __syncthreads();
      tilepfx[tile * 256 + tid] = (uint16_t) run;
      run += h[tid];
      h[tid] = 0;
      __syncthreads();
#pragma unroll
      for (int k = 0; k < NW; k++)
        cur[k] = nxt[k];
    }
  chunktot[(int64_t) blockIdx.x * 256 + tid] = run;
}

__global__ __launch_bounds__(RX_THREADS) void k_rx_chunkscan(const u32 *__restrict__ chunktot,
                                                             int64_t nchunks,
                                                             u32 *__restrict__ chunkpfx,
                                                             u64 *__restrict__ supertot)
{ const int     tid = threadIdx.x;
  const int64_t c0  = (int64_t) blockIdx.x * RX_SC;
  u32 run = 0;
  // 32 loads in flight per thread: the kernel is a chain of round trips to memory (37 workgroups at configs[2]), and
  // with 8 per round it took 53 us, 196 times a step
  for (int j = 0; j < RX_SC; j += 32)
    { u32 v[32];
#pragma unroll
      for (int k = 0; k < 32; k++)
        v[k] = (c0 + j + k < nchunks) ? chunktot[(c0 + j + k) * 256 + tid] : 0u;
#pragma unroll
      for (int k = 0; k < 32; k++)
        if (c0 + j + k < nchunks)
          { chunkpfx[(c0 + j + k) * 256 + tid] = run;
            run += v[k];
          }
    }
  supertot[(int64_t) blockIdx.x * 256 + tid] = run;
}

__global__ __launch_bounds__(RX_THREADS) void k_rx_superscan(const u64 *__restrict__ ghist,
                                                             const u64 *__restrict__ supertot,
                                                             int64_t nsuper, u64 *__restrict__ superpfx)
{ __shared__ u64 tmp[8];
  const int tid = threadIdx.x;
  // the digit's totals are the sums of the super-chunk totals (a few dozen rows): no histogram of the digit has to
  // exist beside its stream -- which is what lets a producer of the records (the splitter, the expansion) hand over
  // nothing but the stream
  (void) ghist;
  u64 tot = 0;
  for (int64_t sc = 0; sc < nsuper; sc++)
    tot += supertot[sc * 256 + tid];
  u64 gsum;
  u64 run = fk_block_exscan_256<u64>(tot, tmp, &gsum);
  for (int64_t sc = 0; sc < nsuper; sc++)
    { superpfx[sc * 256 + tid] = run;
      run += supertot[sc * 256 + tid];
    }
}

template <int RW, int ITEMS, bool HASHED>
__global__ __launch_bounds__(RX_THREADS) void k_rx_scatter(const u32 *__restrict__ src,
                                                           u32 *__restrict__ dst, int64_t n,
                                                           int byte_idx, int next_byte,
                                                           const uint16_t *__restrict__ tilepfx,
                                                           const u32 *__restrict__ chunkpfx,
                                                           const u64 *__restrict__ superpfx,
                                                           const uint8_t *__restrict__ curdig,
                                                           uint8_t *__restrict__ nextdig,
                                                           int64_t ntiles, int hbytes, int unstable)
{ constexpr int TILE = RX_THREADS * ITEMS;

  FK_DYN_LDS_ALIGNED(unsigned char, smem, 16);
  u32      *recs     = (u32 *) smem;                                   // TILE*RW, never reordered
  int64_t  *goff     = (int64_t *) (smem + (size_t) TILE * RW * 4);    // 256
  u64      *tmp64    = (u64 *) (goff + 256);                           // 8
  u32      *whist    = (u32 *) (tmp64 + 8);                            // 4*256
  u32      *binstart = whist + RX_WAVES * 256;                         // 256
  u32      *tmp32    = binstart + 256;                                 // 8
  u32      *pad      = tmp32 + 8;                                      // 4
  uint16_t *perm     = (uint16_t *) (pad + 4);                         // TILE: sorted slot -> record
  uint8_t  *tdig     = (uint8_t *) (perm + TILE);                      // TILE: this pass's digit per record

  const int tid  = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;

  const int64_t per  = (ntiles + 7) >> 3;
  const int64_t tile = (int64_t) (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (tile >= ntiles)
    return;
  for (int i = tid; i < RX_WAVES * 256; i += RX_THREADS)
    whist[i] = 0;

  const int64_t tstart = tile * TILE;
  const int     tn     = (n - tstart < TILE) ? (int) (n - tstart) : TILE;
  const int     ndw    = tn * RW;

  { const u32   *gsrc = src + tstart * RW;
    const uint4 *g4   = (const uint4 *) gsrc;
    uint4       *l4   = (uint4 *) recs;
    if (tn == TILE)
      { static_assert((ITEMS * RW) % 4 == 0, "tile must be a whole number of 16-byte loads per thread");
        constexpr int NV = ITEMS * RW / 4;
        uint4 v[NV];
#pragma unroll
        for (int k = 0; k < NV; k++)
          v[k] = g4[tid + k * RX_THREADS];
#pragma unroll
        for (int k = 0; k < NV; k++)
          l4[tid + k * RX_THREADS] = v[k];
      }
    else
      { const int n4 = ndw >> 2;
        for (int i = tid; i < n4; i += RX_THREADS)
          l4[i] = g4[i];
        for (int i = (n4 << 2) + tid; i < ndw; i += RX_THREADS)
          recs[i] = gsrc[i];
      }
  }
  // the digits of this pass come from the stream the previous pass (or the histogram kernel) wrote
  if (tn == TILE)
    { const u32 *gd = (const u32 *) (curdig + tstart) + tid * (ITEMS / 4);
      u32 *ld = (u32 *) tdig + tid * (ITEMS / 4);
#pragma unroll
      for (int k = 0; k < ITEMS / 4; k++)
        ld[k] = gd[k];
    }
  else
    for (int i = tid; i < tn; i += RX_THREADS)
      tdig[i] = curdig[tstart + i];
  // this tile's bin offsets do not depend on any other workgroup
  const u64 gpre = superpfx[(tile / (RX_CH * RX_SC)) * 256 + tid]
                 + (u64) chunkpfx[(tile / RX_CH) * 256 + tid] + (u64) tilepfx[tile * 256 + tid];
  __syncthreads();

  // Everything after the loads exists twice: for a full tile (all but the last one) every "is this
  // slot inside the tile" predicate is a compile-time true and its exec-mask bookkeeping disappears.
  auto pass = [&](auto full_tile)
  { constexpr bool FULL = decltype(full_tile)::value;
  const int  wbase = wave * 64 * ITEMS;
  const u64  lt    = fk_lanemask_lt();
  const unsigned char *lbytes = (const unsigned char *) smem;

  u32 info[ITEMS];
  u32 old[ITEMS];
  // unstable (first pass of a hashed grouping: nothing depends on the order inside a bin): one LDS
  // atomic per record instead of the 8-ballot match -- the kernel is bound by instruction issue
#ifdef FK_ABLATION
  if (unstable & RX_ABL_NORANK)
    {
#pragma unroll
      for (int it = 0; it < ITEMS; it++)
        info[it] = (u32) tdig[wbase + it * 64 + lane];
    }
  else
#endif
  if (HASHED && (unstable & 1))
    {
#pragma unroll
      for (int it = 0; it < ITEMS; it++)
        { const int  r     = wbase + it * 64 + lane;
          const bool valid = FULL || (r < tn);
          const u32  d     = valid ? (u32) tdig[r] : 0u;
          const u32  rk    = valid ? atomicAdd(&whist[wave * 256 + d], 1u) : 0u;
          info[it] = d | (rk << 8);
        }
    }
  else
  {
#pragma unroll
  for (int it = 0; it < ITEMS; it++)
    { const int  r     = wbase + it * 64 + lane;
      const bool valid = FULL || (r < tn);
      const u32  d     = valid ? (u32) tdig[r] : 0u;
      u64 mask = __ballot(valid);
#pragma unroll
      for (int b = 0; b < 8; b++)
        { const bool bit = (d >> b) & 1u;
          const u64  bm  = __ballot(bit);
          mask &= bit ? bm : ~bm;
        }
      const u32 below  = (u32) __popcll(mask & lt);
      const u32 leader = valid ? (u32) (__ffsll((unsigned long long) mask) - 1) : (u32) lane;
      info[it] = d | (below << 8) | (leader << 16);
      old[it] = 0;
      if (valid && below == 0)
        old[it] = atomicAdd(&whist[wave * 256 + d], (u32) __popcll(mask));
    }
#pragma unroll
  for (int it = 0; it < ITEMS; it++)
    { const u32 e    = info[it];
      const u32 base = (u32) __shfl((int) old[it], (int) ((e >> 16) & 0xffu), 64);
      info[it] = (e & 0xffu) | ((base + ((e >> 8) & 0xffu)) << 8);     // d | rank-in-wave << 8
    }
  }
  __syncthreads();

  { u32 run = 0;
#pragma unroll
    for (int w = 0; w < RX_WAVES; w++)
      { const u32 t = whist[w * 256 + tid];
        whist[w * 256 + tid] = run;
        run += t;
      }
    u32 tsum;
    const u32 bstart = fk_block_exscan_256<u32>(run, tmp32, &tsum);
    binstart[tid] = bstart;
    goff[tid] = (int64_t) gpre - (int64_t) bstart;
  }
  __syncthreads();

#pragma unroll
  for (int it = 0; it < ITEMS; it++)
    { const int r = wbase + it * 64 + lane;
      if (FULL || r < tn)
        { const u32 e   = info[it];
          const u32 d   = e & 0xffu;
          const u32 pos = binstart[d] + whist[wave * 256 + d] + (e >> 8);
          perm[pos] = (uint16_t) r;
        }
    }
  __syncthreads();

  // one thread per record: sorted slot p -> source record, one RW-dword store at its new position
  // (lanes of a wave cover consecutive slots, i.e. consecutive addresses inside a bin's run), and
  // the digit the next pass sorts on next to it
#pragma unroll
  for (int it = 0; it < ITEMS; it++)
    { const int p = it * RX_THREADS + tid;
      if (FULL || p < tn)
        {
#ifdef FK_ABLATION
          const int     sr = (unstable & RX_ABL_NOPERM) ? p : perm[p];
          const u32     d  = tdig[sr];
          const int64_t o  = (unstable & RX_ABL_LINEAR) ? tstart + p : goff[d] + p;
#else
          const int     sr = perm[p];
          const u32     d  = tdig[sr];
          const int64_t o  = goff[d] + p;
#endif
          rx_rec<RW> r = *(const rx_rec<RW> *) (recs + sr * RW);
          *(rx_rec<RW> *) (dst + o * RW) = r;
          if (next_byte >= 0)
            { u32 nd = HASHED ? rx_hash_digit<RW>(r.w, next_byte, hbytes)
                              : (u32) lbytes[sr * RW * 4 + next_byte];
#ifdef FK_ABLATION
              if (HASHED && (unstable & RX_ABL_NOHASH)) nd = (r.w[0] >> (8 * (next_byte & 3))) & 0xffu;   // (a byte that differs from pass to pass: the same byte every time would leave the records SORTED after two passes and the later ones would write whole lines -- the first version of this ablation measured that, not the hash)
#endif
              nextdig[o] = (uint8_t) nd;
            }
        }
    }
  };
  if (tn == TILE)
    pass(std::true_type{});
  else
    pass(std::false_type{});
}

// ---------------------------------------------------------------------------------------------
// Wide variant of k_rx_scatter: 1024 threads and a tile of 1024*ITEMS records per workgroup (one
// workgroup per CU, ~140 KB of LDS), so a bin's run inside a tile is 2.7x longer (R = 12: 32 records
// = 384 bytes instead of 144) and most of the scattered writes are whole 128-byte lines.  With one
// workgroup per CU nothing else would hide the load latency, so the workgroup is persistent: it
// walks over its tiles and fetches the NEXT tile (records, digits, bin offsets) into registers
// before it ranks and writes the current one.  Workgroup b belongs to XCD b % 8 and takes tiles
// (b%8)*ceil(T/8) + b/8 + k*(grid/8): the 32 workgroups of an XCD work on neighbouring tiles.
#define RXW_THREADS 1024
#define RXW_WAVES   16

template <int RW> struct RxCfgW
{ static constexpr int ITEMS = (RW == 1) ? 16 : (RW <= 3) ? 8 : 4; };

template <int RW, int ITEMS, bool HASHED>
__global__ __launch_bounds__(RXW_THREADS) void k_rx_scatter_w(const u32 *__restrict__ src,
                                                              u32 *__restrict__ dst, int64_t n,
                                                              int next_byte,
                                                              const uint16_t *__restrict__ tilepfx,
                                                              const u32 *__restrict__ chunkpfx,
                                                              const u64 *__restrict__ superpfx,
                                                              const uint8_t *__restrict__ curdig,
                                                              uint8_t *__restrict__ nextdig,
                                                              int64_t ntiles, int hbytes, int unstable,
                                                              const uint8_t *__restrict__ carry)
{ // carry != NULL: the digit the NEXT pass sorts on exists already, record by record in the order of src (the producer
  // of the records hashed them once and wrote both digits); it travels with the records instead of being hashed again
  constexpr int TILE = RXW_THREADS * ITEMS;
  static_assert((RX_CH - 1) * TILE < 65536 || RW < 4, "u16 tile prefixes inside a chunk");
  constexpr int NV   = ITEMS * RW / 4;
  constexpr int ND   = ITEMS / 4;
  static_assert((ITEMS * RW) % 4 == 0 && ITEMS % 4 == 0, "tile must split into 16-byte and 4-byte loads");

  FK_DYN_LDS_ALIGNED(unsigned char, smem, 16);
  u32      *recs     = (u32 *) smem;                                   // TILE*RW, never reordered
  int64_t  *goff     = (int64_t *) (smem + (size_t) TILE * RW * 4);    // 256
  u32      *whist    = (u32 *) (goff + 256);                           // RXW_WAVES*256
  u32      *binstart = whist + RXW_WAVES * 256;                        // 256
  u32      *tmp32    = binstart + 256;                                 // 8
  uint16_t *perm     = (uint16_t *) (tmp32 + 8);                       // TILE: sorted slot -> record
  uint8_t  *tdig     = (uint8_t *) (perm + TILE);                      // TILE: this pass's digit per record
  uint8_t  *tcar     = tdig + TILE;                                    // TILE: the next pass's, when it is carried

  const int tid  = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const u64 lt   = fk_lanemask_lt();
  const unsigned char *lbytes = (const unsigned char *) smem;

  const int64_t per   = (ntiles + 7) >> 3;
  const int64_t first = (int64_t) (blockIdx.x >> 3);
  const int64_t step  = (int64_t) (gridDim.x >> 3);
  const int64_t tbase = (int64_t) (blockIdx.x & 7) * per;

  uint4 v[NV];
  u32   dg[ND], dc[ND];
  u64   gp = 0;
#pragma unroll
  for (int q = 0; q < NV; q++)
    v[q] = make_uint4(0, 0, 0, 0);
#pragma unroll
  for (int q = 0; q < ND; q++)
    dg[q] = dc[q] = 0;
#define RXW_FETCH(T)                                                                              \
  { const int64_t t_ = (T);                                                                       \
    if ((t_ + 1) * TILE <= n)                                                                     \
      { const uint4 *g4 = (const uint4 *) (src + t_ * TILE * RW);                                 \
        _Pragma("unroll")                                                                         \
        for (int q = 0; q < NV; q++)                                                              \
          v[q] = g4[tid + q * RXW_THREADS];                                                       \
        const u32 *gd = (const u32 *) (curdig + t_ * TILE) + tid * ND;                            \
        _Pragma("unroll")                                                                         \
        for (int q = 0; q < ND; q++)                                                              \
          dg[q] = gd[q];                                                                          \
        if (carry != NULL)                                                                        \
          { const u32 *gc = (const u32 *) (carry + t_ * TILE) + tid * ND;                         \
            _Pragma("unroll")                                                                     \
            for (int q = 0; q < ND; q++)                                                          \
              dc[q] = gc[q];                                                                      \
          }                                                                                       \
      }                                                                                           \
    if (tid < 256)                                                                                \
      gp = superpfx[(t_ / (RX_CH * RX_SC)) * 256 + tid] + (u64) chunkpfx[(t_ / RX_CH) * 256 + tid] \
         + (u64) tilepfx[t_ * 256 + tid];                                                         \
  }

  int64_t k = first;
  if (k < per && tbase + k < ntiles)
    RXW_FETCH(tbase + k)
  for (; k < per && tbase + k < ntiles; k += step)
    { const int64_t tile   = tbase + k;
      const int64_t tstart = tile * TILE;
      const int     tn     = (n - tstart < TILE) ? (int) (n - tstart) : TILE;
      const int     ndw    = tn * RW;

      if (tn == TILE)
        { uint4 *l4 = (uint4 *) recs;
#pragma unroll
          for (int j = 0; j < NV; j++)
            l4[tid + j * RXW_THREADS] = v[j];
          u32 *ld = (u32 *) tdig + tid * ND;
#pragma unroll
          for (int j = 0; j < ND; j++)
            ld[j] = dg[j];
          if (carry != NULL)
            { u32 *lc = (u32 *) tcar + tid * ND;
#pragma unroll
              for (int j = 0; j < ND; j++)
                lc[j] = dc[j];
            }
        }
      else
        { const u32 *gsrc = src + tstart * RW;
          for (int i = tid; i < ndw; i += RXW_THREADS)
            recs[i] = gsrc[i];
          for (int i = tid; i < tn; i += RXW_THREADS)
            tdig[i] = curdig[tstart + i];
          if (carry != NULL)
            for (int i = tid; i < tn; i += RXW_THREADS)
              tcar[i] = carry[tstart + i];
        }
      const u64 gpre = gp;
      for (int i = tid; i < RXW_WAVES * 256; i += RXW_THREADS)
        whist[i] = 0;
      __syncthreads();

      // the next tile travels while this one is ranked and written
      { const int64_t kn = k + step;
        if (kn < per && tbase + kn < ntiles)
          RXW_FETCH(tbase + kn)
      }

      const int wbase = wave * 64 * ITEMS;
      u32 info[ITEMS];
      u32 old[ITEMS];
#ifdef FK_ABLATION
      if (unstable & RX_ABL_NORANK)
        {
#pragma unroll
          for (int it = 0; it < ITEMS; it++)
            info[it] = (u32) tdig[wbase + it * 64 + lane];
        }
      else
#endif
      if (HASHED && (unstable & 1))
        {
#pragma unroll
          for (int it = 0; it < ITEMS; it++)
            { const int  r     = wbase + it * 64 + lane;
              const bool valid = (r < tn);
              const u32  d     = valid ? (u32) tdig[r] : 0u;
              const u32  rk    = valid ? atomicAdd(&whist[wave * 256 + d], 1u) : 0u;
              info[it] = d | (rk << 8);
            }
        }
      else
      {
#pragma unroll
      for (int it = 0; it < ITEMS; it++)
        { const int  r     = wbase + it * 64 + lane;
          const bool valid = (r < tn);
          const u32  d     = valid ? (u32) tdig[r] : 0u;
          u64 mask = __ballot(valid);
#pragma unroll
          for (int b = 0; b < 8; b++)
            { const bool bit = (d >> b) & 1u;
              const u64  bm  = __ballot(bit);
              mask &= bit ? bm : ~bm;
            }
          const u32 below  = (u32) __popcll(mask & lt);
          const u32 leader = valid ? (u32) (__ffsll((unsigned long long) mask) - 1) : (u32) lane;
          info[it] = d | (below << 8) | (leader << 16);
          old[it] = 0;
          if (valid && below == 0)
            old[it] = atomicAdd(&whist[wave * 256 + d], (u32) __popcll(mask));
        }
#pragma unroll
      for (int it = 0; it < ITEMS; it++)
        { const u32 e    = info[it];
          const u32 base = (u32) __shfl((int) old[it], (int) ((e >> 16) & 0xffu), 64);
          info[it] = (e & 0xffu) | ((base + ((e >> 8) & 0xffu)) << 8);     // d | rank-in-wave << 8
        }
      }
      __syncthreads();

      // bins: exclusive scan over waves, then over the 256 bins (threads 0..255 = waves 0..3)
      u32 run = 0, incl = 0;
      if (tid < 256)
        {
#pragma unroll
          for (int w = 0; w < RXW_WAVES; w++)
            { const u32 t = whist[w * 256 + tid];
              whist[w * 256 + tid] = run;
              run += t;
            }
          incl = run;
#pragma unroll
          for (int o = 1; o < 64; o <<= 1)
            { const u32 y = __shfl_up(incl, o, 64);
              if (lane >= o) incl += y;
            }
          if (lane == 63) tmp32[wave] = incl;
        }
      __syncthreads();
      if (tid < 256)
        { u32 base = 0;
#pragma unroll
          for (int w = 0; w < 4; w++)
            if (w < wave) base += tmp32[w];
          const u32 bstart = base + incl - run;
          binstart[tid] = bstart;
          goff[tid] = (int64_t) gpre - (int64_t) bstart;
        }
      __syncthreads();

#pragma unroll
      for (int it = 0; it < ITEMS; it++)
        { const int r = wbase + it * 64 + lane;
          if (r < tn)
            { const u32 e   = info[it];
              const u32 d   = e & 0xffu;
              const u32 pos = binstart[d] + whist[wave * 256 + d] + (e >> 8);
              perm[pos] = (uint16_t) r;
            }
        }
      __syncthreads();

#pragma unroll
      for (int it = 0; it < ITEMS; it++)
        { const int p = it * RXW_THREADS + tid;
          if (p < tn)
            {
#ifdef FK_ABLATION
              const int     sr = (unstable & RX_ABL_NOPERM) ? p : perm[p];
              const u32     d  = tdig[sr];
              const int64_t o  = (unstable & RX_ABL_LINEAR) ? tstart + p : goff[d] + p;
#else
              const int     sr = perm[p];
              const u32     d  = tdig[sr];
              const int64_t o  = goff[d] + p;
#endif
              rx_rec<RW> r = *(const rx_rec<RW> *) (recs + sr * RW);
              *(rx_rec<RW> *) (dst + o * RW) = r;
              if (next_byte >= 0)
                { u32 nd;
                  if (carry != NULL)
                    nd = tcar[sr];
                  else
                    { nd = HASHED ? rx_hash_digit<RW>(r.w, next_byte, hbytes)
                                  : (u32) lbytes[sr * RW * 4 + next_byte];
#ifdef FK_ABLATION
                      if (HASHED && (unstable & RX_ABL_NOHASH)) nd = (r.w[0] >> (8 * (next_byte & 3))) & 0xffu;   // (a byte that differs from pass to pass: the same byte every time would leave the records SORTED after two passes and the later ones would write whole lines -- the first version of this ablation measured that, not the hash)
#endif
                    }
                  nextdig[o] = (uint8_t) nd;
                }
            }
        }
      __syncthreads();
    }
}

#undef RXW_FETCH

template <int RW, int ITEMS> static size_t rx_wide_lds_bytes()
{ return ((size_t) RXW_THREADS * ITEMS * RW * 4 + 256 * 8 + RXW_WAVES * 256 * 4 + 256 * 4 + 8 * 4
          + (size_t) RXW_THREADS * ITEMS * (RW == 5 ? 4 : 3) + 16);   // (perm 2 bytes; this pass's digit 1; the carried one 1 --
}                                                                        //  only 20-byte records ever bring one: 32-byte tiles fill the LDS)

#ifdef FK_ABLATION
template <int RW, int ITEMS> static size_t rx_lds_bytes(bool hashed)
{ return ((size_t) RX_THREADS * ITEMS * RW * 4 + 256 * 8 + 8 * 8 + RX_WAVES * 256 * 4 + 256 * 4
          + 8 * 4 + 16 + (size_t) RX_THREADS * ITEMS * (hashed ? 4 : 2) + 16);
}

#endif

template <int RW, int ITEMS> static size_t rx_stream_lds_bytes()
{ return ((size_t) RX_THREADS * ITEMS * RW * 4 + 256 * 8 + 8 * 8 + RX_WAVES * 256 * 4 + 256 * 4
          + 8 * 4 + 16 + (size_t) RX_THREADS * ITEMS * 3 + 16);
}

// top > 0 (the MSD engine, fk_tsort.hip): of the non-constant digits only the `top` most significant -- the last
// `top` of bytes[] -- are passes; ctx->rx_top_pbytes = bytes the records are then in order on.
template <int RW, int ITEMS, bool HASHED, int WT>
static int lsd_sort_stream_t(fk_ctx *ctx, int64_t n, void *d_src, void *d_trg, const int *bytes,
                             int nbytes, void **result, int hbytes, int top = 0)
{ constexpr int TILE = WT * ITEMS;
  const int64_t ntiles  = (n + TILE - 1) / TILE;
  const int64_t nchunks = (ntiles + RX_CH - 1) / RX_CH;
  const int64_t nsuper  = (nchunks + RX_SC - 1) / RX_SC;
  hipStream_t   s = ctx->stream;
  u32 want = 0;

  ctx->sort_stats.passes = 0;
  ctx->sort_stats.nelem  = n;
  ctx->sort_stats.rsize  = RW * 4;
  ctx->sort_stats.pass_ms_total = 0.;
  ctx->sort_stats.scatter_ms_total = 0.;
  ctx->sort_stats.hist_ms = 0.;
  *result = d_src;
  if (n == 0 || nbytes == 0)
    return (FK_OK);
  if (nbytes > 60)
    { fk_set_error(ctx, "too many key bytes (%d)", nbytes);
      return (FK_EINVAL);
    }
  for (int i = 0; i < nbytes; i++)
    { if (bytes[i] < 0 || bytes[i] >= (HASHED ? 8 : RW * 4))
        { fk_set_error(ctx, "key byte %d outside record of %d bytes", bytes[i], RW * 4);
          return (FK_EINVAL);
        }
      want |= (1u << bytes[i]);
    }

  uint8_t  *dig_a = (uint8_t *) fk_slot(ctx, FK_SLOT_DIG_A, n + 64);
  uint8_t  *dig_b = (uint8_t *) fk_slot(ctx, FK_SLOT_DIG_B, n + 64);
  uint16_t *tilepfx  = (uint16_t *) fk_slot(ctx, FK_SLOT_RX_TILE, ntiles * 256 * 2);
  u32      *chunktot = (u32 *) fk_slot(ctx, FK_SLOT_RX_CHUNK, nchunks * 256 * 4 * 2);
  u64      *supertot = (u64 *) fk_slot(ctx, FK_SLOT_RX_SUPER, nsuper * 256 * 8 * 2);
  if (dig_a == NULL || dig_b == NULL || tilepfx == NULL || chunktot == NULL || supertot == NULL)
    return (FK_ENOMEM);
  u32 *chunkpfx = chunktot + nchunks * 256;
  u64 *superpfx = supertot + nsuper * 256;

  // the producer of the records (k_ex_expand) may have left the histograms of hash digits 0 and 1
  // and the stream of digit 0 behind
  // ... or the splitter the stream of hash digit 0 of whole super-mer records, beside the records (no histograms:
  // k_rx_superscan sums the digit totals from the tile histograms)
  // (the slots above are reserved first: if that took the digit stream's memory, dig_lost says so and the records'
  //  own pass makes the stream as before)
  const uint8_t *pre_dig = (HASHED && !ctx->dig_lost && ctx->pre_dig != NULL && ctx->pre_dig_n == n && bytes[0] == 0 && hbytes == RW * 4)
                           ? ctx->pre_dig : NULL;
  ctx->pre_dig = NULL;
  const bool pre = (pre_dig != NULL)
                   || (HASHED && ctx->pre_hist_n == n && nbytes <= 2 && bytes[0] == 0
                       && (nbytes < 2 || bytes[1] == 1) && hbytes == ctx->wid.kmer_bytes);
  ctx->pre_hist_n = 0;
  if (!pre)
    FK_HIP(ctx, hipMemsetAsync(ctx->d_digit_hist, 0, 32 * 256 * sizeof(u64), s));
  FK_HIP(ctx, hipEventRecord(ctx->ev0, s));
  if (!pre)
  { int64_t nb = (n + RX_THREADS - 1) / RX_THREADS;
    if (nb > 2048) nb = 2048;
    // the first executed pass is not known before the histograms are: emit the stream of bytes[0]
    // and re-emit below in the (rare) case that this digit turns out to be constant
    if (HASHED)
      hipLaunchKernelGGL(k_hash_hist<RW>, dim3((unsigned) nb), dim3(RX_THREADS), 0, s,
                         (const u32 *) d_src, n, nbytes, ctx->d_digit_hist, dig_a, bytes[0], hbytes);
    else
      hipLaunchKernelGGL(k_digit_hist<RW>, dim3((unsigned) nb), dim3(RX_THREADS), 0, s,
                         (const u32 *) d_src, n, want, ctx->d_digit_hist, dig_a, bytes[0]);
    FK_LAUNCH_CHECK(ctx);
  }
  FK_HIP(ctx, hipEventRecord(ctx->ev1, s));
  int run[64], nrun = 0;                 // the passes that actually permute something
  if (HASHED)
    { // digits of a hash are never constant on real input, and a pass over a constant digit is merely the
      // identity: no reason to bring the histograms to the host and wait for them before the first pass
      for (int i = 0; i < nbytes; i++)
        run[nrun++] = bytes[i];
    }
  else
    { FK_HIP(ctx, hipMemcpyAsync(ctx->h_scratch, ctx->d_digit_hist, (size_t) RW * 4 * 256 * 8, hipMemcpyDeviceToHost, s));
      FK_HIP(ctx, hipStreamSynchronize(s));
      float ms = 0.f;
      FK_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
      ctx->sort_stats.hist_ms = ms;
      for (int i = 0; i < nbytes; i++)
        { const u64 *h = ctx->h_scratch + (size_t) bytes[i] * 256;
          bool constant = false;
          for (int x = 0; x < 256; x++)
            if (h[x] == (u64) n)
              constant = true;
          if (!constant)
            run[nrun++] = bytes[i];
        }
    }
  if (top > 0)
    { // bytes[] holds the leading key bytes, least significant first: of the non-constant ones the `top` most
      // significant are levels (every byte in front of the last level that is not a level is constant); with no
      // more non-constant bytes than levels the records come out in order on all the bytes given
      ctx->rx_top_pbytes = bytes[0] + 1;
      if (nrun > top)
        { for (int i = 0; i < top; i++)
            run[i] = run[nrun - top + i];
          nrun = top;
          ctx->rx_top_pbytes = run[0] + 1;
        }
    }
  if (nrun == 0)
    return (FK_OK);
  if (run[0] != bytes[0])
    { int64_t nb = (n + RX_THREADS - 1) / RX_THREADS;
      if (nb > 2048) nb = 2048;
      FK_HIP(ctx, hipMemsetAsync(ctx->d_digit_hist + 24 * 256, 0, 8 * 256 * sizeof(u64), s));
      if (HASHED)
        hipLaunchKernelGGL(k_hash_hist<RW>, dim3((unsigned) nb), dim3(RX_THREADS), 0, s,
                           (const u32 *) d_src, n, 0, ctx->d_digit_hist + 24 * 256, dig_a, run[0], hbytes);
      else
        hipLaunchKernelGGL(k_digit_hist<RW>, dim3((unsigned) nb), dim3(RX_THREADS), 0, s,
                           (const u32 *) d_src, n, 0u, ctx->d_digit_hist + 24 * 256, dig_a, run[0]);
      FK_LAUNCH_CHECK(ctx);
    }

  // the splitter's second digit plane (hash digit 1 of every record, fk_split.hip): the first pass carries it along
  // (fk_debug_set("radix_engine", 5): it hashes the records again, as before round 5 -- the tests run both)
  const uint8_t *carry0 = (HASHED && RW == 5 && pre_dig != NULL && ctx->dig2_off > 0 && nrun >= 2 && run[0] == 0 && run[1] == 1
                           && ctx->dbg_radix_engine != 5)
                          ? pre_dig + ctx->dig2_off : NULL;
  size_t lds_bytes;
  if constexpr (WT == RX_THREADS) lds_bytes = rx_stream_lds_bytes<RW, ITEMS>();
  else lds_bytes = rx_wide_lds_bytes<RW, ITEMS>();
  u32 *src = (u32 *) d_src, *trg = (u32 *) d_trg;
  uint8_t *dcur = (pre_dig != NULL) ? (uint8_t *) pre_dig : dig_a, *dnext = dig_b;      // (a producer's stream is only read)
  unsigned sgrid = (unsigned) (((ntiles + 7) / 8) * 8);
  if constexpr (WT != RX_THREADS)
    { // the attribute belongs to the (function, device) pair: remembered per context (= per device), not per
      // process -- a host that drives several devices from one process must set it on each of them
      const u64 bit = 1ull << ((RW * 2 + (HASHED ? 1 : 0)) & 63);
      if ((ctx->rx_attr_done & bit) == 0)
        { auto kern = k_rx_scatter_w<RW, ITEMS, HASHED>;
          FK_HIP(ctx, hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int) lds_bytes));
          ctx->rx_attr_done |= bit;
        }
      const unsigned cus = (unsigned) ((ctx->num_cus > 0 ? ctx->num_cus : 256) / 8 * 8);
      if (sgrid > cus) sgrid = cus;
    }
  FK_HIP(ctx, hipEventRecord(ctx->ev0, s));
  for (int i = 0; i < nrun; i++)
    if (ctx->pass_ev[2 * i] == NULL)
      { if (fkx_event_get(ctx->device, true, &ctx->pass_ev[2 * i]) != FK_OK
            || fkx_event_get(ctx->device, true, &ctx->pass_ev[2 * i + 1]) != FK_OK)
          { fk_set_error(ctx, "sort: cannot create events");
            return (FK_EHIP);
          }
      }
  for (int i = 0; i < nrun; i++)
    { const int nextb = (i + 1 < nrun) ? run[i + 1] : -1;
      hipLaunchKernelGGL(k_rx_tilehist<TILE / RX_THREADS>, dim3((unsigned) nchunks), dim3(RX_THREADS), 0, s,
                         (const uint8_t *) dcur, n, tilepfx, chunktot);
      hipLaunchKernelGGL(k_rx_chunkscan, dim3((unsigned) nsuper), dim3(RX_THREADS), 0, s,
                         (const u32 *) chunktot, nchunks, chunkpfx, supertot);
      hipLaunchKernelGGL(k_rx_superscan, dim3(1), dim3(RX_THREADS), 0, s,
                         (const u64 *) (ctx->d_digit_hist + (size_t) run[i] * 256),
                         (const u64 *) supertot, nsuper, superpfx);
      FK_HIP(ctx, hipEventRecord(ctx->pass_ev[2 * i], s));
      if constexpr (WT == RX_THREADS)
        hipLaunchKernelGGL((k_rx_scatter<RW, ITEMS, HASHED>), dim3(sgrid),
                           dim3(RX_THREADS), lds_bytes, s,
                           (const u32 *) src, trg, n, run[i], nextb, (const uint16_t *) tilepfx,
                           (const u32 *) chunkpfx, (const u64 *) superpfx, (const uint8_t *) dcur, dnext, ntiles,
                           hbytes, ((HASHED && i == 0 && ctx->dbg_radix_engine != 4) ? 1 : 0) | RX_ABL_BITS(ctx));
      else
        hipLaunchKernelGGL((k_rx_scatter_w<RW, ITEMS, HASHED>), dim3(sgrid),
                           dim3(RXW_THREADS), lds_bytes, s,
                           (const u32 *) src, trg, n, nextb, (const uint16_t *) tilepfx,
                           (const u32 *) chunkpfx, (const u64 *) superpfx, (const uint8_t *) dcur, dnext, ntiles,
                           hbytes, ((HASHED && i == 0 && ctx->dbg_radix_engine != 4) ? 1 : 0) | RX_ABL_BITS(ctx),
                           (i == 0) ? carry0 : (const uint8_t *) NULL);
      FK_HIP(ctx, hipEventRecord(ctx->pass_ev[2 * i + 1], s));
      FK_LAUNCH_CHECK(ctx);
      u32 *t = src; src = trg; trg = t;
      uint8_t *d = (dcur == pre_dig) ? dig_a : dcur; dcur = dnext; dnext = d;
    }
  FK_HIP(ctx, hipEventRecord(ctx->ev1, s));
  FK_HIP(ctx, hipStreamSynchronize(s));
  { float ms = 0.f;
    FK_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    ctx->sort_stats.pass_ms_total = ms;
    ctx->sort_stats.passes = nrun;
    double sc = 0.;
    for (int i = 0; i < nrun; i++)
      { FK_HIP(ctx, hipEventElapsedTime(&ms, ctx->pass_ev[2 * i], ctx->pass_ev[2 * i + 1]));
        sc += ms;
      }
    ctx->sort_stats.scatter_ms_total = sc;
    if (getenv("FK_SORT_TIMING") != NULL)
      { fprintf(stderr, "  sort timing: n %lld, R %d, %s, %d passes, digit stream %s, second digit %s:", (long long) n, RW * 4,
                HASHED ? "hashed" : "keys", nrun, pre_dig != NULL ? "from the producer" : "made here",
                carry0 != NULL ? "carried" : "hashed / read in the pass");
        for (int i = 0; i < nrun; i++)
          { FK_HIP(ctx, hipEventElapsedTime(&ms, ctx->pass_ev[2 * i], ctx->pass_ev[2 * i + 1]));
            fprintf(stderr, " %.3f ms", ms);
          }
        fprintf(stderr, "\n");
      }
  }
  *result = (void *) src;
  return (FK_OK);
}

template <bool HASHED>
static int sort_dispatch(fk_ctx *ctx, int64_t nelem, void *d_src, void *d_trg, int rsize,
                         const int *bytes, int nbytes, void **result, int hbytes, int top = 0)
{ if (rsize <= 0 || (rsize & 3) != 0 || rsize > 32)
    { fk_set_error(ctx, "record size %d not supported (multiple of 4, <= 32)", rsize);
      return (FK_EUNSUPPORTED);
    }
  // wide (1024-thread, persistent) tiles pay off for records of 16 bytes and more; narrower records
  // are bound by instruction issue, not by the write pattern, and do better with 4 workgroups per CU
  const bool narrow = (ctx->dbg_radix_engine == 2) || (rsize < 16 && ctx->dbg_radix_engine != 3);
#define RX_CASE(RW) return (narrow ? lsd_sort_stream_t<RW, RxCfg<RW>::ITEMS, HASHED, RX_THREADS>(ctx, nelem, d_src, d_trg, bytes, nbytes, result, hbytes, top) \
                                   : lsd_sort_stream_t<RW, RxCfgW<RW>::ITEMS, HASHED, RXW_THREADS>(ctx, nelem, d_src, d_trg, bytes, nbytes, result, hbytes, top))
  switch (rsize >> 2)
  { case 1: RX_CASE(1);
    case 2: RX_CASE(2);
    case 3: RX_CASE(3);
    case 4: RX_CASE(4);
    case 5: RX_CASE(5);
    case 6: RX_CASE(6);
    case 7: RX_CASE(7);
    default: RX_CASE(8);
  }
#undef RX_CASE
}

int fkx_lsd_sort(fk_ctx *ctx, int64_t nelem, void *d_src, void *d_trg, int rsize,
                 const int *bytes, int nbytes, void **result)
{ return sort_dispatch<false>(ctx, nelem, d_src, d_trg, rsize, bytes, nbytes, result, rsize); }

// the levels of the MSD engine (fk_tsort.hip): bytes[] = the leading key bytes, least significant first; of the
// non-constant ones the `top` most significant become passes
int fkx_lsd_sort_top(fk_ctx *ctx, int64_t nelem, void *d_src, void *d_trg, int rsize, const int *bytes, int nbytes,
                     int top, void **result)
{ ctx->rx_top_pbytes = (nbytes > 0) ? bytes[0] + 1 : 0;
  return sort_dispatch<false>(ctx, nelem, d_src, d_trg, rsize, bytes, nbytes, result, rsize, top);
}

// Make records with identical first key_bytes adjacent: npasses stable digit passes over a hash of them.
// This is all the super-mer "sort" has to achieve (count.c:421-426 only run-length encodes
// duplicates); records that collide in 40 bits merely stay un-merged, which the weighted k-mer
// stage absorbs because it sums weights per k-mer anyway.
int fkx_group(fk_ctx *ctx, int64_t nelem, void *d_src, void *d_trg, int rsize, int key_bytes,
              int npasses, void **result)
{ static const int bytes[8] = { 0, 1, 2, 3, 4, 5, 6, 7 };
  if (npasses < 1 || npasses > 8 || key_bytes < 1 || key_bytes > rsize)
    return (FK_EINVAL);
  return sort_dispatch<true>(ctx, nelem, d_src, d_trg, rsize, bytes, npasses, result, key_bytes);
}

// census[x] = records whose byte 0 is x (the role of Kparts, count.c:1527-1535, for exact_parts)
template <int RW>
static int census_t(fk_ctx *ctx, const void *d_recs, int64_t n, int64_t *census)
{ hipStream_t s = ctx->stream;
  FK_HIP(ctx, hipMemsetAsync(ctx->d_digit_hist, 0, 256 * sizeof(u64), s));
  int64_t nb = (n + RX_THREADS - 1) / RX_THREADS;
  if (nb > 2048) nb = 2048;
  if (nb > 0)
    hipLaunchKernelGGL(k_digit_hist<RW>, dim3((unsigned) nb), dim3(RX_THREADS), 0, s, (const u32 *) d_recs, n,
                       1u, ctx->d_digit_hist, (uint8_t *) NULL, 0);
  FK_LAUNCH_CHECK(ctx);
  FK_HIP(ctx, hipMemcpyAsync(ctx->h_scratch, ctx->d_digit_hist, 256 * sizeof(u64), hipMemcpyDeviceToHost, s));
  FK_HIP(ctx, hipStreamSynchronize(s));
  for (int x = 0; x < 256; x++)
    census[x] = (int64_t) ctx->h_scratch[x];
  return (FK_OK);
}

int fkx_first_byte_census(fk_ctx *ctx, const void *d_recs, int64_t n, int rsize, int64_t *census)
{ switch (rsize >> 2)
  { case 1: return census_t<1>(ctx, d_recs, n, census);
    case 2: return census_t<2>(ctx, d_recs, n, census);
    case 3: return census_t<3>(ctx, d_recs, n, census);
    case 4: return census_t<4>(ctx, d_recs, n, census);
    case 5: return census_t<5>(ctx, d_recs, n, census);
    default: return (FK_EUNSUPPORTED);
  }
}
