// fk_split.hip -- reads -> 2-bit packed, strand-canonical super-mer records, bucketed by minimizer.
//
// Replaces Distribute_Block + Stuff_Seq (split.c:1016-1393, 864-989) and the unpacking half,
// supermer_list_thread (count.c:165-313): the bit-stuffed ".T" spill format only exists to save
// disk, so the kernel writes the fixed-width records the sort consumes directly.
//
// The reference walks every read sequentially with a history-dependent tie rule (strict < on
// arrival, <= on a forced rescan, split.c:1110,1149,1310).  Which k-mers share a super-mer never
// changes .hist or the .ktab canonical stream, so the device uses a rule that is a
// pure function of the window and therefore position-parallel:
//   * the read buffer is treated as one flat byte string; any byte that is not acgtACGT (read
//     terminators, N, newlines) invalidates the k-mers that cover it -- exactly the k-mers the
//     reference skips (split.c:1079, 1124-1128, 1323-1330);
//   * minimizer of a k-mer = smallest canonical 7-mer of its K-6 7-mer starts under a fixed
//     pseudo-random order (a multiplicative hash of the 14-bit code, fk_mrank14), leftmost on ties (min
//     over packed (rank,position) keys).  Seven bases, not the reference's initial five: the
//     smallest 5-mer wins 6.8 % of all windows at k = 40, so no deal of 5-mer ranks can balance
//     more than ~14 buckets -- the reference pads its heavy minimizers by two bases for the same
//     reason (refine_tree, split.c:437-472: PAD_LEN 5 -> 7); with 7-mers the heaviest rank holds
//     0.4 %;
//   * a super-mer = maximal run of consecutive valid k-mers with the same minimizer POSITION, cut
//     at tile edges; it holds at most K-6 k-mers (MAX_SUPER after one PAD refinement), which the
//     record widths of K-4 hold;
//   * the record is reverse-complemented when the minimizer lies on the - strand (split.c:1281),
//     so the two strands of a locus give byte-identical records;
//   * bucket = f(minimizer value): equal canonical k-mers always share a bucket.
#include "fk_common.h"

#define SP_THREADS 256
#define SP_CH      16                      // k-mer starts per thread
#define SP_TILE    (SP_THREADS * SP_CH)    // 4096 k-mer starts per workgroup
#define SP_MAXK    64                       // fk_create admits k <= 64; 27.1 KB of LDS lets six workgroups share a CU
#define SP_WORDS   (SP_TILE / 16 + SP_MAXK / 16 + 2)   // packed words incl. halo and guard
#define SP_KEYS    (SP_TILE + SP_MAXK)
#define SP_KIDX(i) ((i) + ((i) >> 4))
#define SP_RTW     8                       // tiles a replay workgroup takes
#define SP_LB      10                      // log2 of the records per chunk of a stream (see SplitArgs.lstreams)
#define SP_LSTREAMS 3                      // 8 streams per bucket
#define SP_PL      512                     // super-mer starts the position list of a tile takes at a time
#define SP_RCH     16                      // tiles per chunk of the replay offsets (in-chunk prefixes fit 16 bits)     // one pad word per 16 keys: thread t's chunk starts at bank 17t

struct SplitArgs
{ const unsigned char *bases;
  int64_t   nbytes;
  int       kmer;
  int       smer_bytes;
  int       sww;            // record stride in dwords
  int       nbuckets;
  const uint8_t  *mbucket;  // [FK_NRANKS] bucket of a canonical minimizer rank (global memory: one read per super-mer)
  u64      *counts;         // [nbuckets] records per bucket (count mode)  + [256] = instances
  u64      *cursor;         // [(nbuckets << lstreams) * cstride] running write cursors (emit mode), from 0
  int       cstride;        // u64 words between two buckets' cursors: every tile adds to every bucket's cursor, and
                            // cursors that share a memory channel serialise (FK_CURSOR_STRIDE words = 4 KB apart)
  const u64 *limit;         // [nbuckets] end of each bucket's region (NULL: only `cap` bounds the output)
  u32      *out;
  int64_t   cap;
  u32      *overflowed;
  int       tile_stride;    // count mode: visit every tile_stride-th tile only (sampling)
  u64      *pos;            // POS kernels only: (position of the record's first k-mer << 1) | flip per record
  int64_t   tile0;          // first tile of this launch (a grid holds at most SP_MAXGRID workgroups)
  int64_t   tile_end;       // replay: tiles of the input (a workgroup takes SP_RTW of them)
  // multi-pass split with entry replay: the first pass emits the records of buckets [gb0, gb1) and leaves, per
  // tile, the 4-byte entries (start, flip, length, bucket) of all OTHER super-mers behind; the later passes
  // rebuild their records from the entries and the reads (k_split_replay) without recomputing any minimizer
  u32      *ent;            // entries, 64 sub-regions of ent_cap / 64 each (NULL: not recording)
  u64      *ent_cursor;     // [64 * cstride] write cursors of the sub-regions
  u64      *tile_ent;       // [tiles] (absolute first entry << 13) | entries of the tile
  int64_t   ent_cap;
  int       gb0, gb1;
  // the recording pass also leaves, per tile and bucket, the number of super-mers (u16 [tiles][nbuckets]); two
  // small scans turn them into in-chunk prefixes (in place) and chunk bases, so a replay pass knows where each of
  // its records goes without a single global atomic (24 cursor round trips per tile were 3/4 of a replay pass)
  uint16_t *tile_cnt;       // [tiles * nbuckets]: counts, after the scan exclusive prefixes inside a chunk of SP_RCH tiles
  const u64 *chunk_base;    // [chunks * nbuckets] records of the bucket in all earlier chunks
  // Every tile reserves room with one returning atomicAdd per bucket, and the memory side serves ~80 M of them per
  // second and ADDRESS (measured: 1.23 M tiles on one cursor = 15.0 ms however little else the kernel does, 8.4 ms
  // on four).  So a bucket has 2^lstreams cursors, tile t uses cursor t mod 2^lstreams, and stream c owns
  // the chunks c, c + C, c + 2C, ... (SP_LB records each) of the bucket's region: the region fills evenly from the
  // front, k_split_compact closes the ragged end (a few chunks per bucket) afterwards.
  u64      *plan;           // count mode: [32][256] spread bucket counters (NULL: counts[])
  int       lstreams;
  const u64 *rbase;         // [nbuckets] first record of each bucket's region (the cursors count from 0)
  int       abl;            // -DFK_ABLATION builds: FK_REPLAY_ABL bits -- 1 no record stores, 2 no base loads, 4 no entries
  u32       skipb;          // super-mers whose bucket entry equals skipb are dropped (0xFF in a group
                            // pass of a multi-pass split, where mbucket marks the other groups' ranks;
                            // 0x100 = nothing is dropped)
  // PACKED kernels: `bases` holds two bits per base, 16 bases per dword (fk_pkview in fk_common.h), nbytes = positions
  const int64_t *roff;      // [nreads + 1] first position of every read
  int64_t   nreads;
  const int64_t *inv;       // [2 ninv] (first position, length) of the stretches that hold no acgt
  int64_t   ninv;
  const u32 *tidx;          // [tiles + 2][2] first read that ends at or behind the tile's first position, first stretch that
                            // does (0xffffffff: none)
  uint8_t  *dig;            // != NULL (20-byte records, one-pass emit): byte 0 of the hash of every record, at the record's
                            // slot -- the first digit stream of the grouping sort that follows (fk_radix.hip), made
                            // while the record is in registers instead of by a pass of its own over all records
  uint8_t  *dig2;           // ... and byte 1 of the same hash, a second plane indexed alike (dig + fk_ctx.dig2_off): the first
                            // pass of the grouping sort carries it to the records' new places instead of hashing all
                            // twenty bytes again (round 5: that hash was 28 % of the pass, profiles/r05_scatter_ablation.json)
};

__device__ __forceinline__ u32 sp_window(const u32 *arr, int off)
{ // 16 bases (32 bits) starting at base offset `off` of an MSB-first packed array
  const u32 hi = arr[off >> 4];
  const u32 lo = arr[(off >> 4) + 1];
  // one 64-bit shift; a conditional on the shift being 0 makes the compiler sink the second load
  // into a divergent branch
  return ((u32) (((((u64) hi) << 32) | (u64) lo) >> (32 - 2 * (off & 15))));
}

// 16 ASCII bases (four little-endian dwords, first base in the lowest byte) -> one MSB-first packed word and
// a 16-bit mask of the bytes that are not acgtACGT, four bases per instruction: code = ((c >> 1) & 3) ^ its
// own high bit, validity by looking the code's letter up again (v_perm_b32 on "ACGT") and comparing it with
// the upper-cased byte; the four 2-bit fields of a dword collapse with two shift-ors.  107 instead of 122
// VALU lane-instructions per base for the kernel (SQ_INSTS_VALU) -- at an unchanged 15.5 ms per 5 G bases:
// the kernel is not bound by VALU issue.
__device__ __forceinline__ void sp_pack16(const uint4 v, u32 &word, u32 &bad)
{ const u32 w4[4] = { v.x, v.y, v.z, v.w };
  u32 lsb = 0, bd = 0;
#pragma unroll
  for (int d = 0; d < 4; d++)
    { const u32 w    = w4[d];
      const u32 x    = (w >> 1) & 0x03030303u;
      const u32 code = x ^ ((x >> 1) & 0x01010101u);
      const u32 want = __builtin_amdgcn_perm(0u, 0x54474341u, code);      // "ACGT"[code] per byte
      const u32 diff = (w & 0xDFDFDFDFu) ^ want;
      const u32 nz   = (((diff & 0x7f7f7f7fu) + 0x7f7f7f7fu) | diff) & 0x80808080u;   // bit 7 of every non-zero byte
      const u32 nib  = code | (code >> 6);                                 // c0 | c1 << 2  (and c2 | c3 << 2 at bit 16)
      const u32 byte = (nib | (nib >> 12)) & 0xffu;                        // c0 | c1 << 2 | c2 << 4 | c3 << 6
      const u32 f    = nz >> 7;                                            // flags at bits 0, 8, 16, 24
      const u32 fl   = (f | (f >> 7) | (f >> 14) | (f >> 21)) & 0xfu;
      lsb |= byte << (8 * d);
      bd  |= fl << (4 * d);
    }
  // base j sits at bits [2j, 2j+2) of lsb; the arrays are MSB first: reverse the pairs
  const u32 y = __builtin_bitreverse32(lsb);
  word = ((y >> 1) & 0x55555555u) | ((y & 0x55555555u) << 1);
  bad  = bd;
}

// A 20-byte super-mer record (k = 37 .. 40) in one piece: six packed words in (independent LDS reads), five words
// out, stored as dwordx4 + dword.  Five separate dword stores per lane, each lane at an unrelated address, are five
// write requests per record: the replay passes spent a quarter of their time on them.  Other record widths take the
// generic loop.
struct __attribute__((packed, aligned(4))) sp_rec5 { u32 w[5]; };
__device__ __forceinline__ void sp_put_record5(const u32 *arr, int st, int L, u32 lenbits, int lenw, u32 *dst)
{ const int wi = st >> 4;
  const u32 sh = 32u - 2u * (u32) (st & 15);                  // 2 .. 32
  u32 w[6];
#pragma unroll
  for (int q = 0; q < 6; q++)
    w[q] = arr[wi + q];                                       // wi + 5 < SP_WORDS for every start inside a tile
  sp_rec5 r;
#pragma unroll
  for (int q = 0; q < 5; q++)
    { u32 x = (u32) (((((u64) w[q]) << 32) | (u64) w[q + 1]) >> sh);
      int r2 = 2 * (L - 16 * q);                              // bits of this word that hold bases
      r2 = r2 < 0 ? 0 : (r2 > 32 ? 32 : r2);
      x &= ~(u32) ((0xffffffffffffffffull >> r2) >> 32);
      if (q == lenw)
        x |= lenbits;
      r.w[q] = __builtin_bswap32(x);
    }
  *(sp_rec5 *) dst = r;
}

// rank << 15 of a 7-mer: canonical = the smaller of the forward code and its reverse complement, rank =
// fk_mrank14(canonical).  Both arguments may carry garbage above bit 13.  (Which strand the canonical form
// is on only matters for the one minimizer a super-mer ends up with: step 7 looks it up again.)
__device__ __forceinline__ u32 sp_key7(u32 fw, u32 rc)
{ fw &= 0x3fffu;
  rc &= 0x3fffu;
  return (fk_mrank14(min(fw, rc)) << 15);
}

// valid k-mer starts among the 16 of a thread: no marked position in [i, i + Kw) -- m16[q] bit c marks position 16 q + c
__device__ __forceinline__ u32 sp_clear_starts(const uint16_t *m16, int tid, int Kw)
{ if (Kw >= 16)
    { // the window of start c (in word t) is the rest of word t from bit c, the whole words t+1 .. t+kq-1 and the
      // first c+kr bits of the word pair (t+kq, t+kq+1), Kw = 16 kq + kr -- so the valid starts of a thread are one
      // interval [fls(m0), ctz(E) - kr], a dozen instructions per THREAD (the prefix counts this replaces cost 14 per base)
      const int kq = Kw >> 4, kr = Kw & 15;
      const u32 m0 = m16[tid];
      u32 mid = 0;
      for (int j = 1; j < kq; j++)
        mid |= m16[tid + j];
      const u32 E     = (u32) m16[tid + kq] | ((u32) m16[tid + kq + 1] << 16);
      const int lowc  = 32 - __clz((int) m0);                       // first start past the last marked position of word t
      const int highc = (E != 0 ? __ffs((int) E) - 1 : 32) - kr;    // last start whose window ends before the next one
      const u32 up    = (highc >= 15) ? 0xffffu : (highc < 0 ? 0u : ((2u << highc) - 1u));
      return ((mid != 0) ? 0u : (up & ~((1u << lowc) - 1u) & 0xffffu));
    }
  const u32 M = (u32) m16[tid] | ((u32) m16[tid + 1] << 16);
  const u32 km = (1u << Kw) - 1u;
  u32 v = 0;
#pragma unroll
  for (int c = 0; c < SP_CH; c++)
    v |= (((M >> c) & km) == 0u ? 1u : 0u) << c;
  return (v);
}

// PACKED: tile loader for reads in two bits per base -- the 16 bases of a dword are one byte swap away from the packed
// word, no base conversion (a sixth of the ASCII kernel's instructions); which positions are unusable comes from two
// short sorted lists instead of the bytes themselves: the reads' ends (a k-mer must not run across one: the
// concatenation has no terminators) and the stretches without acgt (their code bits are arbitrary).
template <bool PACKED>
__device__ __forceinline__ void sp_load_tile(const SplitArgs &a, int64_t t0, int nw, u32 *fwd, uint16_t *inv16, uint16_t *bnd16)
{ const int tid = threadIdx.x;
  if (PACKED)
    { const u32 *codes = (const u32 *) a.bases;
      for (int q = tid; q < SP_WORDS; q += SP_THREADS)
        { u32 word = 0, bad = 0xffffu;
          if (q < nw)
            { const int64_t g = t0 + (int64_t) q * 16;
              if (g < a.nbytes)
                { word = __builtin_bswap32(codes[g >> 4]);
                  const int64_t left = a.nbytes - g;
                  bad = (left >= 16) ? 0u : ((0xffffu << (int) left) & 0xffffu);
                }
            }
          fwd[q]   = word;
          inv16[q] = (uint16_t) bad;
          bnd16[q] = 0;
        }
      return;
    }
  for (int q = tid; q < SP_WORDS; q += SP_THREADS)
    { u32 word = 0, bad = 0xffffu;
      if (q < nw)
        { const int64_t g = t0 + (int64_t) q * 16;
          uint4 v;
          if (g + 16 <= a.nbytes)
            v = *(const uint4 *) (a.bases + g);
          else
            { u32 d[4] = { 0u, 0u, 0u, 0u };                   // the last bytes of the input, one by one
              for (int j = 0; j < 16; j++)
                if (g + j < a.nbytes)
                  d[j >> 2] |= (u32) a.bases[g + j] << (8 * (j & 3));
              v = make_uint4(d[0], d[1], d[2], d[3]);
            }
          sp_pack16(v, word, bad);
        }
      fwd[q]   = word;
      inv16[q] = (uint16_t) bad;
    }
}

// PACKED: the read ends and invalid stretches that fall into positions [t0, t0 + R) become bits of the two masks
// (LDS atomics on the u16 arrays viewed as one bitmap; little-endian).  In two halves so that the trip to the lists
// overlaps the trip to the codes: sp_fetch_marks, before the tile's codes are waited for, loads the list elements
// a thread may have to look at -- the tile index brackets them: everything that can fall into this tile and its halo
// lies in front of the first element of tile + 2, so a HiFi tile issues a load or two, not 256 -- and sp_mark_tile,
// behind the barrier that published the zeroed masks, sets the bits.
struct sp_marks { int64_t e, s, n; u32 r_lo, r_hi, s_lo, s_hi; };

__device__ __forceinline__ void sp_fetch_marks(const SplitArgs &a, int64_t tile, sp_marks &m)
{ const int tid = threadIdx.x;
  m.r_lo = a.tidx[2 * tile];     m.s_lo = a.tidx[2 * tile + 1];
  m.r_hi = a.tidx[2 * tile + 4]; m.s_hi = a.tidx[2 * tile + 5];
  if ((int64_t) m.r_hi > a.nreads) m.r_hi = (u32) a.nreads;
  // (the first stretch that ENDS behind the next tile may begin in this one: it is looked at as well)
  m.s_hi = ((int64_t) m.s_hi + 1 > a.ninv) ? (u32) a.ninv : m.s_hi + 1;
  m.e = -1; m.s = 0; m.n = 0;
  if ((int64_t) m.r_lo + tid < (int64_t) m.r_hi)
    m.e = a.roff[(int64_t) m.r_lo + tid + 1] - 1;              // last position of read r_lo + tid
  if ((int64_t) m.s_lo + tid < (int64_t) m.s_hi)
    { m.s = a.inv[2 * ((int64_t) m.s_lo + tid)];
      m.n = a.inv[2 * ((int64_t) m.s_lo + tid) + 1];
    }
}

__device__ __forceinline__ void sp_mark_tile(const SplitArgs &a, int64_t t0, int R, const sp_marks &m, uint16_t *inv16, uint16_t *bnd16)
{ const int     tid  = threadIdx.x;
  const int64_t tend = t0 + R;
  u32 *bb = (u32 *) bnd16, *ib = (u32 *) inv16;
  int64_t e = m.e;
  for (int64_t j = (int64_t) m.r_lo + tid; j < (int64_t) m.r_hi; j += SP_THREADS)
    { if (j != (int64_t) m.r_lo + tid)
        e = a.roff[j + 1] - 1;
      if (e >= t0 && e < tend)
        { const int o = (int) (e - t0);
          atomicOr(&bb[o >> 5], 1u << (o & 31));
        }
    }
  int64_t s = m.s, n = m.n;
  for (int64_t j = (int64_t) m.s_lo + tid; j < (int64_t) m.s_hi; j += SP_THREADS)
    { if (j != (int64_t) m.s_lo + tid)
        { s = a.inv[2 * j]; n = a.inv[2 * j + 1]; }
      // (the bracket may hold a stretch that lies wholly behind the tile -- the first one that ENDS behind the next
      //  tile may begin anywhere: the distances are compared in 64 bits before anything is narrowed)
      if (s >= tend || s + n <= t0)
        continue;
      const int lo = (int) ((s > t0 ? s : t0) - t0), hi = (int) ((s + n < tend ? s + n : tend) - t0);
      for (int w = lo >> 5; hi > lo && w <= ((hi - 1) >> 5); w++)
        { const int b0 = (w << 5) > lo ? 0 : lo - (w << 5);
          const int b1 = ((w + 1) << 5) < hi ? 32 : hi - (w << 5);
          atomicOr(&ib[w], (u32) ((0xffffffffull >> (32 - (b1 - b0))) << b0));
        }
    }
}

template <bool EMIT, bool POS = false, bool PACKED = false>
__global__ __launch_bounds__(SP_THREADS) void k_split(SplitArgs a)
{ __shared__ u32      fwd[SP_WORDS];
  __shared__ u32      rcw[SP_WORDS];
  __shared__ __attribute__((aligned(4))) uint16_t inv16[SP_WORDS];

  __shared__ __attribute__((aligned(16))) u32 keys[SP_KEYS + SP_KEYS / 16 + 1];   // prefix minima (step 2), then the window minima by position
  __shared__ u32      lastkey[SP_THREADS];                // last window minimum of every thread (step 5); then pos16
  __shared__ __attribute__((aligned(4))) uint16_t sbits[SP_THREADS + 16];            // boundary bits: start | invalid
  uint16_t *bnd16 = sbits;      // PACKED: the reads' last positions, steps 1 to 4 -- dead before sbits is written (a
                                // separate 524-byte array cost the seventh workgroup per CU)
  static_assert(SP_THREADS + 16 >= SP_WORDS, "bnd16 lives in sbits");
  __shared__ uint8_t  vlast[SP_THREADS];                 // is the thread's last k-mer start valid? (step 5 of the next thread)
  // per bucket: stream position of the tile's records (u64), records counted (u32), records placed (u32) -- sized
  // by the launch for the context's bucket count: with the 4 KB that 256 buckets take a CU holds six workgroups,
  // with the 768 bytes of 48 buckets seven
  FK_DYN_LDS_ALIGNED(unsigned char, sp_dyn, 8);
  __shared__ u32      tmp32[8];
  __shared__ u32      nother, nother2;
  __shared__ u64      ebase;
  const bool rec = (a.ent != NULL);

  u64 *bbase   = (u64 *) sp_dyn;                    // [nbuckets]
  u32 *bcnt    = (u32 *) (bbase + a.nbuckets);      // [nbuckets]
  u32 *bcnt2   = bcnt + a.nbuckets;                 // [nbuckets]
  uint16_t *pos16 = (uint16_t *) lastkey;           // [SP_PL] positions of the tile's super-mer starts: over lastkey,
                                                    // which is dead once the start masks are made (step 5)
  const int     tid = threadIdx.x;
  const int     K   = a.kmer;
  const int     W   = K - 6;                       // 7-mer starts per k-mer = longest super-mer
  const int64_t t0  = (a.tile0 + (int64_t) blockIdx.x) * a.tile_stride * SP_TILE;
  const int     nw  = SP_TILE / 16 + (K + 14) / 16; // words that hold real bases
  const int     R   = nw * 16;                     // bases covered by the packed arrays
  const bool    one = (a.nbuckets == 1);           // single bucket: no per-record LDS atomics
  const u32     strm = (u32) (a.tile0 + blockIdx.x) & ((1u << a.lstreams) - 1u);   // round robin: equal tile counts

  if (tid < a.nbuckets) bcnt[tid] = 0;
  if (tid == 0) { nother = 0; nother2 = 0; }

  // ---- 1. ASCII -> 2-bit codes (MSB first) + invalid masks; PACKED: the codes as they are ----
  sp_marks marks;
  if (PACKED)
    sp_fetch_marks(a, (a.tile0 + (int64_t) blockIdx.x) * a.tile_stride, marks);
  sp_load_tile<PACKED>(a, t0, nw, fwd, inv16, bnd16);
  __syncthreads();
  if (PACKED)
    sp_mark_tile(a, t0, R, marks, inv16, bnd16);           // read in step 4: two barriers on

  // reverse-complement strand, same packing: rc base p' = R-1-p
  for (int q = tid; q < SP_WORDS; q += SP_THREADS)
    { u32 x = 0;
      if (q < nw)
        { const u32 y = __builtin_bitreverse32(fwd[nw - 1 - q]);
          x = ~(((y >> 1) & 0x55555555u) | ((y & 0x55555555u) << 1));
        }
      rcw[q] = x;
    }
  __syncthreads();
  // ---- 2. canonical 7-mer keys: (rank << 14 | position) << 1 ----------------------------------
  //      one thread rolls a 64-bit window over the 16 positions of a packed word.  With W >= 16
  //      (k >= 20) the keys never go to LDS as such: a thread keeps the suffix minima of its own 16
  //      keys in registers and publishes their PREFIX minima (slot 15 = the block minimum), because
  //      a window of W positions starting in block t is  suffix(t) + whole blocks + prefix of the
  //      block it ends in  -- three operands instead of W (step 3)
  const bool fastmin = (W >= SP_CH);
  const int  i0 = tid * SP_CH;
  u32 mk[SP_CH];
  { const int q = tid;
    const u64 x = (((u64) fwd[q]) << 32) | (u64) fwd[q + 1];
    // the reverse complements of the 7-mers at 16q .. 16q+15 lie at rc offsets R-7-(16q+c): 22 bases
    // from offset 10 of the two rc words nw-q-2, nw-q-1; the one of start c is bits [2c, 2c+14) of them
    const u64 y = (((u64) rcw[nw - q - 2]) << 32) | (u64) rcw[nw - q - 1];
    u32 kk[SP_CH];
#pragma unroll
    for (int c = 0; c < 16; c++)
      kk[c] = sp_key7((u32) (x >> (50 - 2 * c)), (u32) (y >> (2 * c))) | ((u32) (16 * q + c) << 1);
    if (fastmin)
      { mk[SP_CH - 1] = kk[SP_CH - 1];
#pragma unroll
        for (int c = SP_CH - 2; c >= 0; c--)
          mk[c] = min(kk[c], mk[c + 1]);                 // suffix minima of the own block
#pragma unroll
        for (int c = 1; c < SP_CH; c++)
          kk[c] = min(kk[c], kk[c - 1]);                 // prefix minima, published
      }
#pragma unroll
    for (int c = 0; c < 16; c++)
      keys[SP_KIDX(16 * q + c)] = kk[c];
  }
  if (tid < SP_MAXK)                                     // the halo past the tile: one position per lane of wave 0
    { // (a thread per 16-position block, as above, made wave 0 run the whole key code a second time for three
      //  or four of its lanes: a twelfth of the kernel's instructions)
      const int p = SP_TILE + tid, q = p >> 4, c = p & 15;
      const u64 x = (((u64) fwd[q]) << 32) | (u64) fwd[q + 1];
      const int qr = nw - q - 2;                            // may run off the front for the last halo words
      const u64 y = (((u64) (qr >= 0 ? rcw[qr] : 0u)) << 32) | (u64) (qr + 1 >= 0 ? rcw[qr + 1] : 0u);
      u32 kk = sp_key7((u32) (x >> (50 - 2 * c)), (u32) (y >> (2 * c))) | ((u32) p << 1);
      if (fastmin)
        {
#pragma unroll
          for (int o = 1; o < 16; o <<= 1)                  // prefix minima inside the block of 16 lanes
            { const u32 t = __shfl_up(kk, o, 16);
              if (c >= o) kk = min(kk, t);
            }
        }
      if (p < SP_TILE + ((W + 15) & ~15))
        keys[SP_KIDX(p)] = kk;
    }
  __syncthreads();

  // ---- 3. sliding-window minimum for the thread's 16 k-mer starts ------------------------
  if (fastmin)
    { // window of start c ends at offset e = c + W - 1 from the block start: block t + (e >> 4)
      const int q0 = (W - 1) >> 4, r0 = (W - 1) & 15;
      u32 fa = 0xffffffffu;                                // whole blocks t+1 .. t+q0-1 (q0 <= 3: W <= 58)
      if (q0 > 1) fa = keys[SP_KIDX(16 * (tid + 1) + 15)];
      if (q0 > 2) fa = min(fa, keys[SP_KIDX(16 * (tid + 2) + 15)]);
      const u32 fb = (q0 >= 1) ? min(fa, keys[SP_KIDX(16 * (tid + q0) + 15)]) : fa;   // .. t+q0
      const int pbase = 17 * (tid + q0);                   // SP_KIDX of the first slot of block t+q0
#pragma unroll
      for (int c = 0; c < SP_CH; c++)
        { const int off = r0 + c;                          // uniform: 0 .. 30
          const u32 pre = keys[pbase + off + (off >> 4)];
          mk[c] = min(min(mk[c], (off < 16) ? fa : fb), pre);
        }
    }
  else
    {
#pragma unroll
      for (int c = 0; c < SP_CH; c++)
        { u32 m = 0xffffffffu;
#pragma unroll 1
          for (int j = 0; j < W; j++)
            m = min(m, keys[SP_KIDX(i0 + c + j)]);
          mk[c] = m;
        }
    }

  // ---- 4. validity of each k-mer: no invalid base in [i, i+K) ------------------------------
  //      K >= 16: the window of start c (in word t) is the rest of word t from bit c, the whole words
  //      t+1 .. t+kq-1 and the first c+kr bits of the word pair (t+kq, t+kq+1), K = 16 kq + kr -- so the
  //      valid starts of a thread are one interval [fls(m0), ctz(E) - kr], a dozen instructions per
  //      THREAD (the prefix counts this replaces cost 14 per base)
  u32 vmask = sp_clear_starts(inv16, tid, K);
  if (PACKED)                                            // ... and no read end in [i, i + K - 1)
    vmask &= sp_clear_starts(bnd16, tid, K - 1);
  lastkey[tid] = mk[SP_CH - 1];
  if (tid < a.nbuckets) bcnt2[tid] = 0;
  vlast[tid]   = (uint8_t) ((vmask >> (SP_CH - 1)) & 1u);
  __syncthreads();
  if (tid < 16)
    sbits[SP_THREADS + tid] = 0xffffu;      // past the tile everything is a boundary (published by the scan's barriers)

  // ---- 5. super-mer starts: valid and (first of tile | previous invalid | new minimizer) ---
  u32 smask;
  { u32 pk = (tid > 0) ? lastkey[tid - 1] : 0xffffffffu;
    const u32 pv = (tid > 0) ? (u32) vlast[tid - 1] : 0u;
    u32 kc = 0;
#pragma unroll
    for (int c = 0; c < SP_CH; c++)
      { kc |= ((mk[c] != pk) ? 1u : 0u) << c;
        pk = mk[c];
      }
    smask = vmask & (~((vmask << 1) | pv) | kc) & 0xffffu;
  }
  // boundaries (a start, or an invalid k-mer) for the length search of step 6; the window minima go to LDS
  // (the prefix minima in `keys` are dead: every thread passed the barrier above), position-indexed
  sbits[tid] = (uint16_t) (smask | (~vmask & 0xffffu));
#pragma unroll
  for (int c = 0; c < SP_CH; c++)
    keys[17 * tid + c] = mk[c];                            // = SP_KIDX(i0 + c): conflict-free dword stores, one address

  // one scan for both: super-mer starts before this thread (low half) and the tile's valid k-mers (total, high half)
  u32 nstart_total;
  u32 sidx0;
  { u32 tot;
    sidx0 = fk_block_exscan_256<u32>((u32) __popc(smask) | ((u32) __popc(vmask) << 16), tmp32, &tot) & 0xffffu;
    nstart_total = tot & 0xffffu;
    // instances = valid k-mers in the tile, spread over 64 counters (summed by the host)
    if (tid == 0 && (tot >> 16) != 0)
      atomicAdd(&a.counts[256 + (blockIdx.x & 63)], (u64) (tot >> 16));
  }
  // (the scan's barriers also publish sbits and the minima)

  // ---- 6. the super-mers of the tile, one per thread -----------------------------------------
  //      A thread lists the positions of its own starts (a loop over the set bits of smask: ~4 rounds per
  //      wave); after that thread s owns super-mer s: position from the list, minimizer from LDS, length
  //      from the boundary bits.  (Walking the 16 positions of a thread with a branch per position kept 6 %
  //      of the lanes busy and was a third of the kernel's instructions.)  The list takes SP_PL starts; a
  //      tile with more (never seen on reads: a start every four bases) is taken in several rounds.
#define SP_LIST(base)                                                          \
  { u32 sm_ = smask, k_ = sidx0 - (base);                                      \
    while (sm_ != 0)                                                           \
      { const int c_ = __ffs((int) sm_) - 1;                                   \
        sm_ &= sm_ - 1;                                                        \
        if (k_ < (u32) SP_PL) pos16[k_] = (uint16_t) (i0 + c_);                \
        k_ += 1;                                                               \
      }                                                                        \
  }
  SP_LIST(0u);
  __syncthreads();
  // bucket and rank inside its bucket of the thread's starts tid and tid + 256 of the first round (the returning LDS
  // atomic that counts the bucket hands out the rank: no second atomic when the records are placed)
  u32 bsave = 0, rank0 = 0, rank1 = 0;
  if (one)
    { if (tid == 0) bcnt[0] = nstart_total;
      rank0 = tid; rank1 = tid + SP_THREADS;
    }
  else
    for (u32 base = 0; base < nstart_total; base += SP_PL)
      { if (base != 0)
          { __syncthreads();
            SP_LIST(base);
            __syncthreads();
          }
        const u32 lim = min(nstart_total, base + (u32) SP_PL);
        for (u32 s = base + tid; s < lim; s += SP_THREADS)
          { const int ip = pos16[s - base];
            const u32 key = keys[SP_KIDX(ip)];
            const u32 b   = a.mbucket[key >> 15];
            u32 rk = 0;
            if (rec || b != a.skipb)
              rk = atomicAdd(&bcnt[b], 1u);             // (recording: every bucket is counted, the row goes to tile_cnt)
            if (rec && !(b >= (u32) a.gb0 && b < (u32) a.gb1))
              atomicAdd(&nother, 1u);
            if (s == (u32) tid) { bsave |= b; rank0 = rk; }
            else if (s == (u32) tid + SP_THREADS) { bsave |= b << 16; rank1 = rk; }
          }
      }
  __syncthreads();
  if (EMIT && rec && tid == SP_THREADS - 1)
    { // room for this tile's other entries in sub-region (tile mod 64)
      const u32 lane64 = blockIdx.x & 63u;
      const u64 sub = (u64) (a.ent_cap / 64);
      u64 at = 0;
      if (nother != 0)
        { at = atomicAdd(&a.ent_cursor[(size_t) lane64 * a.cstride], (u64) nother);
          if (at + nother > sub)
            { *a.overflowed = 2;
              at = 0;
              nother = 0;
            }
        }
      ebase = (u64) lane64 * sub + at;
      a.tile_ent[a.tile0 + blockIdx.x] = (ebase << 13) | (u64) nother;
    }

  if (!EMIT)
    { if (tid < a.nbuckets && bcnt[tid] != 0)
        { // a sampled plan spreads the bucket counters over 32 copies 2 KB apart: every tile adds to every
          // bucket's counter, and one address takes ~80 M atomics per second (the plan of configs[2] took 24.6 ms)
          if (a.plan != NULL) atomicAdd(&a.plan[((blockIdx.x & 31u) << 8) + tid], (u64) bcnt[tid]);
          else                atomicAdd(&a.counts[tid], (u64) bcnt[tid]);
        }
      return;
    }

  // ---- 7. reserve room, build and write the records ------------------------------------------
  //      One returning global atomic per bucket and tile reserves the tile's run in the bucket's stream: a trip to
  //      the memory side.  It is issued first and its answer is needed last: the records of the thread's (up to two)
  //      super-mers are built in registers meanwhile -- nothing on that stretch waits for a global load (the bucket and
  //      rank of a start were kept from step 6; region start and limit of a bucket were read before the atomic) --
  //      and only the store waits for the reservation, behind the barrier that publishes it.
  const int sww = a.sww;
  const int lenw = a.smer_bytes >> 2;
  const int lensh = 24 - 8 * (a.smer_bytes & 3);
  const u32 lbm = (1u << SP_LB) - 1u;
  const bool mine = (tid < a.nbuckets && bcnt[tid] != 0 && (!rec || (tid >= a.gb0 && tid < a.gb1)));
  u64 rb = 0, lm = ~0ull, res = 0;
  u32 mycnt = 0;
  if (mine)
    { rb = a.rbase[tid];
      if (a.limit != NULL) lm = a.limit[tid];
      mycnt = bcnt[tid];
    }
  if (rec && tid < a.nbuckets)
    a.tile_cnt[(size_t) (a.tile0 + blockIdx.x) * a.nbuckets + tid] = (uint16_t) bcnt[tid];
  if (mine)
    res = atomicAdd(&a.cursor[(((size_t) tid << a.lstreams) + strm) * a.cstride], (u64) mycnt);

  // (i, flip, n) of the super-mer that starts at list entry sl; its minimizer's key
  auto describe = [&](u32 sl, int &i, u32 &flip, int &n, u32 &key)
    { i   = pos16[sl];
      key = keys[SP_KIDX(i)];
      // strand of the minimizer: its 7-mer against the reverse complement, both from the packed words
      const int pm = (int) ((key >> 1) & 0x3fffu);
      flip = ((sp_window(rcw, R - 7 - pm) >> 18) < (sp_window(fwd, pm) >> 18)) ? 1u : 0u;
      // length: distance to the next boundary, at most W <= 58 positions on; 64 (80 for k > 53) boundary bits
      // from the word of i
      const int qi = i >> 4, ci = i & 15;
      u64 ab = ((u64) sbits[qi] | ((u64) sbits[qi + 1] << 16) | ((u64) sbits[qi + 2] << 32)
                | ((u64) sbits[qi + 3] << 48)) >> (ci + 1);
      if (W > 47)
        ab |= (u64) sbits[qi + 4] << (63 - ci);
      n = __ffsll((long long) ab);
    };
  // where record `rank` of bucket b goes (bbase: start of the chunk the tile's run begins in; bcnt: offset in it)
  auto slot_of = [&](u32 b, u32 rank, u64 &slot) -> bool
    { const u64 A = bbase[b];
      const u32 t = bcnt[b] + rank;
      slot = A + ((u64) (t >> SP_LB) << (SP_LB + a.lstreams)) + (u64) (t & lbm);
      return (A != ~0ull);
    };
  auto put_generic = [&](int i, u32 flip, int n, u32 *dst)
    { const int  L   = n - 1 + K;
      const u32 *arr = flip ? rcw : fwd;
      const int  st  = flip ? (R - (i + L)) : i;
      if (sww == 5)
        sp_put_record5(arr, st, L, ((u32) (n - 1)) << lensh, lenw, dst);
      else
        for (int q = 0; q < sww; q++)
          { u32 x = 0;
            const int rem = L - 16 * q;
            if (rem > 0)
              { x = sp_window(arr, st + 16 * q);
                if (rem < 16)
                  x &= ~(0xffffffffu >> (2 * rem));
              }
            if (q == lenw)
              x |= ((u32) (n - 1)) << lensh;
            dst[q] = __builtin_bswap32(x);
          }
    };

  const bool piped = (nstart_total <= (u32) SP_PL);       // all starts are in the list: two per thread at most
  sp_rec5 r0, r1;
  u32 d0 = 0, d1 = 0;                                     // hash digit 0 of the two records (a.dig)
  u32 m0 = 0, m1 = 0;                                     // i | flip << 12 | n << 13 | b << 20; bit 31: a record to place, bit 30: an entry to record
  if (piped)
    {
#pragma unroll
      for (int h = 0; h < 2; h++)
        { const u32 sl = (u32) tid + (u32) h * SP_THREADS;
          if (sl < nstart_total)
            { int i, n; u32 flip, key;
              describe(sl, i, flip, n, key);
              const u32 b = one ? 0u : ((bsave >> (16 * h)) & 0xffffu);
              const u32 m = (u32) i | (flip << 12) | ((u32) n << 13) | (b << 20);
              bool place = true;
              if (rec)
                { if (!(b >= (u32) a.gb0 && b < (u32) a.gb1))
                    { if (h == 0) m0 = m | 0x40000000u; else m1 = m | 0x40000000u;   // an entry: written behind the barrier
                      place = false;                                                 // (ebase is published there)
                    }
                }
              else if (b == a.skipb)
                place = false;
              if (place)
                { if (h == 0) m0 = m | 0x80000000u; else m1 = m | 0x80000000u;
                  if (sww == 5)
                    { const int  L   = n - 1 + K;
                      const u32 *arr = flip ? rcw : fwd;
                      const int  st  = flip ? (R - (i + L)) : i;
                      sp_put_record5(arr, st, L, ((u32) (n - 1)) << lensh, lenw, (h == 0) ? r0.w : r1.w);
                      if (a.dig != NULL)
                        { u32 ha, hb;
                          fk_rec_hash<5>((h == 0) ? r0.w : r1.w, 20, ha, hb);
                          if (h == 0) d0 = hb & 0xffffu; else d1 = hb & 0xffffu;
                        }
                    }
                }
            }
        }
    }
  if (mine)
    { u64 A = rb + ((((res >> SP_LB) << a.lstreams) + strm) << SP_LB);
      const u32 off = (u32) res & lbm;
      const u32 tl  = off + mycnt - 1;                      // the run's last record: the highest slot
      const u64 last = A + ((u64) (tl >> SP_LB) << (SP_LB + a.lstreams)) + (u64) (tl & lbm);
      if ((int64_t) last >= a.cap || last >= lm)
        { *a.overflowed = 1;                                // the caller starts over with wider regions
          A = ~0ull;
        }
      bbase[tid] = A;
      bcnt[tid]  = off;
    }
  __syncthreads();

  if (piped)
    {
#pragma unroll
      for (int h = 0; h < 2; h++)
        { const u32 m = (h == 0) ? m0 : m1;
          if ((m & 0x40000000u) && nother != 0)
            a.ent[ebase + atomicAdd(&nother2, 1u)] = m & 0x0fffffffu;
          if (m & 0x80000000u)
            { const u32 b = (m >> 20) & 0x7ffu;
              u64 slot;
              if (!slot_of(b, (h == 0) ? rank0 : rank1, slot))
                continue;
              u32 *dst = a.out + slot * sww;
              const int i = (int) (m & 0xfffu);
              const u32 flip = (m >> 12) & 1u;
              if (POS)
                a.pos[slot] = ((u64) (t0 + i) << 1) | flip;
              if (sww == 5)
                { *(sp_rec5 *) dst = (h == 0) ? r0 : r1;
                  if (a.dig != NULL)
                    { const u32 dd = (h == 0) ? d0 : d1;
                      a.dig[slot]  = (uint8_t) dd;
                      if (a.dig2 != NULL) a.dig2[slot] = (uint8_t) (dd >> 8);
                    }
                }
              else
                put_generic(i, flip, (int) ((m >> 13) & 0x7fu), dst);
            }
        }
      return;
    }
  // a tile with more starts than the list takes (never seen on reads): round by round, ranks from a second counter
  for (u32 base = 0; base < nstart_total; base += SP_PL)
    { __syncthreads();
      SP_LIST(base);
      __syncthreads();
      const u32 lim = min(nstart_total, base + (u32) SP_PL);
      for (u32 s = base + tid; s < lim; s += SP_THREADS)
        { int i, n; u32 flip, key;
          describe(s - base, i, flip, n, key);
          const u32 b = one ? 0u : (u32) a.mbucket[key >> 15];
          if (rec)
            { if (!(b >= (u32) a.gb0 && b < (u32) a.gb1))
                { if (nother != 0)
                    a.ent[ebase + atomicAdd(&nother2, 1u)] = (u32) i | (flip << 12) | ((u32) n << 13) | (b << 20);
                  continue;
                }
            }
          else if (b == a.skipb)
            continue;
          u64 slot;
          if (!slot_of(b, one ? s : atomicAdd(&bcnt2[b], 1u), slot))
            continue;
          if (POS)
            a.pos[slot] = ((u64) (t0 + i) << 1) | flip;
          put_generic(i, flip, n, a.out + slot * sww);
          if (a.dig != NULL && sww == 5)
            { sp_rec5 rr;
              const int  L   = n - 1 + K;
              sp_put_record5(flip ? rcw : fwd, flip ? (R - (i + L)) : i, L, ((u32) (n - 1)) << lensh, lenw, rr.w);
              u32 ha, hb;
              fk_rec_hash<5>(rr.w, 20, ha, hb);
              a.dig[slot]  = (uint8_t) (hb & 0xffu);
              if (a.dig2 != NULL) a.dig2[slot] = (uint8_t) ((hb >> 8) & 0xffu);
            }
        }
    }
#undef SP_LIST
}

// tile_cnt rows of SP_RCH consecutive tiles -> exclusive prefixes inside the chunk (in place) + the chunk's totals
__global__ __launch_bounds__(256) void k_split_chunk(uint16_t *__restrict__ tile_cnt, int64_t ntiles, int nb,
                                                     u32 *__restrict__ chunk_tot)
{ const int64_t ch = (int64_t) blockIdx.x * (256 / 64) + (threadIdx.x >> 6);     // one wave per chunk, lane = bucket (+64, ...)
  const int64_t t0 = ch * SP_RCH;
  if (t0 >= ntiles) return;
  for (int b = threadIdx.x & 63; b < nb; b += 64)
    { u32 run = 0;
      for (int t = 0; t < SP_RCH && t0 + t < ntiles; t++)
        { const size_t at = (size_t) (t0 + t) * nb + b;
          const u32 c = tile_cnt[at];
          tile_cnt[at] = (uint16_t) run;
          run += c;
        }
      chunk_tot[(size_t) ch * nb + b] = run;
    }
}

// Exclusive scan of every bucket's chunk totals over all chunks, in three steps: the chunks are cut into np
// partitions of `len` rows; (1) a thread (r, b) of workgroup p adds up bucket b over partition p * R + r (R = 256 /
// nb partitions per workgroup: the lanes of a row read consecutive words), (2) one workgroup per bucket scans the
// np partition sums, (3) the threads of (1) write the running bases of their rows.  (One workgroup per bucket
// walking all 2.3 M rows of configs[2] with a stride of nb words took 11 ms.)
#define SP_CSP 1024                       // workgroups of steps 1 and 3
__global__ __launch_bounds__(256) void k_split_chunksum(const u32 *__restrict__ chunk_tot, int64_t nchunks, int nb, int64_t len,
                                                        u64 *__restrict__ part)
{ const int R = 256 / nb, r = threadIdx.x / nb, b = threadIdx.x % nb;
  if (r >= R) return;
  const int64_t pr = (int64_t) blockIdx.x * R + r;
  const int64_t lo = pr * len, hi = (lo + len < nchunks) ? lo + len : nchunks;
  u64 sum = 0;
  for (int64_t c = lo; c < hi; c++)
    sum += chunk_tot[(size_t) c * nb + b];
  part[(size_t) pr * nb + b] = sum;
}

__global__ __launch_bounds__(256) void k_split_chunkscan(u64 *__restrict__ part, int64_t np, int nb, u64 *__restrict__ total)
{ __shared__ u64 tmp[8];
  const int b = blockIdx.x;
  const int64_t per = (np + 255) / 256;
  const int64_t lo = (int64_t) threadIdx.x * per, hi = (lo + per < np) ? lo + per : np;
  u64 mine = 0;
  for (int64_t c = lo; c < hi; c++)
    mine += part[(size_t) c * nb + b];
  u64 tot;
  u64 run = fk_block_exscan_256<u64>(mine, tmp, &tot);
  for (int64_t c = lo; c < hi; c++)
    { const u64 v = part[(size_t) c * nb + b];
      part[(size_t) c * nb + b] = run;
      run += v;
    }
  if (threadIdx.x == 0)
    total[b] = tot;
}

__global__ __launch_bounds__(256) void k_split_chunkbase(const u32 *__restrict__ chunk_tot, int64_t nchunks, int nb, int64_t len,
                                                         const u64 *__restrict__ part, u64 *__restrict__ chunk_base)
{ const int R = 256 / nb, r = threadIdx.x / nb, b = threadIdx.x % nb;
  if (r >= R) return;
  const int64_t pr = (int64_t) blockIdx.x * R + r;
  const int64_t lo = pr * len, hi = (lo + len < nchunks) ? lo + len : nchunks;
  u64 run = part[(size_t) pr * nb + b];
  for (int64_t c = lo; c < hi; c++)
    { chunk_base[(size_t) c * nb + b] = run;
      run += chunk_tot[(size_t) c * nb + b];
    }
}

// A later pass of the multi-pass split: the tile's entries of the first pass (position, flip, length, bucket of
// every super-mer that pass did not emit) + the reads -> the records of buckets [gb0, gb1).  Steps 1 and 7 of
// k_split only: no keys, no minima, no validity, no start masks.  (Persistent workgroups that keep the loads of the
// next tile and the descriptor of the one after in flight were tried: 473 instead of 459 ms of split time per
// configs[2] step -- the kernel is not waiting for its two dependent round trips either.)
template <bool PACKED>
__global__ __launch_bounds__(SP_THREADS) void k_split_replay(SplitArgs a)
{ __shared__ u32 fwd[SP_WORDS];
  __shared__ u32 rcw[SP_WORDS];
  __shared__ __attribute__((aligned(8))) u64 bbase[256];
  __shared__ u32 bcnt2[256];

  const int     tid = threadIdx.x;
  const int     K   = a.kmer;
  const int     nw  = SP_TILE / 16 + (K + 14) / 16;
  const int     R   = nw * 16;
  // A workgroup takes SP_RTW consecutive tiles: one tile per workgroup was 36 M workgroups of ~30 instructions per
  // thread at configs[2], and the rate at which the dispatcher starts workgroups bounded the kernel (a pass that
  // only loaded and converted the bases took 60 % of the time of the whole kernel).
  // ... and the loads of the next tile -- its entry descriptor, its bases, the three terms of its bucket offsets, none
  // of which depends on another -- are in flight while the current one is worked on: with eight workgroups of 4 KB
  // per CU and a trip to memory per tile, the bytes in flight bounded the pass at half the copy rate.
  u64   te_n = 0, bb_n = 0;
  uint4 v0_n = make_uint4(0u, 0u, 0u, 0u), v1_n = v0_n;
  auto fetch = [&](int64_t tile)
    { const int64_t t0 = tile * SP_TILE;
      te_n = a.tile_ent[tile];
      const int64_t g0 = t0 + (int64_t) tid * 16, g1 = t0 + (int64_t) (tid + SP_THREADS) * 16;
      v0_n = make_uint4(0u, 0u, 0u, 0u); v1_n = v0_n;
#ifdef FK_ABLATION
      if (a.abl & 2) { v0_n = make_uint4(0x61636774u + tid, 0x74676361u, 0x61616161u, 0x63636363u); v1_n = v0_n; }
      else
#endif
      if (PACKED)                                            // one dword of codes per 16 positions (.x of the pair)
        { if (g0 < a.nbytes) v0_n.x = ((const u32 *) a.bases)[g0 >> 4];
          if (tid + SP_THREADS < nw && g1 < a.nbytes) v1_n.x = ((const u32 *) a.bases)[g1 >> 4];
        }
      else
      { if (g0 + 16 <= a.nbytes) v0_n = *(const uint4 *) (a.bases + g0);
        if (tid + SP_THREADS < nw && g1 + 16 <= a.nbytes) v1_n = *(const uint4 *) (a.bases + g1);
      }
      bb_n = 0;
      if (tid >= a.gb0 && tid < a.gb1)
        bb_n = a.rbase[tid] + a.chunk_base[(size_t) (tile / SP_RCH) * a.nbuckets + tid]
             + (u64) a.tile_cnt[(size_t) tile * a.nbuckets + tid];
    };
  const int64_t tile_first = a.tile0 + (int64_t) blockIdx.x * SP_RTW;
  if (tile_first < a.tile_end)
    fetch(tile_first);
  for (int rt = 0; rt < SP_RTW; rt++)
  { const int64_t tile = tile_first + rt;
    if (tile >= a.tile_end)
      break;
    if (rt > 0)
      __syncthreads();                                       // the previous tile's arrays are free
  const int64_t t0  = tile * SP_TILE;
  const u64     te  = te_n, bb = bb_n;
  const uint4   v0  = v0_n, v1 = v1_n;
  if (rt + 1 < SP_RTW && tile + 1 < a.tile_end)
    fetch(tile + 1);
  const u32     cnt = (u32) (te & 0x1fffu);
  if (cnt == 0)
    continue;
  const u64     eb  = te >> 13;

  bcnt2[tid] = 0;
  // a thread builds the records of the entries tid, tid + 256, ...: the first one is fetched now and used after
  // the bases are converted (no staging in LDS: 5.3 KB per workgroup, the CU's wave limit decides the occupancy)
  u32 e0 = 0;
  if ((u32) tid < cnt)
    e0 = a.ent[eb + tid];
  // where this tile's records of bucket b go: region start + the bucket's records in earlier chunks + in the
  // earlier tiles of this chunk -- exact, no reservation needed
  bbase[tid] = bb;
  for (int q = tid; q < SP_WORDS; q += SP_THREADS)
    { u32 word = 0, bad = 0;
      if (q < nw)
        { const int64_t g = t0 + (int64_t) q * 16;
          uint4 v = (q == tid) ? v0 : v1;
          if (PACKED)
            word = __builtin_bswap32(v.x);
          else
          {
          if (!(g + 16 <= a.nbytes))
            { u32 d[4] = { 0u, 0u, 0u, 0u };                   // the last bytes of the input, one by one
              for (int j = 0; j < 16; j++)
                if (g + j < a.nbytes)
                  d[j >> 2] |= (u32) a.bases[g + j] << (8 * (j & 3));
              v = make_uint4(d[0], d[1], d[2], d[3]);
            }
          sp_pack16(v, word, bad);
          }
        }
      fwd[q] = word;
    }
  __syncthreads();
  for (int q = tid; q < SP_WORDS; q += SP_THREADS)
    { u32 x = 0;
      if (q < nw)
        { const u32 y = __builtin_bitreverse32(fwd[nw - 1 - q]);
          x = ~(((y >> 1) & 0x55555555u) | ((y & 0x55555555u) << 1));
        }
      rcw[q] = x;
    }
  __syncthreads();

  const int sww = a.sww;
  const int lenw = a.smer_bytes >> 2;
  const int lensh = 24 - 8 * (a.smer_bytes & 3);
#ifdef FK_ABLATION
  if (a.abl & 4)
    { if (fwd[tid] == 0x12345u && rcw[tid] == 0x54321u) *a.overflowed = 2;
      continue;
    }
#endif
  for (u32 s = tid; s < cnt; s += SP_THREADS)
    { const u32 e    = (s == (u32) tid) ? e0 : a.ent[eb + s];
      const int i    = e & 0xfffu;
      const u32 flip = (e >> 12) & 1u;
      const int n    = (e >> 13) & 0x7fu;
      const u32 b    = e >> 20;
      if (!(b >= (u32) a.gb0 && b < (u32) a.gb1))
        continue;
      const u64 slot = bbase[b] + atomicAdd(&bcnt2[b], 1u);
      if ((int64_t) slot >= a.cap || (a.limit != NULL && slot >= a.limit[b]))
        { *a.overflowed = 1;
          continue;
        }
      const int  L   = n - 1 + K;
      const u32 *arr = flip ? rcw : fwd;
      const int  st  = flip ? (R - (i + L)) : i;
      u32 *dst = a.out + slot * sww;
#ifdef FK_ABLATION
      if (a.abl & 1)
        { u32 x = sp_window(arr, st) ^ sp_window(arr, st + 16) ^ sp_window(arr, st + 32) ^ sp_window(arr, st + 48);
          if (x == 0x13572468u) dst[0] = x;
          continue;
        }
#endif
      if (sww == 5)
        { sp_rec5 rr;
          sp_put_record5(arr, st, L, ((u32) (n - 1)) << lensh, lenw, rr.w);
          *(sp_rec5 *) dst = rr;
          if (a.dig != NULL)                                   // first digit of the grouping sort, as k_split writes it
            { u32 ha, hb;
              fk_rec_hash<5>(rr.w, 20, ha, hb);
              a.dig[slot]  = (uint8_t) (hb & 0xffu);
              if (a.dig2 != NULL) a.dig2[slot] = (uint8_t) ((hb >> 8) & 0xffu);
            }
        }
      else
      for (int q = 0; q < sww; q++)
        { u32 x = 0;
          const int rem = L - 16 * q;
          if (rem > 0)
            { x = sp_window(arr, st + 16 * q);
              if (rem < 16)
                x &= ~(0xffffffffu >> (2 * rem));
            }
          if (q == lenw)
            x |= ((u32) (n - 1)) << lensh;
          dst[q] = __builtin_bswap32(x);
        }
    }
  }
}

// Closes the ragged end of the chunk-interleaved streams (SplitArgs.lstreams): afterwards the records of bucket b
// are rbase[b] .. rbase[b] + sum of its cursors, contiguous.  One workgroup per bucket: the chunks of the rows that
// are not full in every stream move down in (row, stream) order; a record never moves up and never past an unread
// one, so batches of "all read, barrier, all write" in that order are a memmove.  A few chunks per bucket.
#define SP_CT 1024
__global__ __launch_bounds__(SP_CT) void k_split_compact(u32 *__restrict__ out, int sww, const u64 *__restrict__ cursor,
                                                         int cstride, int lstreams, const u64 *__restrict__ rbase, int b0,
                                                         const u32 *__restrict__ overflowed, uint8_t *__restrict__ dig,
                                                         uint8_t *__restrict__ dig2)
{ const int b = b0 + blockIdx.x, C = 1 << lstreams, tid = threadIdx.x;
  __shared__ u64 L[1 << SP_LSTREAMS];
  if (*overflowed == 1u)                 // a region was too small: records were dropped, the caller starts over
    return;
  if (tid < C)
    L[tid] = cursor[(((size_t) b << lstreams) + tid) * cstride];
  __syncthreads();
  u64 rmin = ~0ull, rmax = 0;
  for (int c = 0; c < C; c++)
    { rmin = min(rmin, L[c] >> SP_LB);
      rmax = max(rmax, (L[c] + ((1u << SP_LB) - 1u)) >> SP_LB);
    }
  const u64 r0 = rbase[b];
  u64 dst = (r0 + ((rmin << lstreams) << SP_LB)) * (u64) sww;
  for (u64 r = rmin; r < rmax; r++)
    for (int c = 0; c < C; c++)
      { const int64_t have = (int64_t) L[c] - (int64_t) (r << SP_LB);
        if (have <= 0)
          continue;
        const u64 n   = (u64) (have < (1 << SP_LB) ? have : (1 << SP_LB)) * (u64) sww;
        const u64 src = (r0 + (((r << lstreams) + (u64) c) << SP_LB)) * (u64) sww;
        if (src != dst && dig != NULL)                  // the records' digit bytes move with them (a chunk: <= 1024 of them)
          { const u64 nr = n / (u64) sww, sr = src / (u64) sww, dr = dst / (u64) sww;
            const uint8_t v = ((u64) tid < nr) ? dig[sr + tid] : (uint8_t) 0;
            const uint8_t v2 = ((u64) tid < nr && dig2 != NULL) ? dig2[sr + tid] : (uint8_t) 0;
            __syncthreads();
            if ((u64) tid < nr)
              { dig[dr + tid] = v;
                if (dig2 != NULL) dig2[dr + tid] = v2;
              }
            __threadfence_block();
            __syncthreads();
          }
        if (src != dst)
          for (u64 off = 0; off < n; off += 4 * SP_CT)
            { u32 v[4];
#pragma unroll
              for (int u = 0; u < 4; u++)
                { const u64 i = off + (u64) u * SP_CT + tid;
                  v[u] = (i < n) ? out[src + i] : 0u;
                }
              __syncthreads();
#pragma unroll
              for (int u = 0; u < 4; u++)
                { const u64 i = off + (u64) u * SP_CT + tid;
                  if (i < n)
                    out[dst + i] = v[u];
                }
              __threadfence_block();
              __syncthreads();
            }
        dst += n;
      }
}

#if !defined(FK_HOST_EMU) || defined(FK_EMU_FULL)      // (host halves stay out of the CPU tests' build of the kernels, tests/csrc/hip_emu.h)
// the cursors of the streamed emit: (256 buckets x 2^SP_LSTREAMS streams + 64 entry sub-regions), 4 KB apart
static u64 *sp_cursors(fk_ctx *ctx)
{ if (ctx->d_cursors == NULL
      && hipMalloc((void **) &ctx->d_cursors, ((size_t) (256 << SP_LSTREAMS) + 64) * FK_CURSOR_STRIDE * sizeof(u64)) != hipSuccess)
    { ctx->d_cursors = NULL;
      fk_set_error(ctx, "split: no memory for the cursors");
    }
  return (ctx->d_cursors);
}

// after the emit over buckets [b0, b1): close the stream ends, then bring the cursors to the host; totals[b - b0]
// = records of bucket b.  Synchronises the stream.
static int sp_finish_streams(fk_ctx *ctx, const SplitArgs &a, int b0, int b1, int64_t *totals)
{ hipStream_t s = ctx->stream;
  const int C = 1 << a.lstreams;
  hipLaunchKernelGGL(k_split_compact, dim3((unsigned) (b1 - b0)), dim3(SP_CT), 0, s, a.out, a.sww, (const u64 *) a.cursor,
                     a.cstride, a.lstreams, a.rbase, b0, (const u32 *) a.overflowed, a.dig, a.dig2);
  FK_LAUNCH_CHECK(ctx);
  u64 *h = ctx->h_scratch + 8192;                  // pinned; (b1 - b0) * C <= 2048 words
  FK_HIP(ctx, hipMemcpy2DAsync(h, sizeof(u64), a.cursor + ((size_t) b0 << a.lstreams) * a.cstride, (size_t) a.cstride * sizeof(u64),
                               sizeof(u64), (size_t) (b1 - b0) * C, hipMemcpyDeviceToHost, s));
  FK_HIP(ctx, hipStreamSynchronize(s));
  for (int b = b0; b < b1; b++)
    { int64_t t = 0;
      for (int c = 0; c < C; c++)
        t += (int64_t) h[(size_t) (b - b0) * C + c];
      totals[b - b0] = t;
    }
  return (FK_OK);
}

// A launch covers gridDim.x * blockDim.x < 2^32 work-items: inputs of more than 2^23 tiles (32 G bases)
// are taken in several launches (a single one silently wraps).
#define SP_MAXGRID (1 << 23)

template <bool EMIT, bool POS>
static void sp_launch(SplitArgs a, int64_t ngrid, hipStream_t s)
{ for (int64_t t = 0; t < ngrid; t += SP_MAXGRID)
    { a.tile0 = t;
      const int64_t nb = (ngrid - t < SP_MAXGRID) ? (ngrid - t) : SP_MAXGRID;
      if (a.roff != NULL)                              // packed reads (sp_packed_args)
        hipLaunchKernelGGL((k_split<EMIT, false, true>), dim3((unsigned) nb), dim3(SP_THREADS), (size_t) a.nbuckets * 16, s, a);
      else
        hipLaunchKernelGGL((k_split<EMIT, POS, false>), dim3((unsigned) nb), dim3(SP_THREADS), (size_t) a.nbuckets * 16, s, a);
    }
}

#endif   // FK_HOST_EMU

// For every tile, where its walk through the two sorted lists of a packed read set begins: the first read whose last
// position is not in front of the tile, the first invalid stretch that does not end in front of it (0xffffffff: none).
// One thread per 64 consecutive tiles: a binary search finds the element for its first tile, the others follow by
// walking the list forward -- the work is tiles + elements whatever the shape of the data (the first version gave every
// list element the tiles that begin inside it: one stretch of N at the end of 150 G positions was one thread writing
// 36 M entries).
#define SP_TIDX_CH 64
__global__ __launch_bounds__(256) void k_pk_tidx(const int64_t *__restrict__ roff, int64_t nreads, const int64_t *__restrict__ inv,
                                                 int64_t ninv, int64_t ntiles, int64_t tile_len, u32 *__restrict__ tidx)
{ const int64_t c = (int64_t) blockIdx.x * 256 + threadIdx.x;
  const int64_t t_lo = c * SP_TIDX_CH, t_hi = (t_lo + SP_TIDX_CH < ntiles) ? t_lo + SP_TIDX_CH : ntiles;
  if (t_lo >= ntiles)
    return;
  { // reads: last position of read j = roff[j + 1] - 1, non-decreasing
    const int64_t p0 = t_lo * tile_len;
    int64_t lo = 0, hi = nreads;
    while (lo < hi)
      { const int64_t mid = (lo + hi) >> 1;
        if (roff[mid + 1] - 1 < p0) lo = mid + 1; else hi = mid;
      }
    int64_t j = lo;
    for (int64_t t = t_lo; t < t_hi; t++)
      { while (j < nreads && roff[j + 1] - 1 < t * tile_len)
          j += 1;
        tidx[2 * t] = (j < nreads) ? (u32) j : 0xffffffffu;
      }
  }
  { // stretches: last position of stretch j = inv[2 j] + inv[2 j + 1] - 1, increasing
    const int64_t p0 = t_lo * tile_len;
    int64_t lo = 0, hi = ninv;
    while (lo < hi)
      { const int64_t mid = (lo + hi) >> 1;
        if (inv[2 * mid] + inv[2 * mid + 1] - 1 < p0) lo = mid + 1; else hi = mid;
      }
    int64_t j = lo;
    for (int64_t t = t_lo; t < t_hi; t++)
      { while (j < ninv && inv[2 * j] + inv[2 * j + 1] - 1 < t * tile_len)
          j += 1;
        tidx[2 * t + 1] = (j < ninv) ? (u32) j : 0xffffffffu;
      }
  }
}

#if !defined(FK_HOST_EMU) || defined(FK_EMU_FULL)
// pk != NULL: the split kernels of `a` take packed reads; builds the tile index in its arena slot
static int sp_packed_args(fk_ctx *ctx, SplitArgs &a, const fk_pkview *pk, int64_t ntiles)
{ a.roff = NULL; a.nreads = 0; a.inv = NULL; a.ninv = 0; a.tidx = NULL;
  if (pk == NULL)
    return (FK_OK);
  if (pk->roff == NULL || pk->nreads <= 0 || pk->ninv < 0 || (pk->ninv > 0 && pk->inv == NULL)
      || pk->nreads >= 0xffffffffll || pk->ninv >= 0xffffffffll)
    { fk_set_error(ctx, "packed reads: %lld reads, %lld invalid stretches -- between 1 and 2^32 - 2 reads, at most 2^32 - 2 stretches",
                   (long long) pk->nreads, (long long) pk->ninv);
      return (FK_EINVAL);
    }
  u32 *tidx = (u32 *) fk_slot(ctx, FK_SLOT_PK_TIDX, (ntiles + 2) * 8);      // a tile also looks at the entries of tile + 2
  if (tidx == NULL)
    return (FK_ENOMEM);
  hipStream_t s = ctx->stream;
  FK_HIP(ctx, hipMemsetAsync(tidx, 0xff, (size_t) (ntiles + 2) * 8, s));
  const int64_t nch = (ntiles + SP_TIDX_CH - 1) / SP_TIDX_CH;        // (entries ntiles, ntiles + 1 keep the memset's "none")
  hipLaunchKernelGGL(k_pk_tidx, dim3((unsigned) ((nch + 255) / 256)), dim3(256), 0, s, pk->roff, pk->nreads, pk->inv, pk->ninv,
                     ntiles, (int64_t) SP_TILE, tidx);
  FK_LAUNCH_CHECK(ctx);
  a.roff = pk->roff; a.nreads = pk->nreads; a.inv = pk->inv; a.ninv = pk->ninv; a.tidx = tidx;
  return (FK_OK);
}

// ---------------------------------------------------------------------------------------------
// counts_known: bucket_counts[] (and *nsuper, *ninst) hold the result of an earlier counting call
// on the same input, so only the emit kernel runs.
int fkx_split(fk_ctx *ctx, const void *d_bases, int64_t nbytes, void *d_out, int64_t cap,
              int64_t *nsuper, int64_t *ninst, int64_t *bucket_counts, bool counts_known, void *d_pos, const fk_pkview *pk)
{ hipStream_t s = ctx->stream;
  const int   K = ctx->prm.kmer;
  const int   nb = ctx->prm.nbuckets;
  u64 *d_counts = ctx->d_scratch;            // [0..255] buckets, [256] instances
  u64 *d_cursor = ctx->d_scratch + 512;      // [0..255]
  u32 *d_ovf    = (u32 *) (ctx->d_scratch + 1024);

  if (K < 8 || K > SP_MAXK)
    { fk_set_error(ctx, "k = %d outside the supported range [8,%d]", K, SP_MAXK);
      return (FK_EUNSUPPORTED);
    }
  if (counts_known && (bucket_counts == NULL || d_out == NULL))
    return (FK_EINVAL);
  if (!counts_known)
    { if (nsuper) *nsuper = 0;
      if (ninst) *ninst = 0;
      if (bucket_counts)
        for (int b = 0; b < nb; b++)
          bucket_counts[b] = 0;
    }
  if (nbytes < K)
    return (FK_OK);

  SplitArgs a = SplitArgs();
  a.bases = (const unsigned char *) d_bases;
  a.nbytes = nbytes;
  a.kmer = K;
  a.smer_bytes = ctx->wid.smer_bytes;
  a.sww = ctx->wid.smer_stride / 4;
  a.nbuckets = nb;
  a.mbucket = ctx->d_mbucket;
  a.counts = d_counts;
  a.cursor = d_cursor; a.cstride = 1;
  a.out = (u32 *) d_out;
  a.cap = cap;
  a.overflowed = d_ovf;
  a.tile_stride = 1;
  a.limit = NULL;
  a.pos = (u64 *) d_pos;
  a.skipb = 0x100u; a.ent = NULL;

  const int64_t nstarts = nbytes - K + 1;
  const int64_t ntiles  = (nstarts + SP_TILE - 1) / SP_TILE;
  if (pk != NULL && d_pos != NULL)
    { fk_set_error(ctx, "split: record positions are not available for packed reads");
      return (FK_EUNSUPPORTED);
    }
  { const int rcp = sp_packed_args(ctx, a, pk, ntiles);
    if (rcp != FK_OK) return (rcp);
  }

  FK_HIP(ctx, hipMemsetAsync(ctx->d_scratch, 0, 1032 * sizeof(u64), s));
  int64_t tot = 0;
  u64    *base = ctx->h_scratch + 512;          // pinned, so the async upload below is safe
  if (!counts_known)
    { sp_launch<false, false>(a, ntiles, s);
      FK_LAUNCH_CHECK(ctx);
      FK_HIP(ctx, hipMemcpyAsync(ctx->h_scratch, d_counts, 320 * sizeof(u64), hipMemcpyDeviceToHost, s));
      FK_HIP(ctx, hipStreamSynchronize(s));
      for (int b = 0; b < nb; b++)
        { base[b] = (u64) tot;
          tot += (int64_t) ctx->h_scratch[b];
          if (bucket_counts)
            bucket_counts[b] = (int64_t) ctx->h_scratch[b];
        }
      if (nsuper) *nsuper = tot;
      if (ninst)
        { int64_t t = 0;
          for (int x = 0; x < 64; x++)
            t += (int64_t) ctx->h_scratch[256 + x];
          *ninst = t;
        }
    }
  else
    for (int b = 0; b < nb; b++)
      { base[b] = (u64) tot;
        tot += bucket_counts[b];
      }
  if (cap == 0 || d_out == NULL)
    return (FK_OK);
  if (cap < tot)
    { fk_set_error(ctx, "super-mer buffer too small: %lld records needed, %lld given",
                   (long long) tot, (long long) cap);
      return (FK_EINVAL);
    }

  // exact regions: one stream per bucket (the cursors, zeroed above, count from the region starts)
  FK_HIP(ctx, hipMemcpyAsync(ctx->d_scratch + 768, base, nb * sizeof(u64), hipMemcpyHostToDevice, s));
  a.rbase = ctx->d_scratch + 768; a.lstreams = 0;
  if (d_pos != NULL)
    sp_launch<true, true>(a, ntiles, s);
  else
    sp_launch<true, false>(a, ntiles, s);
  FK_LAUNCH_CHECK(ctx);
  FK_HIP(ctx, hipMemcpyAsync(ctx->h_scratch, d_ovf, sizeof(u32), hipMemcpyDeviceToHost, s));
  FK_HIP(ctx, hipStreamSynchronize(s));
  if (*(u32 *) ctx->h_scratch != 0)
    { fk_set_error(ctx, "internal: super-mer emit overflowed its buffer");
      return (FK_EHIP);
    }
  return (FK_OK);
}

// Single-bucket fast path of the pipeline: size the output from a 1/32 sample of the tiles, emit
// once with overflow detection, and only if the estimate was too small fall back to the exact
// count-then-emit of fkx_split.  *d_out is the arena slot that received the records.
int fkx_split_fast(fk_ctx *ctx, const void *d_bases, int64_t nbytes, void **d_out, int64_t *nsuper,
                   int64_t *ninst, int64_t *bucket_counts, int64_t *bucket_offsets, const fk_pkview *pk, uint8_t **d_dig)
{ int64_t bc[256];
  if (d_dig != NULL)
    *d_dig = NULL;
  const bool want_dig = (d_dig != NULL && ctx->wid.smer_stride == 20);
  for (int b = 0; b < 256; b++)
    bucket_counts[b] = bucket_offsets[b] = 0;
  hipStream_t s = ctx->stream;
  const int   K = ctx->prm.kmer;
  u64 *d_counts = ctx->d_scratch;
  u64 *d_cursor = ctx->d_scratch + 512;
  u32 *d_ovf    = (u32 *) (ctx->d_scratch + 1024);
  const int stride = ctx->wid.smer_stride;

  *nsuper = 0; *ninst = 0; *d_out = NULL;
  if (K < 8 || K > SP_MAXK)
    { fk_set_error(ctx, "k = %d outside the supported range [8,%d]", K, SP_MAXK);
      return (FK_EUNSUPPORTED);
    }
  if (nbytes < K)
    return (FK_OK);
  if (ctx->prm.nbuckets == 1)
    { SplitArgs a = SplitArgs();
      a.bases = (const unsigned char *) d_bases;
      a.nbytes = nbytes;
      a.kmer = K;
      a.smer_bytes = ctx->wid.smer_bytes;
      a.sww = stride / 4;
      a.nbuckets = 1;
          a.mbucket = ctx->d_mbucket;
      a.counts = d_counts;
      a.cursor = d_cursor; a.cstride = 1;
      a.limit = NULL; a.pos = NULL; a.skipb = 0x100u; a.ent = NULL;
      a.overflowed = d_ovf;
      const int64_t nstarts = nbytes - K + 1;
      const int64_t ntiles  = (nstarts + SP_TILE - 1) / SP_TILE;
      const int     sample  = 32;
      { const int rcp = sp_packed_args(ctx, a, pk, ntiles);
        if (rcp != FK_OK) return (rcp);
      }
      if (ntiles >= 64 * sample)
        { FK_HIP(ctx, hipMemsetAsync(ctx->d_scratch, 0, 1032 * sizeof(u64), s));
          a.out = NULL; a.cap = 0; a.tile_stride = sample;
          sp_launch<false, false>(a, ntiles / sample, s);
          FK_LAUNCH_CHECK(ctx);
          FK_HIP(ctx, hipMemcpyAsync(ctx->h_scratch, d_counts, sizeof(u64), hipMemcpyDeviceToHost, s));
          FK_HIP(ctx, hipStreamSynchronize(s));
          const double per_tile = (double) ctx->h_scratch[0] / (double) (ntiles / sample);
          int64_t cap = (int64_t) (per_tile * (double) ntiles * 1.03) + 65536;
          void *out = fk_slot(ctx, FK_SLOT_SM_A, cap * stride);
          if (out == NULL)
            return (FK_ENOMEM);
          cap = ctx->slot_cap[FK_SLOT_SM_A] / stride;          // use the headroom too
          a.dig = want_dig ? fkx_dig_slot(ctx, cap) : NULL;      // (the plane of the second digit lies behind the first one's)
          a.dig2 = (a.dig != NULL && ctx->dig2_off > 0) ? a.dig + ctx->dig2_off : NULL;
          FK_HIP(ctx, hipMemsetAsync(ctx->d_scratch, 0, 1032 * sizeof(u64), s));
          a.out = (u32 *) out; a.cap = cap; a.tile_stride = 1;
          if (sp_cursors(ctx) == NULL)
            return (FK_ENOMEM);
          a.cursor = ctx->d_cursors; a.cstride = FK_CURSOR_STRIDE; a.lstreams = SP_LSTREAMS;
          a.rbase = ctx->d_scratch + 768;                        // zero: the region starts at record 0
          FK_HIP(ctx, hipMemsetAsync(ctx->d_cursors, 0, ((size_t) 1 << SP_LSTREAMS) * FK_CURSOR_STRIDE * sizeof(u64), s));
          sp_launch<true, false>(a, ntiles, s);
          FK_LAUNCH_CHECK(ctx);
          FK_HIP(ctx, hipMemcpyAsync(ctx->h_scratch, ctx->d_scratch, 1032 * sizeof(u64),
                                     hipMemcpyDeviceToHost, s));
          int64_t tot1 = 0;
          const int rcs = sp_finish_streams(ctx, a, 0, 1, &tot1);
          if (rcs != FK_OK)
            return (rcs);
          if (*(u32 *) (ctx->h_scratch + 1024) == 0)
            { *nsuper = tot1;
              { int64_t t = 0;
                for (int x = 0; x < 64; x++)
                  t += (int64_t) ctx->h_scratch[256 + x];
                *ninst = t;
              }
              *d_out  = out;
              if (d_dig != NULL) *d_dig = a.dig;
              bucket_counts[0] = *nsuper;
              return (FK_OK);
            }
          a.dig = NULL; a.dig2 = NULL;
          a.cursor = d_cursor; a.cstride = 1; a.lstreams = 0;
          // estimate too small (very uneven input): exact path below
        }
    }
  if (ctx->prm.nbuckets > 1)
    { // one emit pass into regions sized from a 1/32 tile sample (padded); exact pair on overflow
      int64_t cap = 0, offs[257];
      int rc = fkx_split_plan(ctx, d_bases, nbytes, &cap, offs, pk);
      if (rc != FK_OK)
        return (rc);
      if (cap > 0)
        { void *out = fk_slot(ctx, FK_SLOT_SM_A, cap * stride);
          if (out == NULL)
            return (FK_ENOMEM);
          uint8_t *dg = want_dig ? fkx_dig_slot(ctx, cap) : NULL;
          rc = fkx_split_planned(ctx, d_bases, nbytes, out, cap, offs, bc, ninst, 0, -1, 0, pk, dg);
          if (rc == FK_OK && d_dig != NULL)
            *d_dig = dg;
          if (rc == FK_OK)
            { int64_t tot = 0;
              for (int b = 0; b < ctx->prm.nbuckets; b++)
                { bucket_counts[b] = bc[b];
                  bucket_offsets[b] = offs[b];
                  tot += bc[b];
                }
              *nsuper = tot;
              *d_out = out;
              return (FK_OK);
            }
          if (rc != FK_ESTATE)
            return (rc);
        }
    }
  int rc = fkx_split(ctx, d_bases, nbytes, NULL, 0, nsuper, ninst, bc, false, NULL, pk);
  if (rc != FK_OK || *nsuper == 0)
    return (rc);
  void *out = fk_slot(ctx, FK_SLOT_SM_A, *nsuper * stride);
  if (out == NULL)
    return (FK_ENOMEM);
  *d_out = out;
  rc = fkx_split(ctx, d_bases, nbytes, out, *nsuper, nsuper, ninst, bc, true, NULL, pk);
  int64_t run = 0;
  for (int b = 0; b < ctx->prm.nbuckets && b < 256; b++)
    { bucket_counts[b] = bc[b];
      bucket_offsets[b] = run;
      run += bc[b];
    }
  return (rc);
}

// Sampled plan for the bucketed (sharded) split: estimated records per bucket from a 1/32 tile
// sample, padded by 5 %; region b starts at offsets[b].  Returns the total capacity needed.
int fkx_split_plan(fk_ctx *ctx, const void *d_bases, int64_t nbytes, int64_t *cap,
                   int64_t *offsets, const fk_pkview *pk)
{ hipStream_t s = ctx->stream;
  const int   K = ctx->prm.kmer;
  const int   nb = ctx->prm.nbuckets;
  *cap = 0;
  for (int b = 0; b <= nb; b++)
    offsets[b] = 0;
  if (K < 8 || K > SP_MAXK)
    { fk_set_error(ctx, "k = %d outside the supported range [8,%d]", K, SP_MAXK);
      return (FK_EUNSUPPORTED);
    }
  if (nbytes < K)
    return (FK_OK);
  const int64_t nstarts = nbytes - K + 1;
  const int64_t ntiles  = (nstarts + SP_TILE - 1) / SP_TILE;
  int sample = 32;
  while (sample > 1 && ntiles / sample < 64)
    sample >>= 1;
  SplitArgs a = SplitArgs();
  a.bases = (const unsigned char *) d_bases;
  a.nbytes = nbytes;
  a.kmer = K;
  a.smer_bytes = ctx->wid.smer_bytes;
  a.sww = ctx->wid.smer_stride / 4;
  a.nbuckets = nb;
  a.mbucket = ctx->d_mbucket;
  a.counts = ctx->d_scratch;
  a.cursor = ctx->d_scratch + 512; a.cstride = 1;
  a.limit = NULL; a.pos = NULL; a.skipb = 0x100u; a.ent = NULL;
  a.out = NULL; a.cap = 0;
  a.overflowed = (u32 *) (ctx->d_scratch + 1024);
  a.tile_stride = sample;
  { const int rcp = sp_packed_args(ctx, a, pk, ntiles);
    if (rcp != FK_OK) return (rcp);
  }
  FK_HIP(ctx, hipMemsetAsync(ctx->d_scratch, 0, 1032 * sizeof(u64), s));
  if (ctx->d_plan == NULL)
    FK_HIP(ctx, hipMalloc((void **) &ctx->d_plan, 32 * 256 * sizeof(u64)));
  FK_HIP(ctx, hipMemsetAsync(ctx->d_plan, 0, 32 * 256 * sizeof(u64), s));
  a.plan = ctx->d_plan;
  sp_launch<false, false>(a, ntiles / sample, s);
  FK_LAUNCH_CHECK(ctx);
  u64 *hp = ctx->h_scratch + 8192;                 // pinned, 32 x 256 words
  FK_HIP(ctx, hipMemcpyAsync(hp, ctx->d_plan, 32 * 256 * sizeof(u64), hipMemcpyDeviceToHost, s));
  FK_HIP(ctx, hipStreamSynchronize(s));
  for (int b = 0; b < nb; b++)
    { u64 t = 0;
      for (int c = 0; c < 32; c++)
        t += hp[c * 256 + b];
      ctx->h_scratch[b] = t;
    }
  const double scale = (double) ntiles / (double) (ntiles / sample);
  int64_t tot = 0;
  for (int b = 0; b < nb; b++)
    { offsets[b] = tot;
      tot += (int64_t) ((double) ctx->h_scratch[b] * scale * 1.05) + 8192 + (2 << (SP_LB + SP_LSTREAMS));   // + ragged stream ends
    }
  offsets[nb] = tot;
  *cap = tot;
  return (FK_OK);
}

// Emit into the planned regions; counts[b] receives what bucket b really holds.
// FK_ESTATE: some region was too small (very uneven input) -- use the exact two-call path.
// [b0, b1) a proper sub-range of the buckets: a GROUP PASS of a multi-pass split -- only the super-mers
// of these buckets are emitted (offsets[b0..b1] are their regions in d_out); *ninst still counts every
// valid k-mer of the input.  mode 0: the others are dropped (every pass recomputes all minimizers);
// mode 1: the others' 4-byte entries are recorded per tile (ctx->ent_cap entries of room; ctx->ent_valid
// tells whether they all fitted); mode 2: no minimizers at all -- the records of [b0, b1) are rebuilt from
// the entries a mode-1 pass over the same reads left behind (k_split_replay).
int fkx_split_planned(fk_ctx *ctx, const void *d_bases, int64_t nbytes, void *d_out, int64_t cap,
                      const int64_t *offsets, int64_t *counts, int64_t *ninst, int b0, int b1, int mode,
                      const fk_pkview *pk, uint8_t *d_dig)
{ hipStream_t s = ctx->stream;
  const int   K = ctx->prm.kmer;
  const int   nb = ctx->prm.nbuckets;
  if (b1 < 0) b1 = nb;
  const bool  group = (b0 > 0 || b1 < nb);
  for (int b = 0; b < nb; b++)
    counts[b] = 0;
  *ninst = 0;
  if (nbytes < K)
    return (FK_OK);
  if (b0 < 0 || b1 > nb || b0 >= b1 || (group && nb > 255) || (mode != 0 && !group))
    { fk_set_error(ctx, "planned split: bad bucket range [%d,%d) of %d (mode %d)", b0, b1, nb, mode);
      return (FK_EINVAL);
    }
  if (offsets[b1] > cap)
    { fk_set_error(ctx, "planned split needs %lld records, buffer holds %lld",
                   (long long) offsets[b1], (long long) cap);
      return (FK_EINVAL);
    }
  const int64_t nstarts = nbytes - K + 1;
  const int64_t ntiles  = (nstarts + SP_TILE - 1) / SP_TILE;
  u64 *h = ctx->h_scratch + 512;                     // pinned
  for (int b = 0; b < nb; b++)
    { const bool in = (b >= b0 && b < b1);
      h[b] = in ? (u64) offsets[b] : 0;              // region starts
      h[256 + b] = in ? (u64) offsets[b + 1] : 0;    // limits
    }
  FK_HIP(ctx, hipMemsetAsync(ctx->d_scratch, 0, 1032 * sizeof(u64), s));
  FK_HIP(ctx, hipMemcpyAsync(ctx->d_scratch + 512, h, 512 * sizeof(u64), hipMemcpyHostToDevice, s));
  if (sp_cursors(ctx) == NULL)
    return (FK_ENOMEM);
  if (mode != 2)
    FK_HIP(ctx, hipMemsetAsync(ctx->d_cursors, 0, ((size_t) nb << SP_LSTREAMS) * FK_CURSOR_STRIDE * sizeof(u64), s));
  if (group && mode == 0)
    { uint8_t *hm = ctx->h_mbucket_pass;                          // pinned (the previous pass has been waited for)
      for (int r = 0; r < FK_NRANKS; r++)
        { const int b = ctx->h_mbucket[r];
          hm[r] = (uint8_t) ((b >= b0 && b < b1) ? b : 0xFF);
        }
      FK_HIP(ctx, hipMemcpyAsync(ctx->d_mbucket_pass, hm, FK_NRANKS, hipMemcpyHostToDevice, s));
    }
  SplitArgs a = SplitArgs();
  a.bases = (const unsigned char *) d_bases;
  a.nbytes = nbytes;
  a.kmer = K;
  a.smer_bytes = ctx->wid.smer_bytes;
  a.sww = ctx->wid.smer_stride / 4;
  a.nbuckets = nb;
  a.mbucket = (group && mode == 0) ? ctx->d_mbucket_pass : ctx->d_mbucket;
  a.counts = ctx->d_scratch;
  a.cursor = ctx->d_cursors; a.cstride = FK_CURSOR_STRIDE; a.lstreams = SP_LSTREAMS;
  a.rbase = ctx->d_scratch + 512;
  a.limit = ctx->d_scratch + 768;
  a.pos = NULL;
  a.skipb = (group && mode == 0) ? 0xFFu : 0x100u;
  a.out = (u32 *) d_out; a.cap = cap;
  a.overflowed = (u32 *) (ctx->d_scratch + 1024);
  a.tile_stride = 1;
  a.ent = NULL; a.ent_cursor = ctx->d_cursors + ((size_t) 256 << SP_LSTREAMS) * FK_CURSOR_STRIDE; a.tile_ent = NULL; a.ent_cap = 0;
  a.gb0 = b0; a.gb1 = b1;
  a.dig = (ctx->wid.smer_stride == 20) ? d_dig : NULL;      // (a group pass writes the digits of the records it emits)
  a.dig2 = (a.dig != NULL && ctx->dig2_off > 0) ? a.dig + ctx->dig2_off : NULL;   // (fkx_dig_slot made the planes)
  { const int rcp = sp_packed_args(ctx, a, (mode == 2 && pk != NULL) ? NULL : pk, ntiles);   // (a replay pass needs no tile index)
    if (rcp != FK_OK) return (rcp);
  }
  const int64_t nchunks = (ntiles + SP_RCH - 1) / SP_RCH;
  a.tile_cnt = NULL; a.chunk_base = NULL;
  if (mode != 0)
    { a.ent = (u32 *) ctx->slot_ptr[FK_SLOT_ENT];
      a.tile_ent = (u64 *) ctx->slot_ptr[FK_SLOT_TENT];
      a.tile_cnt = (uint16_t *) ctx->slot_ptr[FK_SLOT_TCNT];
      a.chunk_base = (const u64 *) ctx->slot_ptr[FK_SLOT_CBASE];
      a.ent_cap = ctx->ent_cap;
      if (a.ent == NULL || a.tile_ent == NULL || a.tile_cnt == NULL || a.chunk_base == NULL
          || ctx->slot_cap[FK_SLOT_TENT] < ntiles * 8 || ctx->slot_cap[FK_SLOT_TCNT] < ntiles * nb * 2
          || ctx->slot_cap[FK_SLOT_CBASE] < nchunks * nb * 12 + FK_CBASE_EXTRA
          || (mode == 2 && (!ctx->ent_valid || ctx->ent_ntiles != ntiles)))
        { fk_set_error(ctx, "planned split: no entries to record into / replay from");
          return (FK_ESTATE);
        }
      if (mode == 2)
        for (int b = b0; b < b1; b++)
          if (ctx->ent_totals[b] > offsets[b + 1] - offsets[b])
            { fk_set_error(ctx, "planned split: a bucket region was too small");
              return (FK_ESTATE);
            }
    }
  if (mode == 1)
    { FK_HIP(ctx, hipMemsetAsync(a.ent_cursor, 0, 64 * FK_CURSOR_STRIDE * sizeof(u64), s));
      ctx->ent_valid = false;
    }
  if (mode == 2)
    { a.tile_end = ntiles;
      for (int64_t t = 0; t < ntiles; t += (int64_t) SP_MAXGRID * SP_RTW)
        { a.tile0 = t;
          const int64_t left = (ntiles - t + SP_RTW - 1) / SP_RTW;
          const int64_t nbk = (left < SP_MAXGRID) ? left : SP_MAXGRID;
#ifdef FK_ABLATION
          a.abl = getenv("FK_REPLAY_ABL") ? atoi(getenv("FK_REPLAY_ABL")) : 0;
#endif
          if (pk != NULL)
            hipLaunchKernelGGL(k_split_replay<true>, dim3((unsigned) nbk), dim3(SP_THREADS), 0, s, a);
          else
            hipLaunchKernelGGL(k_split_replay<false>, dim3((unsigned) nbk), dim3(SP_THREADS), 0, s, a);
        }
    }
  else
    sp_launch<true, false>(a, ntiles, s);
  FK_LAUNCH_CHECK(ctx);
  FK_HIP(ctx, hipMemcpyAsync(ctx->h_scratch, ctx->d_scratch, 1032 * sizeof(u64), hipMemcpyDeviceToHost, s));
  int64_t totals[256];
  if (mode != 2)
    { const int rcs = sp_finish_streams(ctx, a, b0, b1, totals);
      if (rcs != FK_OK)
        return (rcs);
    }
  else
    FK_HIP(ctx, hipStreamSynchronize(s));
  const u32 ovf = *(u32 *) (ctx->h_scratch + 1024);
  if (ovf == 1)
    { fk_set_error(ctx, "planned split: a bucket region was too small");
      return (FK_ESTATE);
    }
  if (mode == 1)
    { ctx->ent_valid = (ovf == 0);                   // 2: the entries did not fit -- the records emitted are fine
      ctx->ent_ntiles = ntiles;
      if (ctx->ent_valid)
        { // counts -> in-chunk prefixes + chunk bases + exact totals of every bucket
          u32 *ctot = (u32 *) ((u64 *) ctx->slot_ptr[FK_SLOT_CBASE] + (size_t) nchunks * nb);
          u64 *dtot = ctx->d_scratch + 2048;
          hipLaunchKernelGGL(k_split_chunk, dim3((unsigned) ((nchunks + 3) / 4)), dim3(256), 0, s,
                             (uint16_t *) ctx->slot_ptr[FK_SLOT_TCNT], ntiles, nb, ctot);
          { const int64_t npart = (int64_t) SP_CSP * (256 / nb);
            const int64_t len = (nchunks + npart - 1) / npart;
            u64 *part = (u64 *) ((char *) ctx->slot_ptr[FK_SLOT_CBASE] + (((size_t) nchunks * nb * 12 + 7) & ~(size_t) 7));
            hipLaunchKernelGGL(k_split_chunksum, dim3(SP_CSP), dim3(256), 0, s, (const u32 *) ctot, nchunks, nb, len, part);
            hipLaunchKernelGGL(k_split_chunkscan, dim3((unsigned) nb), dim3(256), 0, s, part, npart, nb, dtot);
            hipLaunchKernelGGL(k_split_chunkbase, dim3(SP_CSP), dim3(256), 0, s, (const u32 *) ctot, nchunks, nb, len,
                               (const u64 *) part, (u64 *) ctx->slot_ptr[FK_SLOT_CBASE]);
          }
          FK_LAUNCH_CHECK(ctx);
          FK_HIP(ctx, hipMemcpyAsync(ctx->h_scratch + 2048, dtot, (size_t) nb * 8, hipMemcpyDeviceToHost, s));
          FK_HIP(ctx, hipStreamSynchronize(s));
          for (int b = 0; b < nb; b++)
            ctx->ent_totals[b] = (int64_t) ctx->h_scratch[2048 + b];
        }
    }
  if (mode == 2)
    for (int b = b0; b < b1; b++)
      counts[b] = ctx->ent_totals[b];
  else
  for (int b = b0; b < b1; b++)
    counts[b] = totals[b - b0];
  int64_t t = 0;
  for (int x = 0; x < 64; x++)
    t += (int64_t) ctx->h_scratch[256 + x];
  *ninst = t;
  return (FK_OK);
}
#endif   // FK_HOST_EMU
