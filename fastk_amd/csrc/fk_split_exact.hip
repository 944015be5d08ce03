// fk_split_exact.hip -- the reference's OWN super-mer rule, a thread per stretch of a read.
//
// The position-parallel splitter (fk_split.hip) is free to cut super-mers differently because no
// count depends on the cuts.  One thing does: the first-byte boundaries of the hidden .ktab part
// files come from the first-byte census of bucket 0's weighted k-mer list (Table_Split,
// count.c:1560-1565; msd_sort's thread ranges, MSDsort.c:330-352), i.e. from the exact multiset of
// DISTINCT super-mers.  With fk_params.exact_parts = 1 the pipeline therefore replays
// Distribute_Block itself (split.c:1016-1393): rolling canonical 5-mer under the frequency-ranked
// base order (Tran/Cran, split.c:529-575, 630-639), strict < on arrival, <= on the forced rescan
// after MAX_SUPER k-mers, the clipping around non-ACGT bases (split.c:1167-1232) and the
// end-of-read flush (split.c:1342-1347).  With the scheme of fk_scheme.hip (padded minimizers, prefix trie, leaves
// dealt to NPARTS buckets) also for inputs the reference cuts into several buckets: the super-mers come out grouped by
// bucket.  For blocks cut like io.c cuts them.
//
// The rule is sequential along a read; what runs in parallel (round 5):
//   * segments -- a read is cut at positions where the reference's state does not depend on what came before
//     (k_xs_find: a minimizer value strictly below the MAX_SUPER values in front of it, no non-acgt base within 2K), one
//     candidate per XS_BLOCK positions; a thread of k_split_exact follows one segment (a whole read when k > 64);
//   * the walk itself keeps the chain of minimizers behind the current one in registers, so a forced closing is a register
//     move and not a walk over the ring of the last 2K values (the ring is still written, for the rare closing that finds
//     the chain run out), and the lanes of a wave write the same ring row in the same turn;
//   * the emit pass leaves an 8-byte note per super-mer in its record slot; k_xs_pack, one thread per record, turns the
//     notes into records 16 bases at a time.
// Two walks (count per segment and bucket, scan, emit) as before.  1 G bases of 15 kbp reads: 516 ms at the start of round 5,
// 18 ms now (kernels: two walks 9.5 ms, k_xs_find 2.2 ms, k_xs_pack 1 ms); the default splitter's stage is 10 ms.
#include "fk_common.h"

#define XS_THREADS 128
#define XS_RING    256          // >= 2 * nextpow2(K) for K <= 128 (MOD_LEN, FastK.c:446-450)
#define XS_MAXPARTS FK_EXACT_MAXPARTS
#define XS_DQ      6            // entries of the minimizer chain kept in registers (k_split_exact)

struct ExactArgs
{ const unsigned char *bases;
  const int64_t *roff;          // [nreads+1] read r = bases[roff[r] .. roff[r+1]-1), then a 0 byte
  int64_t   nreads;
  int       kmer;
  int       bc_prefix;
  int       tran[4];            // rank of a,c,g,t
  int       smer_bytes;
  int       sww;
  u32      *cnt;                // [nparts][nreads] super-mers of each read per bucket (count pass)
  const u64 *off;               // [nparts][nreads] first record of each read in each bucket (emit pass)
  u32      *out;
  u64      *inst;               // [64] valid k-mer instances, spread
  int       pad_len;            // minimizer length MIN_LEN + PAD (5 without a scheme)
  int       pad2;               // 2 * PAD
  const int *trie;              // Min_Part: < 0 children at -trie[x] + base, else bucket; NULL: one bucket
  int       nparts;
  // segments (round 5): the unit of a thread is a stretch of a read that begins where the reference's state is known
  // whatever came before -- at the read's start, or at a position whose minimizer value is strictly below every value of
  // the MAX_SUPER positions in front of it (then `mp < mc` holds whatever mc is, the super-mer in progress is closed
  // there and the next one starts with m = p, mc = mp, last = p) with no non-acgt base within 2K positions in front of it
  // (then nothing of the N bookkeeping reaches across).  k_xs_find picks at most one such position per XS_BLOCK positions.
  int64_t   nseg;               // threads of the split kernels (= nreads when seg_read is NULL: one segment per read)
  const u32 *seg_read;          // [nseg] read of segment i
  const u32 *seg_p0;            // [nseg] its first position in the read (0: the read's start)
  const u32 *seg_p1;            // [nseg] the position it ends WITH (the next segment's first), 0xffffffff: the read's end
  int       dq_cap;             // entries of the minimizer chain k_split_exact keeps (XS_DQ; fk_debug_set("exact_chain") less)
  int       noflip;             // exact_parts = 2 (a run with profiles): the records keep the read's strand, as the reference's
                                // do under -p (split.c:1245: Stuff_Seq(..., 0, ...)) -- a super-mer and its reverse complement
                                // are then two records, and the first-byte census that cuts the .ktab parts sees both
  int       defer;              // the emit pass leaves notes, k_xs_pack writes the records (records of >= 2 words)
};

__device__ __forceinline__ int xs_code(unsigned ch)
{ const unsigned u = ch & 0xDFu;
  const bool ok = (u == 0x41u) | (u == 0x43u) | (u == 0x47u) | (u == 0x54u);
  const unsigned x = (ch >> 1) & 3u;
  return ok ? (int) (x ^ (x >> 1)) : 4;
}

// ONE: a single bucket (no scheme) -- the per-bucket record counters are one register, there is no trie to walk
template <bool EMIT, bool ONE>
__global__ __launch_bounds__(XS_THREADS) void k_split_exact(ExactArgs a)
{ const int64_t sg = (int64_t) blockIdx.x * XS_THREADS + threadIdx.x;      // the segment: cnt / off are indexed by it
  if (sg >= a.nseg)
    return;
  const int64_t r  = (a.seg_read != NULL) ? (int64_t) a.seg_read[sg] : sg;
  const int     p0 = (a.seg_read != NULL) ? (int) a.seg_p0[sg] : 0;
  const u32     p1 = (a.seg_read != NULL) ? a.seg_p1[sg] : 0xffffffffu;
  const int K   = a.kmer;
  const int KM1 = K - 1;
  const int PL1 = a.pad_len - 1;
  const int MS  = K - PL1;                                // MAX_SUPER, split.c:628
  const u32 vmsk = (1u << (2 * a.pad_len)) - 1u;          // PAD_MSK (pad_len <= 15)
  const unsigned char *s = a.bases + a.roff[r] + a.bc_prefix;
  const int q = (int) (a.roff[r + 1] - a.roff[r]) - 1 - a.bc_prefix;     // split.c:1077-1079
  if (q < K)
    { if (!EMIT)
        for (int b = 0; b < a.nparts; b++)
          a.cnt[(int64_t) b * a.nseg + sg] = 0;
      return;
    }
  int rmsk = 1;
  while (rmsk < K) rmsk <<= 1;
  rmsk = 2 * rmsk - 1;

  u32 ring[XS_RING];                      // (min(c,u) << 1) | (u < c) of position j at (j - pofs) & rmsk
  // pofs: every lane of a wave is at the same ring index in the same turn of the main loop (its first turn is index K for
  // a read's start and for a segment inside a read alike), so that the scratch store of a turn is one row of the wave's
  // ring and not 64 rows -- indexed by the position itself it was a cache line per lane and turn
  const int pofs = (p0 > 0) ? p0 + 1 - K : 0;
  const int t0 = a.tran[0], t1 = a.tran[1], t2 = a.tran[2], t3 = a.tran[3];
  auto fwv = [&](int code) -> unsigned { return (unsigned) (code == 1 ? t1 : code == 2 ? t2 : code == 3 ? t3 : t0); };
  auto rcv = [&](int code) -> unsigned { return (unsigned) (code == 1 ? t2 : code == 2 ? t1 : code == 3 ? t0 : t3) << (2 * PL1); };

  // the bases of the scan come 16 at a time (an aligned uint4 that stays in registers until the scan leaves it): a byte
  // load per position was a trip to the cache per position of every lane
  uint4 ch = make_uint4(0u, 0u, 0u, 0u);
  uintptr_t ch_at = ~(uintptr_t) 0;
  auto base_at = [&](int pp) -> unsigned
    { const uintptr_t ad = (uintptr_t) (s + pp);
      if ((ad >> 4) != ch_at)
        { ch = *(const uint4 *) (ad & ~(uintptr_t) 15);
          ch_at = ad >> 4;
        }
      const unsigned o = (unsigned) (ad & 15);
      const u32 w = (o < 8) ? (o < 4 ? ch.x : ch.y) : (o < 12 ? ch.z : ch.w);
      return ((w >> (8 * (o & 3))) & 0xffu);
    };
  u32 nrec[ONE ? 1 : XS_MAXPARTS];        // records of this read so far, per bucket
  for (int b = 0; b < (ONE ? 1 : a.nparts); b++)
    nrec[b] = 0;
  const u64 off0 = (EMIT && ONE) ? a.off[sg] : 0ull;
  u64 ninst = 0;
  const int lenw  = a.smer_bytes >> 2;
  const int lensh = 24 - 8 * (a.smer_bytes & 3);

  auto emit = [&](int first_end, int n, int flip, u32 mval)
    { // k-mers ending at first_end .. first_end+n-1: bases s[first_end-KM1 .. first_end+n-1]; mval: their minimizer
      int bk = 0;
      if (!ONE && a.trie != NULL)                            // split.c:1149-1157
        { int o = (int) (mval >> a.pad2);
          bk = a.trie[o];
          int y = a.pad2 - 2;
          while (bk < 0)
            { o = (int) ((mval >> y) & 3u) - bk;
              bk = a.trie[o];
              y -= 2;
            }
        }
      if (EMIT && a.defer)
        { // where the super-mer lies and how long it is, left in its own record slot: k_xs_pack turns the note into the
          // record with every lane at work -- packing here held the whole wave at each closing of any of its lanes
          u32 *dst = a.out + ((ONE ? off0 : a.off[(int64_t) bk * a.nseg + sg]) + nrec[ONE ? 0 : bk]) * (u64) a.sww;
          const u64 at = (u64) (s - a.bases) + (u64) (first_end - KM1);
          dst[0] = (u32) at;
          dst[1] = (u32) (at >> 32) | ((u32) (n - 1) << 16) | ((u32) flip << 31);
        }
      else if (EMIT)
        { const unsigned char *b = s + (first_end - KM1);
          const int L = n - 1 + K;
          u32 *dst = a.out + ((ONE ? off0 : a.off[(int64_t) bk * a.nseg + sg]) + nrec[ONE ? 0 : bk]) * (u64) a.sww;
          for (int w = 0; w < a.sww; w++)
            { u32 x = 0;
              for (int j = 0; j < 16; j++)
                { const int i = 16 * w + j;
                  if (i < L)
                    { const int c = flip ? 3 - xs_code(b[L - 1 - i]) : xs_code(b[i]);
                      x |= ((u32) (c & 3)) << (30 - 2 * j);
                    }
                }
              if (w == lenw)
                x |= ((u32) (n - 1)) << lensh;
              dst[w] = __builtin_bswap32(x);
            }
        }
      nrec[ONE ? 0 : bk] += 1;
      ninst += (u64) n;
    };

  // The forced rescan (split.c:1304-1320) wants the rightmost minimum of the positions behind m.  Walking the ring for it
  // held the whole wave for MAX_SUPER scratch loads at almost every turn (some lane of 64 is always forcing), so the
  // answer is kept ready instead: dq = the first XS_DQ entries of the chain "rightmost minimum of (m, p], rightmost
  // minimum of what lies behind that one, ..." (values strictly increasing), in registers, statically indexed.  A new
  // position drops the entries it is <= of and goes to the end; `inc` says that entries beyond the ones kept exist and
  // are not known (all larger than the last one kept).  Only when a forced closing finds nothing kept does it walk the
  // ring as before.
  u32  dv[XS_DQ];                          // (value << 1) | flip, as in the ring
  int  dp[XS_DQ];                          // position
  int  cnt = 0;
  bool inc = false;
  unsigned mfl = 0;                        // orientation bit of the current minimizer m
#pragma unroll
  for (int i = 0; i < XS_DQ; i++)
    { dv[i] = 0; dp[i] = 0; }
  auto dq_push = [&](int pos, u32 e)
    { const unsigned v = e >> 1;
      int keep = 0;
#pragma unroll
      for (int i = 0; i < XS_DQ; i++)
        keep += (i < cnt && (dv[i] >> 1) < v) ? 1 : 0;
      if (keep < cnt)
        inc = false;                       // (everything behind a dropped entry is larger than it: dropped as well)
      cnt = keep;
      if (!inc)
        { if (cnt < a.dq_cap)
            {
#pragma unroll
              for (int i = 0; i < XS_DQ; i++)
                if (i == cnt)
                  { dv[i] = e; dp[i] = pos; }
              cnt += 1;
            }
          else
            inc = true;
        }
    };

  unsigned c = 0, u = 0, mp = 0, mc = vmsk + 1u;
  int m = 0, p;
  int ilo = -1, ihi = -1, phi = -1;
  int  last = KM1;
  if (p0 > 0)
    { // a segment that begins at a position where the reference's state does not depend on what came before: the
      // rolling codes of the pad_len bases that end there (all acgt), m = p0, mc = its value, last = p0, no N pending
      for (p = p0 - PL1; p <= p0; p++)
        { const int code = xs_code(base_at(p));
          c = ((c << 2) | fwv(code)) & vmsk;
          u = (u >> 2) | rcv(code);
        }
      const unsigned fl = (u < c);
      mp = fl ? u : c;
      ring[(p0 - pofs) & rmsk] = (mp << 1) | fl;
      m = p0; mc = mp; mfl = fl; last = p0;
    }
  else
  for (p = 0; p < K; p++)                                   // split.c:1096-1134
    { const int code = xs_code(base_at(p));
      c = ((c << 2) | fwv(code)) & vmsk;
      u = (u >> 2) | rcv(code);
      if (p >= PL1)
        { const unsigned fl = (u < c);
          mp = fl ? u : c;
          ring[(p - pofs) & rmsk] = (mp << 1) | fl;
          if (mp < mc)
            { m = p; mc = mp; mfl = fl; cnt = 0; inc = false; }
          else
            dq_push(p, (mp << 1) | fl);
        }
      if (code >= 4)
        { if (p > ihi)
            ilo = KM1;
          ihi = p + K;
        }
    }

  bool done = false;
  for (p = (p0 > 0 ? p0 + 1 : K); !done; p++)               // split.c:1136-1347
    { int  code = 0;
      bool closing, force;
      unsigned fl = 0;
      if (p < q)
        { code = xs_code(base_at(p));
          c = ((c << 2) | fwv(code)) & vmsk;
          u = (u >> 2) | rcv(code);
          fl = (u < c);
          mp = fl ? u : c;
          ring[(p - pofs) & rmsk] = (mp << 1) | fl;
          force   = (p - m >= MS);
          closing = force || (mp < mc);
        }
      else                                                   // end-of-read flush, split.c:1342-1347
        { if (ihi == q)
            break;
          mp = mc;
          force = closing = true;
          done = true;
        }
      if (closing)
        { int n;
          if (ihi >= last)                                   // split.c:1167-1232
            { if (ihi <= p)
                { last = ihi; ihi = -1; n = p - last; }
              else
                { if (phi > last)
                    { last = phi; phi = -1; }
                  n = ilo - last;
                }
            }
          else
            n = p - last;
          if (n > 0)
            emit(last, n, a.noflip ? 0 : (int) mfl, mc);
          if (done || (u32) p == p1)                         // (the next segment begins here, with the state this closing leaves)
            break;
          last = p;
        }
      // the minimizer after position p (split.c:1304-1320: `<=` on the forced rescan, the rightmost minimum behind m)
      if (mp < mc)                                           // (forced or not: p is below everything in the window)
        { m = p; mc = mp; mfl = fl; cnt = 0; inc = false; }
      else
        { dq_push(p, (mp << 1) | fl);
          if (force)
            { if (cnt > 0)
                { m = dp[0]; mc = dv[0] >> 1; mfl = dv[0] & 1u;
#pragma unroll
                  for (int i = 0; i + 1 < XS_DQ; i++)
                    { dv[i] = dv[i + 1]; dp[i] = dp[i + 1]; }
                  cnt -= 1;
                }
              else                                           // (inc: nothing kept of what lies behind m)
                { m += 1;
                  u32 e = ring[(m - pofs) & rmsk];
                  for (int j = m + 1; j <= p; j++)
                    { const u32 x = ring[(j - pofs) & rmsk];
                      if ((x >> 1) <= (e >> 1))
                        { m = j; e = x; }
                    }
                  mc = e >> 1; mfl = e & 1u;
                  inc = (m < p);
                }
            }
        }
      if (code >= 4)                                         // split.c:1323-1330
        { if (p > ihi)
            { phi = ihi; ilo = p; }
          ihi = p + K;
        }
    }

  if (!EMIT)
    for (int b = 0; b < (ONE ? 1 : a.nparts); b++)
      a.cnt[(int64_t) b * a.nseg + sg] = nrec[b];
  if (ninst != 0)
    atomicAdd(&a.inst[sg & 63], ninst);
}

__device__ __forceinline__ u32 xs_code16(const uint4 x)
{ // 16 ASCII bases -> 16 two-bit codes, the first base in the top bits (a 0, c 1, g 2, t 3; either case).  Per word of 4:
  // bits 2:1 of each byte with g and t exchanged, then the four fields gathered into the top byte by one multiply
  // (fields at bit 0, 8, 16, 24 times 2^30 + 2^20 + 2^10 + 1 land at 30, 28, 26, 24; no two partial products overlap)
  auto four = [](u32 w) -> u32
    { const u32 t = (w >> 1) & 0x03030303u;
      const u32 c = t ^ ((t >> 1) & 0x01010101u);
      return ((c * 0x40100401u) >> 24);
    };
  return ((four(x.x) << 24) | (four(x.y) << 16) | (four(x.z) << 8) | four(x.w));
}

// one thread per record slot: the note k_split_exact<true> left there (first base, k-mers, orientation) becomes the
// record -- bases 2-bit packed from the high end of big-endian words, reverse-complemented when the minimizer was taken
// on the other strand, the count of k-mers less one in the byte after them (split.c:1234-1301)
__global__ __launch_bounds__(256) void k_xs_pack(ExactArgs a, int64_t ns)
{ const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (i >= ns) return;
  u32 *dst = a.out + i * (int64_t) a.sww;
  const u32 d0 = dst[0], d1 = dst[1];
  const unsigned char *b = a.bases + ((((u64) (d1 & 0xffffu)) << 32) | d0);
  const int n = (int) ((d1 >> 16) & 0x7fffu) + 1;
  const bool flip = (d1 >> 31) != 0;
  const int L = n - 1 + a.kmer;
  const int lenw  = a.smer_bytes >> 2;
  const int lensh = 24 - 8 * (a.smer_bytes & 3);
  // the bases as a stream of aligned 16-byte pieces: forward from the piece that holds b[0], or -- the minimizer on the other
  // strand -- backward from the piece that holds b[L-1], each piece reversed and complemented; the record's words are that
  // stream shifted by the bases of the first piece that lie outside the super-mer
  const uintptr_t a0 = (uintptr_t) b, a1 = (uintptr_t) (b + (L - 1));
  const int64_t c0 = (int64_t) (a0 >> 4), c1 = (int64_t) (a1 >> 4);
  const int sh = 2 * (int) (flip ? 15 - (a1 & 15) : (a0 & 15));
  auto piece = [&](int k) -> u32
    { const int64_t c = flip ? c1 - k : c0 + k;
      if (c < c0 || c > c1)
        return (0u);
      const u32 x = xs_code16(*(const uint4 *) ((uintptr_t) c << 4));
      if (!flip)
        return (x);
      const u32 r = __brev(x);                                         // (order reversed, the two bits of a base too)
      return (~(((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1)));
    };
  u32 cur = piece(0);
  for (int w = 0; w < a.sww; w++)
    { const u32 nxt = piece(w + 1);
      u32 x = (sh == 0) ? cur : ((cur << sh) | (nxt >> (32 - sh)));
      const int valid = L - 16 * w;
      if (valid < 16)
        x = (valid <= 0) ? 0u : (x & ~(0xffffffffu >> (2 * valid)));
      if (w == lenw)
        x |= ((u32) (n - 1)) << lensh;
      dst[w] = __builtin_bswap32(x);
      cur = nxt;
    }
}

// ---- segment starts ------------------------------------------------------------------------------------------------
// A read of q positions has max(1, ceil(q / XS_BLOCK)) blocks; block 0 begins the read's first segment, every later block
// looks for ONE position p inside itself where a segment may begin (see ExactArgs): the minimizer value of p -- computed
// exactly as k_split_exact computes it -- strictly below the values of the MAX_SUPER positions in front of it, and no
// non-acgt base among the 2K positions up to p.  A block without such a position (low complexity, N runs) begins none: its
// positions belong to the segment in front of it.  Any choice is right; more of them is only more threads.
#define XS_BLOCK 1024

__global__ __launch_bounds__(256) void k_xs_blocks(const int64_t *__restrict__ roff, int64_t nreads, int bc_prefix, int kmer,
                                                   u32 *__restrict__ nblk)
{ const int64_t r = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (r >= nreads) return;
  const int64_t q = (roff[r + 1] - roff[r]) - 1 - bc_prefix;
  nblk[r] = (q < kmer || q <= XS_BLOCK) ? 1u : (u32) ((q + XS_BLOCK - 1) / XS_BLOCK);
}

// one thread per read: the read of each of its blocks
__global__ __launch_bounds__(256) void k_xs_blkread(const u32 *__restrict__ nblk, const u64 *__restrict__ boff, int64_t nreads,
                                                    u32 *__restrict__ blk_read, u32 *__restrict__ blk_j)
{ const int64_t r = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (r >= nreads) return;
  const u64 o = boff[r];
  for (u32 j = 0; j < nblk[r]; j++)
    { blk_read[o + j] = (u32) r;
      blk_j[o + j] = j;
    }
}

// one thread per block: where its segment begins (0xffffffff: nowhere), and 1 / 0 into flag
__global__ __launch_bounds__(128) void k_xs_find(ExactArgs a, const u32 *__restrict__ blk_read, const u32 *__restrict__ blk_j,
                                                 int64_t nblocks, u32 *__restrict__ blk_p0, u32 *__restrict__ flag)
{ const int64_t bi = (int64_t) blockIdx.x * 128 + threadIdx.x;
  if (bi >= nblocks) return;
  const int64_t r = blk_read[bi];
  const u32 j = blk_j[bi];
  if (j == 0)
    { blk_p0[bi] = 0; flag[bi] = 1; return; }
  const int K = a.kmer, PL1 = a.pad_len - 1, MS = K - PL1;
  const u32 vmsk = (1u << (2 * a.pad_len)) - 1u;
  const unsigned char *s = a.bases + a.roff[r] + a.bc_prefix;
  const int q = (int) (a.roff[r + 1] - a.roff[r]) - 1 - a.bc_prefix;
  const int t0 = a.tran[0], t1 = a.tran[1], t2 = a.tran[2], t3 = a.tran[3];
  auto fwv = [&](int code) -> unsigned { return (unsigned) (code == 1 ? t1 : code == 2 ? t2 : code == 3 ? t3 : t0); };
  auto rcv = [&](int code) -> unsigned { return (unsigned) (code == 1 ? t2 : code == 2 ? t1 : code == 3 ? t0 : t3) << (2 * PL1); };
  const int lo = (int) j * XS_BLOCK;                          // candidates lo <= p < hi (p < q: the loop region of the read)
  const int hi = (lo + XS_BLOCK < q) ? lo + XS_BLOCK : q;
  u32 win[64];                                               // the values of the last 64 positions (MS <= 60 for k <= 64)
  unsigned c = 0, u = 0;
  int bad = -1;                                              // the last position that holds no acgt
  u32 found = 0xffffffffu;
  const int start = lo - 2 * K - PL1;                        // (lo >= XS_BLOCK = 1024 > 2K + pad for k <= 64: inside the read)
  uint4 ch = make_uint4(0u, 0u, 0u, 0u);
  uintptr_t ch_at = ~(uintptr_t) 0;
  for (int p = start; p < hi; p++)
    { const uintptr_t ad = (uintptr_t) (s + p);              // (16 bases a load, as in k_split_exact)
      if ((ad >> 4) != ch_at)
        { ch = *(const uint4 *) (ad & ~(uintptr_t) 15);
          ch_at = ad >> 4;
        }
      const unsigned o = (unsigned) (ad & 15);
      const u32 cw = (o < 8) ? (o < 4 ? ch.x : ch.y) : (o < 12 ? ch.z : ch.w);
      const int code = xs_code((cw >> (8 * (o & 3))) & 0xffu);
      c = ((c << 2) | fwv(code)) & vmsk;
      u = (u >> 2) | rcv(code);
      if (code >= 4) bad = p;
      const u32 v = (u < c) ? u : c;
      if (p >= lo && p - bad > 2 * K)
        { bool ok = true;
          for (int d = 1; d <= MS; d++)
            if (win[(p - d) & 63] <= v) { ok = false; break; }
          if (ok) { found = (u32) p; break; }
        }
      win[p & 63] = v;
    }
  blk_p0[bi] = found;
  flag[bi] = (found != 0xffffffffu) ? 1u : 0u;
}

// the blocks that begin a segment, in order: segment i = (read, first position), and where it ends
__global__ __launch_bounds__(256) void k_xs_segs(const u32 *__restrict__ blk_read, const u32 *__restrict__ blk_p0,
                                                 const u32 *__restrict__ flag, const u64 *__restrict__ soff, int64_t nblocks,
                                                 u32 *__restrict__ seg_read, u32 *__restrict__ seg_p0)
{ const int64_t bi = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (bi >= nblocks || flag[bi] == 0) return;
  seg_read[soff[bi]] = blk_read[bi];
  seg_p0[soff[bi]] = blk_p0[bi];
}

__global__ __launch_bounds__(256) void k_xs_ends(const u32 *__restrict__ seg_read, const u32 *__restrict__ seg_p0, int64_t nseg,
                                                 u32 *__restrict__ seg_p1)
{ const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (i >= nseg) return;
  seg_p1[i] = (i + 1 < nseg && seg_read[i + 1] == seg_read[i]) ? seg_p0[i + 1] : 0xffffffffu;
}

// byte histogram of bases[lo,hi)                                       frequency_thread, split.c:95-112
__global__ __launch_bounds__(256) void k_base_freq(const unsigned char *__restrict__ bases, int64_t lo,
                                                   int64_t hi, u64 *__restrict__ out)
{ __shared__ u32 h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  for (int64_t i = lo + (int64_t) blockIdx.x * 256 + threadIdx.x; i < hi; i += (int64_t) gridDim.x * 256)
    atomicAdd(&h[bases[i]], 1u);
  __syncthreads();
  if (h[threadIdx.x] != 0)
    atomicAdd(&out[threadIdx.x], (u64) h[threadIdx.x]);
}

#if !defined(FK_HOST_EMU) || defined(FK_EMU_FULL)      // (host halves stay out of the CPU tests' build of the kernels, tests/csrc/hip_emu.h)
// Tran ranking as Determine_Scheme computes it (split.c:529-575), including its quirk: the per-thread
// byte counts are summed INTO thread 0's own vector from j = 0 (split.c:536-539), so read stripe 0 of
// the training block counts twice.  train_reads = reads of the first block (Get_First_Block).
int fkx_train_tran(fk_ctx *ctx, const void *d_bases, const int64_t *h_roff, int64_t train_reads,
                   int nthreads, int *tran)
{ hipStream_t s = ctx->stream;
  u64 *d_f = ctx->d_scratch + 2048;
  for (int i = 0; i < 4; i++)
    tran[i] = i;
  if (train_reads <= 0)
    return (FK_OK);
  FK_HIP(ctx, hipMemsetAsync(d_f, 0, 256 * sizeof(u64), s));
  const int64_t lo = h_roff[0], hi = h_roff[train_reads];
  const int64_t stripe0 = (nthreads > 1) ? train_reads / nthreads : train_reads;
  hipLaunchKernelGGL(k_base_freq, dim3(2048), dim3(256), 0, s, (const unsigned char *) d_bases, lo, hi, d_f);
  hipLaunchKernelGGL(k_base_freq, dim3(2048), dim3(256), 0, s, (const unsigned char *) d_bases, lo,
                     h_roff[stripe0], d_f);
  FK_LAUNCH_CHECK(ctx);
  FK_HIP(ctx, hipMemcpyAsync(ctx->h_scratch, d_f, 256 * sizeof(u64), hipMemcpyDeviceToHost, s));
  FK_HIP(ctx, hipStreamSynchronize(s));
  const u64 *f = ctx->h_scratch;
  const u64 f4[4] = { f['a'] + f['A'], f['c'] + f['C'], f['g'] + f['G'], f['t'] + f['T'] };
  for (int a = 0; a < 4; a++)
    { int rank = 0;
      for (int b = 0; b < 4; b++)
        if (f4[b] < f4[a] || (f4[b] == f4[a] && b < a))
          rank += 1;
      tran[a] = rank;
    }
  return (FK_OK);
}
#endif   // FK_HOST_EMU

// ---- exclusive scan of many counts -------------------------------------------------------------------------------------
// k_exscan_tiles is one workgroup (made for a few thousand tile counts); the exact splitter scans a count per segment and
// bucket -- millions.  Tiles of 4096 counts: their sums, the scan of the sums by k_exscan_tiles, the scan inside each tile.
// (a tile's sum is kept in 32 bits; one that does not fit -- 4096 consecutive counts of a million each: unsegmented,
// very long low-complexity reads -- raises *ovf, which every caller reads back with the total)
__global__ __launch_bounds__(256) void k_xs_tilesum(const u32 *__restrict__ in, int64_t n, u32 *__restrict__ tsum,
                                                    u64 *__restrict__ ovf)
{ __shared__ u64 tmp[8];
  const int64_t i0 = (int64_t) blockIdx.x * 4096 + (int64_t) threadIdx.x * 16;
  u64 mine = 0;
  for (int k = 0; k < 16; k++)
    mine += (i0 + k < n) ? in[i0 + k] : 0u;
  u64 tot;
  fk_block_exscan_256<u64>(mine, tmp, &tot);
  if (threadIdx.x == 0)
    { tsum[blockIdx.x] = (u32) tot;
      if ((tot >> 32) != 0)
        atomicMax((unsigned long long *) ovf, 1ull);
    }
}

__global__ __launch_bounds__(256) void k_xs_tilescan(const u32 *__restrict__ in, int64_t n, const u64 *__restrict__ toff,
                                                     u64 *__restrict__ out)
{ __shared__ u64 tmp[8];
  const int64_t i0 = (int64_t) blockIdx.x * 4096 + (int64_t) threadIdx.x * 16;
  u32 v[16];
  u64 mine = 0;
#pragma unroll
  for (int k = 0; k < 16; k++)
    { v[k] = (i0 + k < n) ? in[i0 + k] : 0u;
      mine += v[k];
    }
  u64 tot;
  u64 run = toff[blockIdx.x] + fk_block_exscan_256<u64>(mine, tmp, &tot);
#pragma unroll
  for (int k = 0; k < 16; k++)
    { if (i0 + k < n)
        out[i0 + k] = run;
      run += v[k];
    }
}

// out[i] = in[0] + .. + in[i-1] for i < n, *total = the sum (a tile's sum must fit 32 bits: counts of records)
#if !defined(FK_HOST_EMU) || defined(FK_EMU_FULL)
#define XS_OVF_WORD 2210          // d_scratch / h_scratch word: "a tile sum of xs_exscan did not fit 32 bits"
static int xs_overflowed(fk_ctx *ctx)
{ if (ctx->h_scratch[XS_OVF_WORD] == 0)
    return (FK_OK);
  fk_set_error(ctx, "exact split: more than 2^32 records in 4096 consecutive segments");
  return (FK_EUNSUPPORTED);
}

static int xs_exscan(fk_ctx *ctx, const u32 *in, int64_t n, u64 *out, u64 *total)
{ hipStream_t s = ctx->stream;
  if (n <= 32768)
    { hipLaunchKernelGGL(k_exscan_tiles, dim3(1), dim3(256), 0, s, in, n, out, total);
      return (FK_OK);
    }
  const int64_t nt = (n + 4095) / 4096;
  const int64_t t4 = (nt * 4 + 63) & ~63ll;
  char *tb = (char *) fk_slot(ctx, FK_SLOT_XS_SCAN, t4 + nt * 8 + 64);
  if (tb == NULL) return (FK_ENOMEM);
  u32 *tsum = (u32 *) tb;
  u64 *toff = (u64 *) (tb + t4);
  hipLaunchKernelGGL(k_xs_tilesum, dim3((unsigned) nt), dim3(256), 0, s, in, n, tsum, ctx->d_scratch + XS_OVF_WORD);
  hipLaunchKernelGGL(k_exscan_tiles, dim3(1), dim3(256), 0, s, (const u32 *) tsum, nt, toff, total);
  hipLaunchKernelGGL(k_xs_tilescan, dim3((unsigned) nt), dim3(256), 0, s, in, n, (const u64 *) toff, out);
  return (FK_OK);
}

/* bucket_counts / bucket_offs (records): the super-mers come out grouped by bucket when the context holds a scheme
   (ctx->scheme_nparts > 1, fkx_train_scheme); one bucket otherwise. */
int fkx_split_exact(fk_ctx *ctx, const void *d_bases, const int64_t *d_roff, int64_t nreads,
                    const int *tran, void **d_out, int64_t *nsuper, int64_t *ninst, int64_t *bucket_counts,
                    int64_t *bucket_offs)
{ hipStream_t s = ctx->stream;
  *nsuper = 0; *ninst = 0; *d_out = NULL;
  const int nparts = (ctx->scheme_nparts > 1) ? ctx->scheme_nparts : 1;
  for (int b = 0; b < nparts; b++)
    bucket_counts[b] = bucket_offs[b] = 0;
  if (nreads == 0)
    return (FK_OK);
  if (ctx->prm.kmer > 128 || ctx->prm.kmer < 8 || nparts > XS_MAXPARTS)
    return (FK_EUNSUPPORTED);
  FK_HIP(ctx, hipMemsetAsync(ctx->d_scratch + XS_OVF_WORD, 0, 8, s));
  u32 *d_cnt = NULL;
  u64 *d_off = NULL;
  u64 *d_inst = ctx->d_scratch + 2048;          // [64] instances, [64] = total records
  ExactArgs a;
  a.bases = (const unsigned char *) d_bases;
  a.roff = d_roff;
  a.nreads = nreads;
  a.kmer = ctx->prm.kmer;
  a.bc_prefix = ctx->prm.bc_prefix;              // exact mode keeps the prefix bytes and skips them here
  for (int i = 0; i < 4; i++)
    a.tran[i] = tran[i];
  a.smer_bytes = ctx->wid.smer_bytes;
  a.sww = ctx->wid.smer_stride / 4;
  a.cnt = d_cnt;
  a.off = d_off;
  a.out = NULL;
  a.inst = d_inst;
  a.nparts = nparts;
  a.defer = (a.sww >= 2) ? 1 : 0;
  a.noflip = (ctx->prm.exact_parts == 2) ? 1 : 0;
  a.dq_cap = (ctx->dbg_exact_chain >= 1 && ctx->dbg_exact_chain < XS_DQ) ? ctx->dbg_exact_chain : XS_DQ;
  a.pad_len = 5 + ((nparts > 1) ? ctx->scheme_pad : 0);
  a.pad2 = (nparts > 1) ? 2 * ctx->scheme_pad : 0;
  a.trie = (nparts > 1) ? ctx->d_min_part : NULL;
  // ---- segments: long reads are cut where the reference's state is known (ExactArgs; fk_debug_set("exact_segments", 0)
  //      keeps one thread per read)
  int64_t nseg = nreads;
  a.nseg = nreads; a.seg_read = a.seg_p0 = a.seg_p1 = NULL;
  if (ctx->prm.kmer <= 64 && ctx->dbg_exact_segments >= 0 && nreads < 0x7fffffffll)
    { const unsigned gr = (unsigned) ((nreads + 255) / 256);
      char *rb = (char *) fk_slot(ctx, FK_SLOT_XS_READS, nreads * 4 + (nreads + 1) * 8 + 256);
      if (rb == NULL) return (FK_ENOMEM);
      u32 *d_nblk = (u32 *) rb;
      u64 *d_boff = (u64 *) (rb + ((nreads * 4 + 63) & ~63ll));
      hipLaunchKernelGGL(k_xs_blocks, dim3(gr), dim3(256), 0, s, d_roff, nreads, a.bc_prefix, a.kmer, d_nblk);
      if (xs_exscan(ctx, (const u32 *) d_nblk, nreads, d_boff, ctx->d_scratch + 2200) != FK_OK) return (FK_ENOMEM);
      FK_LAUNCH_CHECK(ctx);
      FK_HIP(ctx, hipMemcpyAsync(ctx->h_scratch + 2200, ctx->d_scratch + 2200, 8, hipMemcpyDeviceToHost, s));
      FK_HIP(ctx, hipMemcpyAsync(ctx->h_scratch + XS_OVF_WORD, ctx->d_scratch + XS_OVF_WORD, 8, hipMemcpyDeviceToHost, s));
      FK_HIP(ctx, hipStreamSynchronize(s));
      if (xs_overflowed(ctx) != FK_OK) return (FK_EUNSUPPORTED);
      const int64_t nblocks = (int64_t) ctx->h_scratch[2200];
      if (nblocks > nreads && nblocks < 0x7fffffffll)          // (some read is longer than a block)
        { const int64_t nb4 = (nblocks * 4 + 63) & ~63ll;
          char *bb = (char *) fk_slot(ctx, FK_SLOT_XS_BLOCKS, 7 * nb4 + (nblocks + 1) * 8 + 256);
          if (bb == NULL) return (FK_ENOMEM);
          u32 *blk_read = (u32 *) bb, *blk_j = (u32 *) (bb + nb4), *blk_p0 = (u32 *) (bb + 2 * nb4), *flag = (u32 *) (bb + 3 * nb4);
          u32 *seg_read = (u32 *) (bb + 4 * nb4), *seg_p0 = (u32 *) (bb + 5 * nb4), *seg_p1 = (u32 *) (bb + 6 * nb4);
          u64 *soff = (u64 *) (bb + 7 * nb4);
          const unsigned gb = (unsigned) ((nblocks + 255) / 256);
          hipLaunchKernelGGL(k_xs_blkread, dim3(gr), dim3(256), 0, s, (const u32 *) d_nblk, (const u64 *) d_boff, nreads, blk_read, blk_j);
          hipLaunchKernelGGL(k_xs_find, dim3((unsigned) ((nblocks + 127) / 128)), dim3(128), 0, s, a, (const u32 *) blk_read,
                             (const u32 *) blk_j, nblocks, blk_p0, flag);
          if (xs_exscan(ctx, (const u32 *) flag, nblocks, soff, ctx->d_scratch + 2201) != FK_OK) return (FK_ENOMEM);
          hipLaunchKernelGGL(k_xs_segs, dim3(gb), dim3(256), 0, s, (const u32 *) blk_read, (const u32 *) blk_p0, (const u32 *) flag,
                             (const u64 *) soff, nblocks, seg_read, seg_p0);
          FK_LAUNCH_CHECK(ctx);
          FK_HIP(ctx, hipMemcpyAsync(ctx->h_scratch + 2201, ctx->d_scratch + 2201, 8, hipMemcpyDeviceToHost, s));
          FK_HIP(ctx, hipMemcpyAsync(ctx->h_scratch + XS_OVF_WORD, ctx->d_scratch + XS_OVF_WORD, 8, hipMemcpyDeviceToHost, s));
          FK_HIP(ctx, hipStreamSynchronize(s));
          if (xs_overflowed(ctx) != FK_OK) return (FK_EUNSUPPORTED);
          nseg = (int64_t) ctx->h_scratch[2201];
          hipLaunchKernelGGL(k_xs_ends, dim3((unsigned) ((nseg + 255) / 256)), dim3(256), 0, s, (const u32 *) seg_read,
                             (const u32 *) seg_p0, nseg, seg_p1);
          FK_LAUNCH_CHECK(ctx);
          a.nseg = nseg; a.seg_read = seg_read; a.seg_p0 = seg_p0; a.seg_p1 = seg_p1;
          if (ctx->dbg_verbose)
            fprintf(stderr, "  exact split: %lld reads in %lld segments\n", (long long) nreads, (long long) nseg);
        }
    }
  d_cnt = (u32 *) fk_slot(ctx, FK_SLOT_EX_HEADS, nseg * nparts * 4);
  d_off = (u64 *) fk_slot(ctx, FK_SLOT_EX_KOFF, (nseg * nparts + 1) * 8);
  if (d_cnt == NULL || d_off == NULL)
    return (FK_ENOMEM);
  a.cnt = d_cnt;
  a.off = d_off;
  const unsigned grid = (unsigned) ((nseg + XS_THREADS - 1) / XS_THREADS);
  FK_HIP(ctx, hipMemsetAsync(d_inst, 0, 72 * sizeof(u64), s));
  if (nparts == 1)
    hipLaunchKernelGGL((k_split_exact<false, true>), dim3(grid), dim3(XS_THREADS), 0, s, a);
  else
    hipLaunchKernelGGL((k_split_exact<false, false>), dim3(grid), dim3(XS_THREADS), 0, s, a);
  if (xs_exscan(ctx, (const u32 *) d_cnt, nseg * nparts, d_off, d_inst + 64) != FK_OK) return (FK_ENOMEM);
  FK_LAUNCH_CHECK(ctx);
  FK_HIP(ctx, hipMemcpyAsync(ctx->h_scratch, d_inst, 72 * sizeof(u64), hipMemcpyDeviceToHost, s));
  for (int b = 1; b < nparts; b++)                       // where bucket b starts: the scan at its first read
    FK_HIP(ctx, hipMemcpyAsync(ctx->h_scratch + 128 + b, d_off + (int64_t) b * nseg, 8, hipMemcpyDeviceToHost, s));
  FK_HIP(ctx, hipMemcpyAsync(ctx->h_scratch + XS_OVF_WORD, ctx->d_scratch + XS_OVF_WORD, 8, hipMemcpyDeviceToHost, s));
  FK_HIP(ctx, hipStreamSynchronize(s));
  if (xs_overflowed(ctx) != FK_OK) return (FK_EUNSUPPORTED);
  const int64_t ns = (int64_t) ctx->h_scratch[64];
  int64_t ni = 0;
  for (int x = 0; x < 64; x++)
    ni += (int64_t) ctx->h_scratch[x];
  *nsuper = ns;
  *ninst = ni;
  for (int b = 0; b < nparts; b++)
    { bucket_offs[b] = (b == 0) ? 0 : (int64_t) ctx->h_scratch[128 + b];
      if (b > 0) bucket_counts[b - 1] = bucket_offs[b] - bucket_offs[b - 1];
    }
  bucket_counts[nparts - 1] = ns - bucket_offs[nparts - 1];
  if (ns == 0)
    return (FK_OK);
  void *out = fk_slot(ctx, FK_SLOT_SM_A, ns * ctx->wid.smer_stride);
  if (out == NULL)
    return (FK_ENOMEM);
  a.out = (u32 *) out;
  FK_HIP(ctx, hipMemsetAsync(d_inst, 0, 72 * sizeof(u64), s));
  if (nparts == 1)
    hipLaunchKernelGGL((k_split_exact<true, true>), dim3(grid), dim3(XS_THREADS), 0, s, a);
  else
    hipLaunchKernelGGL((k_split_exact<true, false>), dim3(grid), dim3(XS_THREADS), 0, s, a);
  if (a.defer)
    hipLaunchKernelGGL(k_xs_pack, dim3((unsigned) ((ns + 255) / 256)), dim3(256), 0, s, a, ns);
  FK_LAUNCH_CHECK(ctx);
  *d_out = out;
  return (FK_OK);
}
#endif   // FK_HOST_EMU
