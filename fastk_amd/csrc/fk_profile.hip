// fk_profile.hip -- read profiles (FastK -p): for every read the count of each of its k-mers, in read
// order, compressed with the reference's profile codec (README "K-mer Profile Files"; the reference
// builds them in count.c:868-947 from its sorted super-mer lists and stitches them in merge.c).
//
// Here the profiles are produced from the two things the counting path leaves in HBM: the reads and
// the k-mer table with cutoff 1.
//   k_pf_hbuild  the table as an open-addressing dictionary: 64-byte lines of 4 (2) record slots
//   k_pf_counts  one look-up per base position: canonical k-mer from LDS-packed 2-bit codes, its home
//                line read with 16-byte loads -> u16 count (0 where the window is not all acgt)
//   k_pf_zeros   positions of the read terminators (0 bytes) -> read boundaries
//   k_pf_encode  one thread per read, 128 reads per workgroup staged through LDS: length pass, scan,
//                emit pass of the codec
// The codec output is the canonical one-byte-wherever-possible stream; it decodes to the same counts
// as the reference's files, whose run splits follow the reference's internal work panels (merge.c:65,711).
#include "fk_common.h"
#include <vector>

#define PF_TILE  4096
#define PF_HALO  128          // >= kmer - 1, multiple of 16
#define PF_NG    ((PF_TILE + PF_HALO) / 16)
#define PF_ZCH   16384        // bytes per workgroup of k_pf_zeros

// ---------------------------------------------------------------------------------------
// dictionary: open-addressing hash table over the table records.  A slot holds one record in its
// device layout (sdw dwords, padded to SD = 4 or 8 dwords); the dword with the count doubles as the
// occupancy flag (counts are >= 1).  Slots are grouped into 64-byte lines of G = 16 / SD slots: a key's
// home line is hash -> line, insertion takes the first free slot of the line (else the next line), a
// look-up reads the whole line with 16-byte loads -- one HBM sector per look-up at load factor 1/2,
// against index + binary-search probes of the sorted table.

template <int KW>
__device__ __forceinline__ u64 pf_hash(const u32 *d)
{ u64 h = 0x9E3779B97F4A7C15ull;
#pragma unroll
  for (int w = 0; w < KW; w++)
    { h = (h ^ d[w]) * 0xD6E8FEB86659FD93ull;
      h ^= h >> 32;
    }
  return (h);
}

template <int KW, int SW>
static __global__ __launch_bounds__(256) void k_pf_hbuild(const u32 *__restrict__ table, int64_t nt,
                                                           u32 kmask, u32 *__restrict__ slots, u64 nlines)
{ constexpr int SD = (SW <= 4) ? 4 : 8, G = 16 / SD, sdw = SW;
  const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (i >= nt)
    return;
  u32 r[SD], kd[KW];
#pragma unroll
  for (int w = 0; w < SD; w++)
    r[w] = (w < sdw) ? table[i * sdw + w] : 0u;
#pragma unroll
  for (int w = 0; w < KW; w++)
    kd[w] = (w == KW - 1) ? (r[w] & kmask) : r[w];
  const u64 h = pf_hash<KW>(kd);
  u64 line = (u64) (((unsigned __int128) h * nlines) >> 64);
  const int st = (int) (h & (G - 1));
  const u32 last = r[sdw - 1];
  while (true)
    { for (int j = 0; j < G; j++)
        { u32 *sl = slots + (line * G + ((st + j) & (G - 1))) * SD;
          if (atomicCAS(sl + (sdw - 1), 0u, last) == 0u)
            {
#pragma unroll
              for (int w = 0; w < SD; w++)
                if (w < sdw - 1)
                  sl[w] = r[w];
              return;
            }
        }
      line = (line + 1 == nlines) ? 0 : line + 1;
    }
}

// ---------------------------------------------------------------------------------------
// per-position counts

__device__ __forceinline__ u32 pf_rev2(u32 x)      // reverse the order of the 16 2-bit groups
{ u32 y = __brev(x);
  return (((y & 0x55555555u) << 1) | ((y >> 1) & 0x55555555u));
}

// count of the k-mer whose forward strand is a[] (K bases, first base in the top bits of a[0], pad
// bits zero): reverse complement, canonical = the smaller, home line of the dictionary
template <int KW, int SW>
__device__ __forceinline__ u32 pf_count_of(u32 *a, u32 tmask, int ps, const u32 *__restrict__ slots, u64 nlines,
                                           u32 kmask)
{ constexpr int SD = (SW <= 4) ? 4 : 8, G = 16 / SD;
  u32 b[KW + 1];
  u32 cnt = 0;
#pragma unroll
  for (int w = 0; w < KW; w++)
    b[w] = pf_rev2(~a[KW - 1 - w]);
  b[KW] = 0;
#pragma unroll
  for (int w = 0; w < KW; w++)
    b[w] = (u32) (((((u64) b[w]) << 32) | b[w + 1]) << ps >> 32);
  b[KW - 1] &= tmask;
  bool rc_less = (b[KW - 1] < a[KW - 1]);            // mask arithmetic, no short-circuit: no branches
#pragma unroll
  for (int w = KW - 2; w >= 0; w--)
    rc_less = (b[w] < a[w]) | ((b[w] == a[w]) & rc_less);
  if (rc_less)
    {
#pragma unroll
      for (int w = 0; w < KW; w++)
        a[w] = b[w];
    }
  u32 kd[KW];                                     // the key as the record's dwords
#pragma unroll
  for (int w = 0; w < KW; w++)
    kd[w] = __builtin_bswap32(a[w]);
  const u64 h = pf_hash<KW>(kd);
  u64 line = (u64) (((unsigned __int128) h * nlines) >> 64);
  while (true)
    { const uint4 *lp = (const uint4 *) (slots + line * 16);
      u32 v[16];
#pragma unroll
      for (int q = 0; q < 4; q++)
        { const uint4 x = lp[q];
          v[4 * q] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w;
        }
      bool hit = false, room = false;
#pragma unroll
      for (int j = 0; j < G; j++)
        { const u32 last = v[j * SD + SW - 1];
          bool eq = (last != 0u);
#pragma unroll
          for (int w = 0; w < KW; w++)
            eq = eq && (((w == KW - 1) ? (v[j * SD + w] & kmask) : v[j * SD + w]) == kd[w]);
          if (eq)
            { cnt = last >> 16;
              hit = true;
            }
          room = room || (last == 0u);
        }
      if (hit || room)
        break;
      line = (line + 1 == nlines) ? 0 : line + 1;
    }
  return (cnt);
}

template <int KW, int SW>
static __global__ __launch_bounds__(256) void k_pf_counts(const uint8_t *__restrict__ bases, int64_t n, int K,
                                                           const u32 *__restrict__ slots, u64 nlines,
                                                           u32 kmask, uint16_t *__restrict__ out)
{ __shared__ u32      fw[PF_NG + KW + 2];       // 2-bit codes, 16 per word, first base in the top bits
  __shared__ __attribute__((aligned(16))) uint16_t bad[PF_NG + 16];          // bit j of bad[g]: base 16 g + j is not acgt (or past n)
  const int     tid = threadIdx.x;
  const int64_t t0  = (int64_t) blockIdx.x * PF_TILE;

  for (int g = tid; g < PF_NG + KW + 2; g += 256)
    { const int64_t p = t0 + (int64_t) g * 16;
      u32 word = 0, bm = 0;
      if (g < PF_NG && p < n)
        { uint8_t c[16];
          if (p + 16 <= n)
            { const uint4 v = *(const uint4 *) (bases + p);
              const u32 d[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
              for (int j = 0; j < 16; j++)
                c[j] = (uint8_t) (d[j >> 2] >> (8 * (j & 3)));
            }
          else
            {
#pragma unroll
              for (int j = 0; j < 16; j++)
                c[j] = (p + j < n) ? bases[p + j] : (uint8_t) 0;
            }
#pragma unroll
          for (int j = 0; j < 16; j++)
            { const u32 l = c[j] | 0x20u;
              const bool ok = (l == 'a' || l == 'c' || l == 'g' || l == 't');
              const u32 x = (c[j] >> 1) & 3u;
              word |= (x ^ (x >> 1)) << (30 - 2 * j);
              bm   |= ok ? 0u : (1u << j);
            }
        }
      else
        bm = 0xffffu;
      fw[g] = word;
      if (g < PF_NG + 16)
        bad[g] = (uint16_t) bm;
    }
  __syncthreads();

  const int tb = K - 16 * (KW - 1);             // bases in the last key word
  const u32 tmask = (tb == 16) ? 0xffffffffu : ~(0xffffffffu >> (2 * tb));
  const int ps = 2 * (16 - tb);
  const u32 *bad32 = (const u32 *) bad;

#pragma unroll 1
  for (int j = 0; j < PF_TILE / 256; j++)
    { const int     i = tid + j * 256;
      const int64_t p = t0 + i;
      if (p >= n)
        break;
      // any bad base in [i, i+K) ?
      bool ok = (p + K <= n);
      { const int w0 = i >> 5, w1 = (i + K - 1) >> 5;
        u32 acc = 0;
        for (int w = w0; w <= w1; w++)
          { u32 m = bad32[w];
            if (w == w0) m &= 0xffffffffu << (i & 31);
            if (w == w1) m &= 0xffffffffu >> (31 - ((i + K - 1) & 31));
            acc |= m;
          }
        ok = ok && (acc == 0);
      }
      u32 cnt = 0;
      if (ok)
        { const int q = i >> 4, sh = (i & 15) * 2;
          u32 a[KW];
#pragma unroll
          for (int w = 0; w < KW; w++)
            a[w] = (u32) (((((u64) fw[q + w]) << 32) | fw[q + w + 1]) << sh >> 32);
          a[KW - 1] &= tmask;
          cnt = pf_count_of<KW, SW>(a, tmask, ps, slots, nlines, kmask);
        }
      out[p] = (uint16_t) cnt;
    }
}

// ---------------------------------------------------------------------------------------
// sharded run: the rank that owns a super-mer's bucket looks its k-mers up and the counts travel back

// k-mers per super-mer record (its length byte + 1)
static __global__ __launch_bounds__(256) void k_pf_smer_n(const u32 *__restrict__ smers, int64_t ns, int sww,
                                                           int smer_bytes, u32 *__restrict__ n)
{ const int64_t r = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (r < ns)
    n[r] = ((smers[r * sww + (smer_bytes >> 2)] >> (8 * (smer_bytes & 3))) & 0xffu) + 1u;
}

// counts of the k-mers of every record, in record order: out[koff[r] + j] = count of its j-th k-mer
template <int KW, int SW, int RWS>          // RWS: capacity in record dwords (4 or 8), sww the real stride
static __global__ __launch_bounds__(256) void k_pf_smer_counts(const u32 *__restrict__ smers, int64_t ns, int sww,
                                                                const u64 *__restrict__ koff, int K,
                                                                const u32 *__restrict__ slots, u64 nlines,
                                                                u32 kmask, uint16_t *__restrict__ out)
{ const int64_t r = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (r >= ns)
    return;
  u32 w[RWS + KW + 1];
#pragma unroll
  for (int q = 0; q < RWS + KW + 1; q++)
    w[q] = (q < RWS && q < sww) ? __builtin_bswap32(smers[r * sww + q]) : 0u;
  const int n = (int) (koff[r + 1] - koff[r]);
  const int tb = K - 16 * (KW - 1);
  const u32 tmask = (tb == 16) ? 0xffffffffu : ~(0xffffffffu >> (2 * tb));
  const int ps = 2 * (16 - tb);
  uint16_t *o = out + koff[r];
  for (int j = 0; j < n; j++)
    { // bases [j, j+K) of the record: shift the whole word array left by one base per step
      u32 a[KW];
#pragma unroll
      for (int q = 0; q < KW; q++)
        a[q] = w[q];
      a[KW - 1] &= tmask;
      o[j] = (uint16_t) pf_count_of<KW, SW>(a, tmask, ps, slots, nlines, kmask);
#pragma unroll
      for (int q = 0; q < RWS + KW; q++)
        w[q] = (w[q] << 2) | (w[q + 1] >> 30);
      w[RWS + KW] <<= 2;
    }
}

// the counts that came back, placed at the positions the records were cut from
static __global__ __launch_bounds__(256) void k_pf_scatter(const u64 *__restrict__ pos, const u64 *__restrict__ koff,
                                                            int64_t ns, const uint16_t *__restrict__ in,
                                                            uint16_t *__restrict__ cnts)
{ const int64_t r = (int64_t) blockIdx.x * 256 + threadIdx.x;
  if (r >= ns)
    return;
  const u64 p = pos[r] >> 1;
  const bool flip = (pos[r] & 1u) != 0;
  const int n = (int) (koff[r + 1] - koff[r]);
  const uint16_t *c = in + koff[r];
  for (int j = 0; j < n; j++)
    cnts[p + (flip ? n - 1 - j : j)] = c[j];
}

// ---------------------------------------------------------------------------------------
// read boundaries: positions of the 0 bytes

template <bool EMIT>
static __global__ __launch_bounds__(256) void k_pf_zeros(const uint8_t *__restrict__ bases, int64_t n,
                                                          u32 *__restrict__ cnt, const u64 *__restrict__ off,
                                                          int64_t *__restrict__ ends)
{ __shared__ u32 tmp[8];
  const int64_t p0 = (int64_t) blockIdx.x * PF_ZCH + (int64_t) threadIdx.x * (PF_ZCH / 256);
  u64 zm = 0;                                  // PF_ZCH / 256 = 64 bytes per thread
#pragma unroll
  for (int k = 0; k < 4; k++)
    { const int64_t p = p0 + 16 * k;
      if (p + 16 <= n)
        { const uint4 v = *(const uint4 *) (bases + p);
          const u32 d[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
          for (int j = 0; j < 16; j++)
            if (((d[j >> 2] >> (8 * (j & 3))) & 0xffu) == 0)
              zm |= 1ull << (16 * k + j);
        }
      else
        for (int j = 0; j < 16; j++)
          if (p + j < n && bases[p + j] == 0)
            zm |= 1ull << (16 * k + j);
    }
  u32 mine = (u32) __popcll(zm), tot;
  const u32 ex = fk_block_exscan_256<u32>(mine, tmp, &tot);
  if (!EMIT)
    { if (threadIdx.x == 0)
        cnt[blockIdx.x] = tot;
      return;
    }
  int64_t o = (int64_t) off[blockIdx.x] + ex;
  while (zm != 0)
    { const int j = __ffsll((unsigned long long) zm) - 1;
      ends[o++] = p0 + j;
      zm &= zm - 1;
    }
}

// ---------------------------------------------------------------------------------------
// the codec (README: first count in 1-2 bytes, then forward differences: 00x = run of x equal
// counts (x <= 63), 01x = 6-bit two's complement difference, 1x.y = 15-bit difference mod 2^15)

#define PF_ER      128          // reads per workgroup of k_pf_encode (one thread each)
#define PF_ECAP    20480        // staged counts per workgroup (40 KB of LDS)
#define PF_OCAP    24576        // staged codec bytes per workgroup

// A workgroup takes PF_ER consecutive reads.  Their counts are one contiguous stretch of cnts: when it
// fits, it is staged in LDS with coalesced loads (a thread walking its own read in global memory
// touches a different cache line than its 63 neighbours on every step) and, in the emit pass, the
// codec bytes are collected in LDS and written out as one contiguous run.
template <bool EMIT>
static __global__ __launch_bounds__(PF_ER) void k_pf_encode(const uint16_t *__restrict__ cnts,
                                                            const int64_t *__restrict__ ends, int64_t nreads,
                                                            int64_t nbytes, int K, int bc, u32 *__restrict__ lens,
                                                            const u64 *__restrict__ offs, uint8_t *__restrict__ out)
{ __shared__ __attribute__((aligned(16))) uint16_t lc[PF_ECAP];
  __shared__ __attribute__((aligned(16))) uint8_t  lo[EMIT ? PF_OCAP : 16];
  const int64_t r0 = (int64_t) blockIdx.x * PF_ER;
  const int64_t r  = r0 + threadIdx.x;
  const int64_t rl = (r0 + PF_ER < nreads ? r0 + PF_ER : nreads) - 1;        // last read of the group
  const int64_t g0 = (r0 == 0) ? 0 : ends[r0 - 1] + 1;                       // first position of the group
  const int64_t g1 = ends[rl] < nbytes ? ends[rl] : nbytes;
  const int64_t a0 = g0 & ~(int64_t) 7;                                      // 16-byte aligned start
  const bool    staged = (g1 - a0 <= PF_ECAP);
  if (staged)
    { const int n8 = (int) ((g1 - a0 + 7) >> 3);
      const uint4 *src = (const uint4 *) (cnts + a0);      // cnts has 64 bytes of slack at the end
      for (int i = threadIdx.x; i < n8; i += PF_ER)
        ((uint4 *) lc)[i] = src[i];
    }
  u64 ob = 0;
  bool ostaged = false;
  if (EMIT)
    { ob = offs[r0];
      ostaged = (offs[rl + 1] - ob <= PF_OCAP);
    }
  __syncthreads();

  u32 len = 0;
  if (r < nreads)
    { const int64_t s0 = (r == 0) ? 0 : ends[r - 1] + 1;
      const int64_t e = ends[r] < nbytes ? ends[r] : nbytes;
      const int64_t s = (s0 + bc < e) ? s0 + bc : e;        // -bc: the profile is the trimmed read's
      const int64_t np = e - s - K + 1;
      uint8_t *o = NULL;
      if (EMIT)
        o = ostaged ? lo + (offs[r] - ob) : out + offs[r];
      if (np > 0)
        { const uint16_t *c = staged ? lc + (s - a0) : cnts + s;
          u32 prev = c[0];
          if (prev < 128)
            { if (EMIT) o[len] = (uint8_t) prev;
              len += 1;
            }
          else
            { if (EMIT) { o[len] = (uint8_t) (0x80u | (prev >> 8)); o[len + 1] = (uint8_t) prev; }
              len += 2;
            }
          u32 run = 0;
          for (int64_t j = 1; j < np; j++)
            { const u32 x = c[j];
              if (x == prev)
                { if (++run == 63)
                    { if (EMIT) o[len] = 63;
                      len += 1;
                      run = 0;
                    }
                  continue;
                }
              if (run != 0)
                { if (EMIT) o[len] = (uint8_t) run;
                  len += 1;
                  run = 0;
                }
              const int d = (int) x - (int) prev;
              if (d > -32 && d < 32)
                { if (EMIT) o[len] = (uint8_t) (0x40u | ((u32) d & 0x3fu));
                  len += 1;
                }
              else
                { const u32 dd = (u32) d & 0x7fffu;
                  if (EMIT) { o[len] = (uint8_t) (0x80u | (dd >> 8)); o[len + 1] = (uint8_t) dd; }
                  len += 2;
                }
              prev = x;
            }
          if (run != 0)
            { if (EMIT) o[len] = (uint8_t) run;
              len += 1;
            }
        }
      if (!EMIT)
        lens[r] = len;
    }
  if (EMIT && ostaged)
    { __syncthreads();
      const int tot = (int) (offs[rl + 1] - ob);
      uint8_t *dst = out + ob;
      const int head = (int) ((4 - ((uintptr_t) dst & 3)) & 3);           // bytes up to dword alignment
      if ((int) threadIdx.x < head && (int) threadIdx.x < tot)
        dst[threadIdx.x] = lo[threadIdx.x];
      const int nd = (tot > head) ? (tot - head) >> 2 : 0;
      for (int i = threadIdx.x; i < nd; i += PF_ER)
        { const uint8_t *q = lo + head + 4 * i;
          ((u32 *) (dst + head))[i] = (u32) q[0] | ((u32) q[1] << 8) | ((u32) q[2] << 16) | ((u32) q[3] << 24);
        }
      for (int i = head + 4 * nd + (int) threadIdx.x; i < tot; i += PF_ER)
        dst[i] = lo[i];
    }
}

// ---------------------------------------------------------------------------------------
// The reference's OWN bytes (exact_parts runs).  Its .prof stream is not a function of the counts alone: the profile
// of a read is stitched from the profiles of its super-mers (count.c:868-947 encodes every super-mer's counts on its
// own: first count, differences -- one byte for |d| < 32 --, runs of equal counts in bytes of up to 63, last count),
// and Merge_Profiles (merge.c:263-716) joins them: the difference across a junction takes ONE byte only for
// -30 <= d <= 31 (merge.c:456,590), a run of equal counts may continue across junctions, stretches of invalid k-mers
// arrive as packets of at most MAX_NRUN zeros with run numbers of their own (split.c:1167-1232), and -- the part that
// depends on nothing in the data -- a pending run is written out whenever the run number passes a multiple of
// PAN_SIZE = 1024 NPARTS (merge.c:263-267,706-716), counted per input thread.  So the kernel below walks every read
// through Distribute_Block's super-mer rule (the state machine of fk_split_exact.hip) and feeds the events -- super-mer
// of n k-mers, packet of n invalid k-mers, end of read -- to that stitching, byte for byte.
//   MODE 0: run numbers a read consumes;  MODE 1: bytes of its profile;  MODE 2: the bytes.
#define PX_THREADS 64
#define PX_RING    256

struct ExactProfArgs
{ const unsigned char *bases;
  const int64_t *ends;          // [nreads] position of each read's terminator
  int64_t   nreads, nbytes;
  int       kmer, bc_prefix;
  int       tran[4];
  int       pad_len;            // minimizer length 5 + PAD
  const uint16_t *cnts;         // count of the k-mer starting at every position
  const u64 *rid0;              // [nreads] first run number of the read, counted from its input thread's first read
  u32       pan;                // PAN_SIZE
  u32      *nrun;               // MODE 0 out
  u32      *lens;               // MODE 1 out
  const u64 *offs;              // MODE 2 in
  uint8_t  *out;
};

template <int MODE>
static __global__ __launch_bounds__(PX_THREADS) void k_pf_exact(ExactProfArgs a)
{ const int64_t r = (int64_t) blockIdx.x * PX_THREADS + threadIdx.x;
  if (r >= a.nreads)
    return;
  const int K = a.kmer, KM1 = K - 1;
  const int PL1 = a.pad_len - 1;
  const int MS  = K - PL1;
  const int MAX_NRUN = 1 + 63 * 2 * MS;                     // split.c:77
  const u32 vmsk = (1u << (2 * a.pad_len)) - 1u;
  const int64_t s0 = (r == 0) ? 0 : a.ends[r - 1] + 1;
  const int64_t e0 = a.ends[r] < a.nbytes ? a.ends[r] : a.nbytes;
  const int64_t sb = (s0 + a.bc_prefix < e0) ? s0 + a.bc_prefix : e0;
  const unsigned char *s = a.bases + sb;
  const uint16_t *cn = a.cnts + sb;                          // cn[j]: the k-mer that starts at base j of the read
  const int q = (int) (e0 - sb);

  // ---- the stitcher (merge.c:395-716)
  u64  rid = (MODE == 0) ? 0 : a.rid0[r];
  u32  nev = 0, len = 0;
  bool wlast = true;
  u32  lcont = 0, lz = 0;
  uint8_t *o = (MODE == 2) ? a.out + a.offs[r] : NULL;
  auto put = [&](u32 b) { if (MODE == 2) o[len] = (uint8_t) b; len += 1; };
  auto next_run = [&]()
    { // a new run number: at a multiple of PAN_SIZE the previous panel has ended, its pending run was written
      if (MODE != 0 && rid % a.pan == 0 && lz != 0)
        { put(lz); lz = 0; }
      rid += 1;
      nev += 1;
    };
  auto junction = [&](u32 d)                                 // d = difference as 16 bits
    { if (d == 0)
        { lz += 1;
          if (lz >= 63) { put(63); lz = 0; }
        }
      else
        { if (lz) { put(lz); lz = 0; }
          if (d > 0xffe1u || d < 32u)
            put(0x40u | (d & 0x3fu));
          else
            { put(0x80u | ((d >> 8) & 0xffu)); put(d & 0xffu); }
        }
    };
  auto ev_zeros = [&](int n, bool last)                      // a packet of n invalid k-mers
    { next_run();
      if (MODE == 0) return;
      if (wlast)
        { put(0); lz = 0; }
      else
        junction((0u - lcont) & 0xffffu);
      if (n > 1)
        { int t = n + (int) lz - 1;
          for ( ; t >= 63; t -= 63)
            put(63);
          lz = (u32) t;
        }
      lcont = 0;
      if (last)
        { if (lz) { put(lz); lz = 0; }
          wlast = true;
        }
      else
        wlast = false;
    };
  auto ev_interval = [&](int lo, int hi, bool last)          // invalid k-mers [lo, hi): packets of at most MAX_NRUN
    { int f = lo;
      for (int l = hi - MAX_NRUN; f < l; f += MAX_NRUN)
        ev_zeros(MAX_NRUN, false);
      ev_zeros(hi - f, last);
    };
  auto ev_super = [&](int first_end, int n, bool last)       // the k-mers ending at first_end .. first_end + n - 1
    { next_run();
      if (MODE == 0) return;
      const uint16_t *c = cn + (first_end - KM1);
      uint8_t tok[160];                                        // count.c:868-947: at most 2 bytes per difference
      int nt = 0;
      u32 p0 = c[0], pv = p0, run = 0;
      for (int j = 1; j < n; j++)
        { const u32 x = c[j];
          if (x == pv)
            { if (run > 0)
                { if (run >= 63) { tok[nt++] = (uint8_t) run; run = 1; }
                  else run += 1;
                }
              else
                run = 1;
            }
          else
            { if (run > 0) { tok[nt++] = (uint8_t) run; run = 0; }
              const int d = (int) x - (int) pv;
              if (d > -32 && d < 32)
                tok[nt++] = (uint8_t) (0x40u | ((u32) d & 0x3fu));
              else
                { tok[nt++] = (uint8_t) (0x80u | (((u32) d >> 8) & 0xffu));
                  tok[nt++] = (uint8_t) ((u32) d & 0xffu);
                }
            }
          pv = x;
        }
      if (run > 0) tok[nt++] = (uint8_t) run;
      // merge.c:541-700
      if (wlast)
        { if (p0 < 128) put(p0);
          else { put(0x80u | (p0 >> 8)); put(p0 & 0xffu); }
          lz = 0;
        }
      else
        junction((p0 - lcont) & 0xffffu);
      lcont = p0;
      if (n > 1)
        { lcont = pv;
          int i0 = 0;
          if (lz)
            while (i0 < nt && tok[i0] < 64)
              { lz += tok[i0];
                if (lz >= 63) { put(63); lz -= 63; }
                i0 += 1;
              }
          if (i0 < nt)
            { if (lz) { put(lz); lz = 0; }
              bool one = true;
              for (int u = i0; u < nt; u++)
                if (tok[u] & 0x80) { one = false; u += 1; }
                else one = true;
              int stop = nt;
              if (one)
                { lz = tok[nt - 1];
                  if (lz < 63) stop = nt - 1;
                  else lz = 0;
                }
              for (int u = i0; u < stop; u++)
                put(tok[u]);
            }
        }
      wlast = last;
      if (last && lz) { put(lz); lz = 0; }
    };

  if (q < K)                                                 // split.c:1079-1086: a run number, no profile
    { next_run();
      if (MODE == 0) a.nrun[r] = nev;
      if (MODE == 1) a.lens[r] = 0;
      return;
    }

  // ---- Distribute_Block's walk (split.c:1096-1393), events instead of records.  A super-mer is handed over when
  // the next event (or the end of the read) is known: only then is it known whether it ends the read.
  int rmsk = 1;
  while (rmsk < K) rmsk <<= 1;
  rmsk = 2 * rmsk - 1;
  u32 ring[PX_RING];
  const int t0 = a.tran[0], t1 = a.tran[1], t2 = a.tran[2], t3 = a.tran[3];
  auto code_of = [&](unsigned ch) -> int
    { const unsigned u = ch & 0xDFu;
      return (u == 0x41u) ? 0 : (u == 0x43u) ? 1 : (u == 0x47u) ? 2 : (u == 0x54u) ? 3 : 4;
    };
  auto fwv = [&](int code) -> u32 { return (u32) (code == 1 ? t1 : code == 2 ? t2 : code == 3 ? t3 : t0); };
  auto rcv = [&](int code) -> u32 { return (u32) (code == 1 ? t2 : code == 2 ? t1 : code == 3 ? t0 : t3) << (2 * PL1); };
  int  pend_e = -1, pend_n = 0;                              // a super-mer waiting for its "ends the read" flag
  auto flush_pending = [&](bool last)
    { if (pend_n > 0) ev_super(pend_e, pend_n, last);
      pend_n = 0;
    };

  u32 c = 0, u = 0, mp = 0, mc = vmsk + 1u;
  int m = 0, p;
  int ilo = -1, ihi = -1, plo = 0, phi = -1;                 // nfst, nlst, pfst, plst
  for (p = 0; p < K; p++)
    { const int code = code_of(s[p]);
      c = ((c << 2) | fwv(code)) & vmsk;
      u = (u >> 2) | rcv(code);
      if (p >= PL1)
        { mp = (u < c) ? u : c;
          ring[p & rmsk] = mp;
          if (mp < mc) { m = p; mc = mp; }
        }
      if (code >= 4)
        { if (p > ihi)
            ilo = KM1;
          ihi = p + K;
        }
    }
  int  last = KM1;
  bool done = false;
  for (p = K; !done; p++)
    { int  code = 0;
      bool closing, force;
      if (p < q)
        { code = code_of(s[p]);
          c = ((c << 2) | fwv(code)) & vmsk;
          u = (u >> 2) | rcv(code);
          mp = (u < c) ? u : c;
          ring[p & rmsk] = mp;
          force   = (p - m >= MS);
          closing = force || (mp < mc);
        }
      else
        { if (ihi == q)                                      // split.c:1342: no forced closing then
            break;
          mp = mc;
          force = closing = true;
          done = true;
        }
      if (closing)
        { int n;
          if (ihi >= last)
            { if (ihi <= p)
                { last = ihi;
                  flush_pending(false);
                  ev_interval(ilo, ihi, false);
                  ihi = -1;
                  n = p - last;
                }
              else
                { if (phi > last)
                    { last = phi;
                      flush_pending(false);
                      ev_interval(plo, phi, false);
                      phi = -1;
                    }
                  n = ilo - last;
                }
            }
          else
            n = p - last;
          if (n > 0)
            { flush_pending(false);
              pend_e = last; pend_n = n;
            }
          if (done)
            break;
          if (force)
            { m += 1;
              mc = ring[m & rmsk];
              for (int j = m + 1; j <= p; j++)
                { const u32 v = ring[j & rmsk];
                  if (v <= mc) { m = j; mc = v; }
                }
            }
          else
            { m = p; mc = mp; }
          last = p;
        }
      if (code >= 4)
        { if (p > ihi)
            { plo = ilo; phi = ihi; ilo = p; }
          ihi = p + K;
        }
    }
  if (ihi >= q)                                              // split.c:1349-1377: invalid k-mers up to the end
    { flush_pending(false);
      ev_interval(ilo, q, true);
    }
  else
    flush_pending(true);
  if (MODE == 0) a.nrun[r] = nev;
  if (MODE == 1) a.lens[r] = len;
}

// exclusive scan of n u32 values into u64 offsets with many workgroups: block sums, a scan of the sums
// (k_exscan_tiles, one workgroup), block-local scans on top of them; out[n] gets the total
static __global__ __launch_bounds__(256) void k_pf_blocksum(const u32 *__restrict__ in, int64_t n, u32 *__restrict__ sums)
{ __shared__ u32 tmp[8];
  const int64_t i0 = ((int64_t) blockIdx.x * 256 + threadIdx.x) * 16;
  u32 mine = 0;
#pragma unroll
  for (int k = 0; k < 16; k++)
    mine += (i0 + k < n) ? in[i0 + k] : 0u;
  u32 tot;
  (void) fk_block_exscan_256<u32>(mine, tmp, &tot);
  if (threadIdx.x == 0)
    sums[blockIdx.x] = tot;
}

static __global__ __launch_bounds__(256) void k_pf_blockscan(const u32 *__restrict__ in, int64_t n,
                                                             const u64 *__restrict__ base, u64 *__restrict__ out)
{ __shared__ u64 tmp[8];
  const int64_t i0 = ((int64_t) blockIdx.x * 256 + threadIdx.x) * 16;
  u32 v[16];
  u64 mine = 0;
#pragma unroll
  for (int k = 0; k < 16; k++)
    { v[k] = (i0 + k < n) ? in[i0 + k] : 0u;
      mine += v[k];
    }
  u64 tot;
  u64 run = base[blockIdx.x] + fk_block_exscan_256<u64>(mine, tmp, &tot);
#pragma unroll
  for (int k = 0; k < 16; k++)
    { if (i0 + k < n)
        out[i0 + k] = run;
      run += v[k];
    }
}

// ---------------------------------------------------------------------------------------

template <int KW, int SW>
static void pf_launch_build(hipStream_t s, const u32 *table, int64_t nt, u32 kmask, u32 *slots, u64 nlines)
{ if (nt > 0)
    hipLaunchKernelGGL((k_pf_hbuild<KW, SW>), dim3((unsigned) ((nt + 255) / 256)), dim3(256), 0, s, table, nt, kmask,
                       slots, nlines);
}

template <int KW, int SW>
static void pf_launch_counts(hipStream_t s, const u32 *slots, u64 nlines, u32 kmask, int64_t ntiles,
                             const uint8_t *bases, int64_t n, int K, uint16_t *out)
{ hipLaunchKernelGGL((k_pf_counts<KW, SW>), dim3((unsigned) ntiles), dim3(256), 0, s, bases, n, K, slots, nlines,
                     kmask, out);
}

template <int KW, int SW>
static void pf_launch_smer_counts(hipStream_t s, const u32 *smers, int64_t ns, int sww, const u64 *koff, int K,
                                  const u32 *slots, u64 nlines, u32 kmask, uint16_t *out)
{ const unsigned nb = (unsigned) ((ns + 255) / 256);
  if (sww <= 4)
    hipLaunchKernelGGL((k_pf_smer_counts<KW, SW, 4>), dim3(nb), dim3(256), 0, s, smers, ns, sww, koff, K, slots,
                       nlines, kmask, out);
  else
    hipLaunchKernelGGL((k_pf_smer_counts<KW, SW, 8>), dim3(nb), dim3(256), 0, s, smers, ns, sww, koff, K, slots,
                       nlines, kmask, out);
}

// dispatch on (key dwords, record dwords) of the k-mer records
#define PF_DISPATCH(ctx, KW, sdw, CALL)                                                              \
  do                                                                                                 \
    { if      (KW == 1 && sdw == 1) { CALL(1, 1); } else if (KW == 1 && sdw == 2) { CALL(1, 2); }     \
      else if (KW == 2 && sdw == 2) { CALL(2, 2); } else if (KW == 2 && sdw == 3) { CALL(2, 3); }     \
      else if (KW == 3 && sdw == 3) { CALL(3, 3); } else if (KW == 3 && sdw == 4) { CALL(3, 4); }     \
      else if (KW == 4 && sdw == 4) { CALL(4, 4); } else if (KW == 4 && sdw == 5) { CALL(4, 5); }     \
      else                                                                                           \
        { fk_set_error(ctx, "profiles: k = %d not supported", ctx->wid.kmer);                        \
          return (FK_EUNSUPPORTED);                                                                  \
        }                                                                                            \
    }                                                                                                \
  while (0)

struct pf_dict
{ u32 *slots;
  u64  nlines;
  u32  kmask;
  int  KW, sdw;
};

// the table (nt records of kmer_stride bytes, distinct k-mers, any order, counts >= 1) as the
// dictionary of the look-up kernels, load factor <= 1/2
static int pf_dictionary(fk_ctx *ctx, const void *d_table, int64_t nt, pf_dict *d)
{ const fk_widths &w = ctx->wid;
  hipStream_t s = ctx->stream;
  d->KW  = (w.kmer_bytes + 3) / 4;
  d->sdw = w.kmer_stride / 4;
  const int kb_last = w.kmer_bytes - 4 * (d->KW - 1);        // key bytes in the last key dword
  d->kmask = (kb_last == 4) ? 0xffffffffu : ((1u << (8 * kb_last)) - 1u);
  const int G = (d->sdw <= 4) ? 4 : 2;                       // slots per 64-byte line
  d->nlines = (u64) std::max<int64_t>(1, (2 * nt + G - 1) / G);
  // the dictionary of an unchanged table is kept between calls (reads profiled piece by piece)
  if (ctx->pf_dict_table == d_table && ctx->pf_dict_nt == nt && d_table != NULL
      && ctx->slot_ptr[FK_SLOT_PF_IDX] != NULL && ctx->slot_cap[FK_SLOT_PF_IDX] >= (int64_t) d->nlines * 64)
    { d->slots = (u32 *) ctx->slot_ptr[FK_SLOT_PF_IDX];
      return (FK_OK);
    }
  ctx->pf_dict_table = NULL;
  d->slots = (u32 *) fk_slot(ctx, FK_SLOT_PF_IDX, (int64_t) d->nlines * 64);
  if (d->slots == NULL) return (FK_ENOMEM);
  FK_HIP(ctx, hipMemsetAsync(d->slots, 0, (size_t) d->nlines * 64, s));
#define PF_CALL(kw, sw) pf_launch_build<kw, sw>(s, (const u32 *) d_table, nt, d->kmask, d->slots, d->nlines)
  PF_DISPATCH(ctx, d->KW, d->sdw, PF_CALL);
#undef PF_CALL
  FK_LAUNCH_CHECK(ctx);
  ctx->pf_dict_table = d_table;
  ctx->pf_dict_nt    = nt;
  return (FK_OK);
}

static int fkx_profile_encode(fk_ctx *ctx, const uint8_t *bases, int64_t nbytes, const uint16_t *cnts,
                              int64_t *nreads_out, int64_t *nprof_out, void **d_data, uint64_t **d_offs);

// Profiles of the reads in d_bases[0..nbytes) (reads end at 0 bytes; a last read without terminator
// ends at nbytes) against the table d_table (nt records of kmer_stride bytes with distinct k-mers, any
// order, counts >= 1).  Results stay in HBM: *d_data (nprof bytes) and *d_offs (nreads + 1 offsets).
int fkx_profiles(fk_ctx *ctx, const void *d_bases, int64_t nbytes, const void *d_table, int64_t nt,
                 int64_t *nreads_out, int64_t *nprof_out, void **d_data, uint64_t **d_offs)
{ hipStream_t s = ctx->stream;
  const int K = ctx->wid.kmer;
  const uint8_t *bases = (const uint8_t *) d_bases;

  *nreads_out = 0;
  *nprof_out = 0;
  *d_data = NULL;
  *d_offs = NULL;
  if (nbytes <= 0)
    return (FK_OK);
  if (K - 1 > PF_HALO)
    { fk_set_error(ctx, "profiles: k = %d not supported", K);
      return (FK_EUNSUPPORTED);
    }

  // 1. + 2. dictionary, then counts per position
  pf_dict d;
  int rc = pf_dictionary(ctx, d_table, nt, &d);
  if (rc != FK_OK) return (rc);
  uint16_t *cnts = (uint16_t *) fk_slot(ctx, FK_SLOT_PF_CNT, nbytes * 2 + 64);
  if (cnts == NULL) return (FK_ENOMEM);
  const int64_t ntiles = (nbytes + PF_TILE - 1) / PF_TILE;
#define PF_CALL(kw, sw) pf_launch_counts<kw, sw>(s, d.slots, d.nlines, d.kmask, ntiles, bases, nbytes, K, cnts)
  PF_DISPATCH(ctx, d.KW, d.sdw, PF_CALL);
#undef PF_CALL
  FK_LAUNCH_CHECK(ctx);
  return fkx_profile_encode(ctx, bases, nbytes, cnts, nreads_out, nprof_out, d_data, d_offs);
}

// Sharded run, owner side: counts of the k-mers of ns super-mer records (device stride) from the
// dictionary of d_table, record after record: d_out[0 .. *ninst) u16.  cap in counts.
int fkx_profile_lookup_supermers(fk_ctx *ctx, const void *d_smers, int64_t ns, const void *d_table, int64_t nt,
                                 void *d_out, int64_t cap, int64_t *ninst)
{ const fk_widths &w = ctx->wid;
  hipStream_t s = ctx->stream;
  *ninst = 0;
  if (ns == 0)
    return (FK_OK);
  pf_dict d;
  int rc = pf_dictionary(ctx, d_table, nt, &d);
  if (rc != FK_OK) return (rc);
  const int sww = w.smer_stride / 4;
  u32 *n   = (u32 *) fk_slot(ctx, FK_SLOT_PF_LEN, ns * 4 + 64);
  u64 *off = (u64 *) fk_slot(ctx, FK_SLOT_PF_OFF, (ns + 1) * 8 + 64);
  const int64_t nblk = (ns + 4095) / 4096;
  u32 *bs = (u32 *) fk_slot(ctx, FK_SLOT_PF_ZC, nblk * 4 + 64);
  u64 *bo = (u64 *) fk_slot(ctx, FK_SLOT_PF_ZO, nblk * 8 + 64);
  if (n == NULL || off == NULL || bs == NULL || bo == NULL) return (FK_ENOMEM);
  hipLaunchKernelGGL(k_pf_smer_n, dim3((unsigned) ((ns + 255) / 256)), dim3(256), 0, s, (const u32 *) d_smers, ns,
                     sww, w.smer_bytes, n);
  hipLaunchKernelGGL(k_pf_blocksum, dim3((unsigned) nblk), dim3(256), 0, s, (const u32 *) n, ns, bs);
  hipLaunchKernelGGL(k_exscan_tiles, dim3(1), dim3(256), 0, s, (const u32 *) bs, nblk, bo, off + ns);
  hipLaunchKernelGGL(k_pf_blockscan, dim3((unsigned) nblk), dim3(256), 0, s, (const u32 *) n, ns, (const u64 *) bo, off);
  FK_LAUNCH_CHECK(ctx);
  FK_HIP(ctx, hipMemcpyAsync(ctx->h_scratch, off + ns, 8, hipMemcpyDeviceToHost, s));
  FK_HIP(ctx, hipStreamSynchronize(s));
  *ninst = (int64_t) ctx->h_scratch[0];
  if (d_out == NULL || cap == 0)
    return (FK_OK);
  if (cap < *ninst)
    { fk_set_error(ctx, "profile look-up: buffer holds %lld counts, %lld needed", (long long) cap, (long long) *ninst);
      return (FK_EINVAL);
    }
  if (sww > 8)
    { fk_set_error(ctx, "profiles: k = %d not supported", w.kmer);
      return (FK_EUNSUPPORTED);
    }
#define PF_CALL(kw, sw) pf_launch_smer_counts<kw, sw>(s, (const u32 *) d_smers, ns, sww, (const u64 *) off, w.kmer, \
                                                      d.slots, d.nlines, d.kmask, (uint16_t *) d_out)
  PF_DISPATCH(ctx, d.KW, d.sdw, PF_CALL);
#undef PF_CALL
  FK_LAUNCH_CHECK(ctx);
  FK_HIP(ctx, hipStreamSynchronize(s));
  return (FK_OK);
}

// Sharded run, reader side: counts that came back for the ns records this rank sent (same order),
// placed at the positions d_pos the records were cut from; reset clears the per-position array first.
// fkx_profile_encode_counts then turns the array into profiles.
int fkx_profile_scatter(fk_ctx *ctx, const void *d_smers, const void *d_pos, int64_t ns, const void *d_in,
                        int64_t nbytes, bool reset)
{ const fk_widths &w = ctx->wid;
  hipStream_t s = ctx->stream;
  uint16_t *cnts = (uint16_t *) fk_slot(ctx, FK_SLOT_PF_CNT, nbytes * 2 + 64);
  if (cnts == NULL) return (FK_ENOMEM);
  if (reset)
    FK_HIP(ctx, hipMemsetAsync(cnts, 0, (size_t) nbytes * 2 + 64, s));
  if (ns == 0)
    return (FK_OK);
  const int sww = w.smer_stride / 4;
  u32 *n   = (u32 *) fk_slot(ctx, FK_SLOT_PF_LEN, ns * 4 + 64);
  u64 *off = (u64 *) fk_slot(ctx, FK_SLOT_PF_OFF, (ns + 1) * 8 + 64);
  const int64_t nblk = (ns + 4095) / 4096;
  u32 *bs = (u32 *) fk_slot(ctx, FK_SLOT_PF_ZC, nblk * 4 + 64);
  u64 *bo = (u64 *) fk_slot(ctx, FK_SLOT_PF_ZO, nblk * 8 + 64);
  if (n == NULL || off == NULL || bs == NULL || bo == NULL) return (FK_ENOMEM);
  hipLaunchKernelGGL(k_pf_smer_n, dim3((unsigned) ((ns + 255) / 256)), dim3(256), 0, s, (const u32 *) d_smers, ns,
                     sww, w.smer_bytes, n);
  hipLaunchKernelGGL(k_pf_blocksum, dim3((unsigned) nblk), dim3(256), 0, s, (const u32 *) n, ns, bs);
  hipLaunchKernelGGL(k_exscan_tiles, dim3(1), dim3(256), 0, s, (const u32 *) bs, nblk, bo, off + ns);
  hipLaunchKernelGGL(k_pf_blockscan, dim3((unsigned) nblk), dim3(256), 0, s, (const u32 *) n, ns, (const u64 *) bo, off);
  hipLaunchKernelGGL(k_pf_scatter, dim3((unsigned) ((ns + 255) / 256)), dim3(256), 0, s, (const u64 *) d_pos,
                     (const u64 *) off, ns, (const uint16_t *) d_in, cnts);
  FK_LAUNCH_CHECK(ctx);
  FK_HIP(ctx, hipStreamSynchronize(s));
  return (FK_OK);
}

int fkx_profile_encode_counts(fk_ctx *ctx, const void *d_bases, int64_t nbytes, int64_t *nreads_out,
                              int64_t *nprof_out, void **d_data, uint64_t **d_offs)
{ *nreads_out = 0;
  *nprof_out = 0;
  *d_data = NULL;
  *d_offs = NULL;
  if (nbytes <= 0)
    return (FK_OK);
  uint16_t *cnts = (uint16_t *) fk_slot(ctx, FK_SLOT_PF_CNT, nbytes * 2 + 64);     // what fkx_profile_scatter filled
  if (cnts == NULL) return (FK_ENOMEM);
  return fkx_profile_encode(ctx, (const uint8_t *) d_bases, nbytes, cnts, nreads_out, nprof_out, d_data, d_offs);
}

// read boundaries and codec over the per-position counts
static int fkx_profile_encode(fk_ctx *ctx, const uint8_t *bases, int64_t nbytes, const uint16_t *cnts,
                              int64_t *nreads_out, int64_t *nprof_out, void **d_data, uint64_t **d_offs)
{ hipStream_t s = ctx->stream;
  const int K = ctx->wid.kmer;

  // 3. read boundaries
  const int64_t nz = (nbytes + PF_ZCH - 1) / PF_ZCH;
  u32 *zc = (u32 *) fk_slot(ctx, FK_SLOT_PF_ZC, nz * 4 + 64);
  u64 *zo = (u64 *) fk_slot(ctx, FK_SLOT_PF_ZO, nz * 8 + 64);
  if (zc == NULL || zo == NULL) return (FK_ENOMEM);
  hipLaunchKernelGGL(k_pf_zeros<false>, dim3((unsigned) nz), dim3(256), 0, s, bases, nbytes, zc,
                     (const u64 *) NULL, (int64_t *) NULL);
  hipLaunchKernelGGL(k_exscan_tiles, dim3(1), dim3(256), 0, s, (const u32 *) zc, nz, zo, ctx->d_scratch);
  FK_LAUNCH_CHECK(ctx);
  uint8_t lastb = 0;
  FK_HIP(ctx, hipMemcpyAsync(ctx->h_scratch, ctx->d_scratch, 8, hipMemcpyDeviceToHost, s));
  FK_HIP(ctx, hipMemcpyAsync((char *) ctx->h_scratch + 8, bases + nbytes - 1, 1, hipMemcpyDeviceToHost, s));
  FK_HIP(ctx, hipStreamSynchronize(s));
  const int64_t nzero = (int64_t) ctx->h_scratch[0];
  lastb = *((uint8_t *) ctx->h_scratch + 8);
  const int64_t nreads = nzero + (lastb != 0 ? 1 : 0);
  if (getenv("FK_PF_CHECK") != NULL)
    { // harness aid (tests/fuzz_parity.py under tools/fuzz_many.sh): the read terminators counted again on the host from a
      // BLOCKING copy of the buffer, and the tile counts added up again -- which of the three disagrees when a run
      // reports another number of profiles than reads?
      std::vector<uint8_t> hb((size_t) nbytes);
      std::vector<u32> hz((size_t) nz);
      if (hipMemcpy(hb.data(), bases, (size_t) nbytes, hipMemcpyDeviceToHost) == hipSuccess
          && hipMemcpy(hz.data(), zc, (size_t) nz * 4, hipMemcpyDeviceToHost) == hipSuccess)
        { int64_t cz = 0, sz = 0, runs = 0;
          for (int64_t i = 0; i < nbytes; i++) cz += (hb[(size_t) i] == 0);
          for (int64_t i = 0; i < nz; i++) sz += hz[(size_t) i];
          char where[256] = "";
          for (int64_t i = 1; i < nbytes; i++)
            if (hb[(size_t) i] == 0 && hb[(size_t) i - 1] == 0)
              { if (runs < 8)
                  snprintf(where + strlen(where), sizeof(where) - strlen(where), " %lld(+%d)", (long long) i,
                           (int) (((uintptr_t) bases + (uintptr_t) i) & 63));
                runs += 1;
              }
          fprintf(stderr, "FK_PF_CHECK %s: total from the device %lld, tile counts add up to %lld, a blocking copy of the %lld "
                          "bytes holds %lld zeros (%lld places with two in a row:%s), last byte %d\n",
                  (cz == nzero && sz == nzero) ? "ok" : "MISMATCH", (long long) nzero, (long long) sz, (long long) nbytes,
                  (long long) cz, (long long) runs, where, (int) lastb);
        }
    }
  *nreads_out = nreads;
  if (nreads == 0)
    return (FK_OK);
  int64_t *ends = (int64_t *) fk_slot(ctx, FK_SLOT_PF_ENDS, (nreads + 1) * 8);
  if (ends == NULL) return (FK_ENOMEM);
  hipLaunchKernelGGL(k_pf_zeros<true>, dim3((unsigned) nz), dim3(256), 0, s, bases, nbytes, (u32 *) NULL,
                     (const u64 *) zo, ends);
  FK_LAUNCH_CHECK(ctx);
  if (lastb != 0)
    { // (the source is a stack variable: a blocking copy, ordered behind the kernel above)
      const int hrc = fkx_h2d_pageable(ctx, s, ends + nzero, &nbytes, 8);
      if (hrc != FK_OK) return (hrc);
    }

  // 4. codec: lengths, offsets, bytes
  u32 *lens = (u32 *) fk_slot(ctx, FK_SLOT_PF_LEN, nreads * 4 + 64);
  u64 *offs = (u64 *) fk_slot(ctx, FK_SLOT_PF_OFF, (nreads + 1) * 8 + 64);
  if (lens == NULL || offs == NULL) return (FK_ENOMEM);
  const unsigned nb = (unsigned) ((nreads + PF_ER - 1) / PF_ER);
  // exact_parts: the reference's own bytes, when the reads are the pushed ones, whole (no read cut by a block edge)
  // and their input threads are known; else the canonical stream
  bool exact = ctx->prm.exact_parts && ctx->exact_tran_set && ctx->pf_own_reads && !ctx->blocks_bad && ctx->nblocks > 0
               && ctx->wid.kmer <= 128;
  if (exact)
    { int64_t tot = 0;
      for (int64_t b = 0; b < ctx->nblocks; b++)
        { tot += ctx->blocks[b].nreads;
          if (ctx->blocks[b].rem > 0 || ctx->blocks[b].tid < 0 || ctx->blocks[b].tid >= 4096) exact = false;
        }
      if (tot != nreads) exact = false;
    }
  ExactProfArgs xa;
  const unsigned xnb = (unsigned) ((nreads + PX_THREADS - 1) / PX_THREADS);
  if (exact)
    { xa.bases = bases; xa.ends = ends; xa.nreads = nreads; xa.nbytes = nbytes;
      xa.kmer = K; xa.bc_prefix = ctx->prm.bc_prefix;
      for (int i = 0; i < 4; i++) xa.tran[i] = ctx->exact_tran[i];
      xa.pad_len = 5 + (ctx->scheme_nparts > 1 ? ctx->scheme_pad : 0);
      xa.cnts = cnts;
      xa.pan = 1024u * (u32) (ctx->scheme_nparts > 1 ? ctx->scheme_nparts : 1);
      xa.nrun = lens; xa.lens = lens; xa.offs = NULL; xa.out = NULL; xa.rid0 = NULL;
      // run numbers: per read on the device, their running sums per input thread on the host (blocks of several
      // threads interleave in HBM in push order)
      hipLaunchKernelGGL(k_pf_exact<0>, dim3(xnb), dim3(PX_THREADS), 0, s, xa);
      FK_LAUNCH_CHECK(ctx);
      std::vector<u32> nrun((size_t) nreads);
      std::vector<u64> rid0((size_t) nreads);
      { const int hrc = fkx_d2h_pageable(ctx, s, nrun.data(), lens, (size_t) nreads * 4);
        if (hrc != FK_OK) return (hrc);
      }
      { std::vector<u64> run_of_tid(4096, 0);
        int64_t r = 0;
        for (int64_t b = 0; b < ctx->nblocks; b++)
          for (int64_t i = 0; i < ctx->blocks[b].nreads; i++, r++)
            { rid0[(size_t) r] = run_of_tid[(size_t) ctx->blocks[b].tid];
              run_of_tid[(size_t) ctx->blocks[b].tid] += nrun[(size_t) r];
            }
      }
      u64 *d_rid0 = (u64 *) fk_slot(ctx, FK_SLOT_PF_RID, nreads * 8 + 64);
      if (d_rid0 == NULL) return (FK_ENOMEM);
      { const int hrc = fkx_h2d_pageable(ctx, s, d_rid0, rid0.data(), (size_t) nreads * 8);
        if (hrc != FK_OK) return (hrc);
      }
      xa.rid0 = d_rid0;
      hipLaunchKernelGGL(k_pf_exact<1>, dim3(xnb), dim3(PX_THREADS), 0, s, xa);
    }
  else
  hipLaunchKernelGGL(k_pf_encode<false>, dim3(nb), dim3(PF_ER), 0, s, (const uint16_t *) cnts, (const int64_t *) ends,
                     nreads, nbytes, K, ctx->prm.bc_prefix, lens, (const u64 *) NULL, (uint8_t *) NULL);
  { const int64_t nblk = (nreads + 4095) / 4096;
    u32 *bs = (u32 *) fk_slot(ctx, FK_SLOT_PF_ZC, nblk * 4 + 64);            // the zero counts are done with
    u64 *bo = (u64 *) fk_slot(ctx, FK_SLOT_PF_ZO, nblk * 8 + 64);
    if (bs == NULL || bo == NULL) return (FK_ENOMEM);
    hipLaunchKernelGGL(k_pf_blocksum, dim3((unsigned) nblk), dim3(256), 0, s, (const u32 *) lens, nreads, bs);
    hipLaunchKernelGGL(k_exscan_tiles, dim3(1), dim3(256), 0, s, (const u32 *) bs, nblk, bo, offs + nreads);
    hipLaunchKernelGGL(k_pf_blockscan, dim3((unsigned) nblk), dim3(256), 0, s, (const u32 *) lens, nreads,
                       (const u64 *) bo, offs);
  }
  FK_LAUNCH_CHECK(ctx);
  FK_HIP(ctx, hipMemcpyAsync(ctx->h_scratch, offs + nreads, 8, hipMemcpyDeviceToHost, s));
  FK_HIP(ctx, hipStreamSynchronize(s));
  const int64_t nprof = (int64_t) ctx->h_scratch[0];
  uint8_t *data = (uint8_t *) fk_slot(ctx, FK_SLOT_PF_OUT, nprof + 64);
  if (data == NULL) return (FK_ENOMEM);
  if (exact)
    { xa.offs = offs; xa.out = data;
      hipLaunchKernelGGL(k_pf_exact<2>, dim3(xnb), dim3(PX_THREADS), 0, s, xa);
    }
  else
  hipLaunchKernelGGL(k_pf_encode<true>, dim3(nb), dim3(PF_ER), 0, s, (const uint16_t *) cnts, (const int64_t *) ends,
                     nreads, nbytes, K, ctx->prm.bc_prefix, (u32 *) NULL, (const u64 *) offs, data);
  FK_LAUNCH_CHECK(ctx);
  *nprof_out = nprof;
  *d_data = data;
  *d_offs = (uint64_t *) offs;
  return (FK_OK);
}
