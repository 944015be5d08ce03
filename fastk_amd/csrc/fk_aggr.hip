// fk_aggr.hip -- count weighted k-mers that two hashed digit passes have grouped by 16 hash bits.
//
// Replaces, with the same results, the reference's sort-then-scan of the weighted k-mer list
// (Weighted_Kmer_Sort MSDsort.c:536-544 -> hist_kmers MSDsort.c:491-509 -> table_write_thread
// count.c:564-616).  Only the table has to be in k-mer order, and only k-mers with count >= the
// cutoff are in the table, so the W weighted records are not sorted at all: fkx_group brings all
// copies of a k-mer into one of 65,536 hash bins (two 8-bit digit passes), and here one workgroup
// per bin sums the weights of equal k-mers in a hash table that lives in LDS, updates the 0x8000-bin
// histogram (saturation and max_inst exactly as MSDsort.c:498-506) and appends the qualifying
// (k-mer, count) records to the table buffer, which the caller then sorts on KMER_BYTES.
//
// A bin whose distinct k-mers do not fit the LDS table is processed in 2, 4, ... rounds, each
// round taking the records whose next hash bits select it (the bin is re-read, mostly from L2).
#include "fk_common.h"

#define AG_THREADS 1024
#define AG_WAVES   (AG_THREADS / 64)
#define AG_HB      4096                 // LDS-private histogram bins
#define AG_LOCK    0x80000000u          // count word: 0 empty, AG_LOCK key being written, else count
#define AG_HIGH    (1u << 29)           // a count that reaches this is cut back by AG_CUT, the
#define AG_CUT     (1u << 28)           //   removed instances go straight to max_inst
#define AG_MAXR    64
#define AG_BINS    65536
#define AG_UNROLL  8                    // records per thread in a full table fill (the bin merging rule aims at 1024 x 8)
#define AG_BATCH   1                    // records per lane in a batch a wave takes at a time (1 or 2)
#define AG_P       4                    // slots one probe looks at (ag_read_slots is written for 4)
#define AG_NSCAL   16                   // scalars behind the histogram (8 results + 8 phase timers of ablation builds)
#ifdef FK_ABLATION
#define AG_T(k) do { if (tid == 0) { const u64 now_ = __builtin_readcyclecounter(); ph[k] += now_ - tlast; tlast = now_; } } while (0)
#else
#define AG_T(k) do { } while (0)
#endif

// a record as one memory operation: three dword loads / stores per lane at a 12-byte stride are three requests per
// record where one (dwordx3) does (a quarter of the replay passes' time was this)
template <int N> struct __attribute__((packed, aligned(4))) ag_rec { u32 w[N]; };

template <int KW> struct AgCfg
{ static constexpr int SLOTS = (KW <= 3) ? 8192 : 4096;
  // fill limit: once it is passed every thread may still claim AG_BATCH slots (it looks at the overflow flag once
  // per batch), and the table must never fill up
  static constexpr int LIMIT = (SLOTS * 3 / 4 < SLOTS - AG_THREADS * AG_BATCH - 64) ? SLOTS * 3 / 4
                                                                                    : SLOTS - AG_THREADS * AG_BATCH - 64;
  static constexpr size_t LDS = (size_t) SLOTS * (KW > 3 ? 32 : 16) + AG_HB * 4;
};

// position of the first record of every bin: bounds[b] = lower bound of (hash16 >= b), bounds[65536] = n
template <int KW>
__global__ __launch_bounds__(256) void k_ag_bounds(const u32 *__restrict__ recs, int64_t n, int kbytes,
                                                   u64 *__restrict__ bounds)
{ const u32 b = blockIdx.x * 256 + threadIdx.x;
  if (b > AG_BINS) return;
  if (b == AG_BINS) { bounds[b] = (u64) n; return; }
  int64_t lo = 0, hi = n;
  while (lo < hi)
    { const int64_t mid = (lo + hi) >> 1;
      u32 r[KW];
#pragma unroll
      for (int w = 0; w < KW; w++)
        r[w] = recs[mid * KW + w];
      u32 ha, hb;
      fk_rec_hash<KW>(r, kbytes, ha, hb);
      if ((hb & 0xffffu) < b) lo = mid + 1;
      else hi = mid;
    }
  bounds[b] = (u64) lo;
}

// AG_P table slots starting at `slot` as AG_P single ds_read_b128: a slot's key and count word must
// come from ONE LDS access (see k_ag_count); plain C++ loads of a uint4 may be split by the compiler
// into a 96-bit and a 32-bit read, which lets a reader pair a stale key with a published count.
template <int SLOTS>
__device__ __forceinline__ void ag_read_slots(u32 base, u32 slot, uint4 (&v)[4])
{  const u32 a0 = base + ((slot + 0) & (SLOTS - 1)) * 16u, a1 = base + ((slot + 1) & (SLOTS - 1)) * 16u;
  const u32 a2 = base + ((slot + 2) & (SLOTS - 1)) * 16u, a3 = base + ((slot + 3) & (SLOTS - 1)) * 16u;
  asm volatile("ds_read_b128 %0, %4\n\t"
               "ds_read_b128 %1, %5\n\t"
               "ds_read_b128 %2, %6\n\t"
               "ds_read_b128 %3, %7\n\t"
               "s_waitcnt lgkmcnt(0)"
               : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
               : "v"(a0), "v"(a1), "v"(a2), "v"(a3)
               : "memory");
}

// One slot (key + count word) with a single ds_write_b128 (see ag_read_slots).
__device__ __forceinline__ void ag_write_slot(u32 addr, uint4 v)
{ typedef unsigned int ag_u32x4 __attribute__((ext_vector_type(4)));
  const ag_u32x4 x = { v.x, v.y, v.z, v.w };
  asm volatile("ds_write_b128 %0, %1" : : "v"(addr), "v"(x) : "memory");
}

// The same AG_P reads plus one dword (the workgroup's overflow flag) in the same batch, so that the
// flag costs no LDS round trip of its own in front of every probe.
template <int SLOTS>
__device__ __forceinline__ void ag_read_slots_flag(u32 base, u32 slot, uint4 (&v)[4], u32 flag_addr, u32 &flag)
{  const u32 a0 = base + ((slot + 0) & (SLOTS - 1)) * 16u, a1 = base + ((slot + 1) & (SLOTS - 1)) * 16u;
  const u32 a2 = base + ((slot + 2) & (SLOTS - 1)) * 16u, a3 = base + ((slot + 3) & (SLOTS - 1)) * 16u;
  asm volatile("ds_read_b32 %4, %9\n\t"
               "ds_read_b128 %0, %5\n\t"
               "ds_read_b128 %1, %6\n\t"
               "ds_read_b128 %2, %7\n\t"
               "ds_read_b128 %3, %8\n\t"
               "s_waitcnt lgkmcnt(0)"
               : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(flag)
               : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(flag_addr)
               : "memory");
}

// The first look of two records: 2 x AG_P slots and the overflow flag in one batch of reads.
template <int SLOTS>
__device__ __forceinline__ void ag_read_slots2_flag(u32 base, u32 slota, u32 slotb, uint4 (&va)[4], uint4 (&vb)[4],
                                                    u32 flag_addr, u32 &flag)
{ const u32 a0 = base + ((slota + 0) & (SLOTS - 1)) * 16u, a1 = base + ((slota + 1) & (SLOTS - 1)) * 16u;
  const u32 a2 = base + ((slota + 2) & (SLOTS - 1)) * 16u, a3 = base + ((slota + 3) & (SLOTS - 1)) * 16u;
  const u32 b0 = base + ((slotb + 0) & (SLOTS - 1)) * 16u, b1 = base + ((slotb + 1) & (SLOTS - 1)) * 16u;
  const u32 b2 = base + ((slotb + 2) & (SLOTS - 1)) * 16u, b3 = base + ((slotb + 3) & (SLOTS - 1)) * 16u;
  asm volatile("ds_read_b32 %8, %17\n\t"
               "ds_read_b128 %0, %9\n\t"
               "ds_read_b128 %1, %10\n\t"
               "ds_read_b128 %2, %11\n\t"
               "ds_read_b128 %3, %12\n\t"
               "ds_read_b128 %4, %13\n\t"
               "ds_read_b128 %5, %14\n\t"
               "ds_read_b128 %6, %15\n\t"
               "ds_read_b128 %7, %16\n\t"
               "s_waitcnt lgkmcnt(0)"
               : "=&v"(va[0]), "=&v"(va[1]), "=&v"(va[2]), "=&v"(va[3]), "=&v"(vb[0]), "=&v"(vb[1]), "=&v"(vb[2]), "=&v"(vb[3]),
                 "=&v"(flag)
               : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(flag_addr)
               : "memory");
}

// The first of the AG_P slots just read that settles a probe: empty (kind 1), being written (2) or
// holding this key's first three dwords (3); act = AG_P, kind 0 when none does.  Written without
// control flow: a slot settles iff min(count word, key difference, count word ^ AG_LOCK) == 0, and
// the kind follows from the count word of the chosen slot alone (neither empty nor locked => the
// key matched).  The straightforward nested conditionals compile to ~45 exec-mask instructions per
// slot here; this form is ~7 VALU per slot.
template <int KW>
__device__ __forceinline__ void ag_classify(const uint4 (&v)[AG_P], const u32 *cur, int &act, u32 &kind,
                                            u32 &cact)
{ u32 a = AG_P, c = 0;
#pragma unroll
  for (int j = AG_P - 1; j >= 0; j--)
    { u32 e = v[j].x ^ cur[0];
      if (KW > 1) e |= v[j].y ^ cur[KW > 1 ? 1 : 0];
      if (KW > 2) e |= v[j].z ^ cur[KW > 2 ? 2 : 0];
      const u32  t = min(min(v[j].w, e), v[j].w ^ AG_LOCK);
      const bool h = (t == 0u);
      a = h ? (u32) j : a;
      c = h ? v[j].w : c;
    }
  u32 k = 3u;
  k = (c == AG_LOCK) ? 2u : k;
  k = (c == 0u) ? 1u : k;
  k = (a == (u32) AG_P) ? 0u : k;
  act = (int) a; kind = k; cact = c;
}

// exclusive scan over the 1024 threads of the block.  tmp: AG_WAVES u32 of LDS.
__device__ __forceinline__ u32 ag_block_exscan(u32 v, u32 *tmp, u32 *total)
{ const u32 lane = fk_lane();
  const u32 wave = threadIdx.x >> 6;
  u32 x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1)
    { u32 y = __shfl_up(x, o, 64);
      if ((int) lane >= o) x += y;
    }
  if (lane == 63) tmp[wave] = x;
  __syncthreads();
  u32 base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < AG_WAVES; w++)
    { const u32 t = tmp[w];
      if ((u32) w < wave) base += t;
      tot += t;
    }
  __syncthreads();
  *total = tot;
  return (base + x - v);
}

// scal: [0] max_inst  [1] distinct k-mers  [2] table entries  [3] failure flag  [4] bin ticket
//       [5] extra rounds taken  [6] table buffer too small (tcap records)
//
// LDS table: SLOTS entries of 16 bytes {key dword 0, 1, 2, count word}; records of 4 or 5 dwords keep
// dwords 3, 4 in a second array.  A lane's aligned 16-byte LDS access is served in one piece, so a
// single ds_read_b128 yields a consistent (key, count word) pair: count word 0 = empty, AG_LOCK =
// slot claimed with ds_cmpst, key not there yet (its creator then stores key + weight with ONE b128
// write, after the key's tail for wide records; LDS operations of a wave execute in order),
// anything else = published count.
// DEDUP: the records are super-mers (whole record = key, weight 1); every distinct record comes out
// once, followed by a dword with its multiplicity (records of KW + 1 dwords), nothing else is computed.
template <int KW, bool DEDUP>
__global__ __launch_bounds__(AG_THREADS) void k_ag_count(const u32 *__restrict__ recs,
                                                         const u64 *__restrict__ bounds, int kbytes,
                                                         int cutoff, u64 *__restrict__ hist_g,
                                                         u64 *__restrict__ scal, u32 *__restrict__ table,
                                                         int LIMIT, int variant, int gshift, u32 sat, u64 tcap)
{ constexpr int SLOTS = AgCfg<KW>::SLOTS;
  constexpr int U = AG_BATCH;
  extern __shared__ uint4 ag_lds[];
  uint4 *A     = ag_lds;                                   // [SLOTS]
  // LDS byte address of the table for the inline-asm reads (low half of the flat address)
  const u32 lds_base = (u32) (uintptr_t) ag_lds;
  uint4 *B     = ag_lds + SLOTS;                           // [SLOTS] when KW > 3
  u32   *lhist = (u32 *) (ag_lds + (KW > 3 ? 2 : 1) * SLOTS);   // [AG_HB]
  __shared__ u32 sh_claimed, sh_ovf, sh_next, sh_tmp[AG_WAVES];
  const u32 ovf_addr = (u32) (uintptr_t) &sh_ovf;
  __shared__ u64 sh_base;
  const int tid = threadIdx.x;

  for (int i = tid; i < SLOTS; i += AG_THREADS)
    A[i] = make_uint4(0, 0, 0, 0);
  for (int i = tid; i < AG_HB; i += AG_THREADS)
    lhist[i] = 0;
  if (tid == 0) { sh_claimed = 0; sh_ovf = 0; sh_next = 0; }
  u64 my_max = 0;
  u32 my_distinct = 0, my_rounds = 0;
  u32 R0 = 1;
#ifdef FK_ABLATION
  u64 ph[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, tlast = __builtin_readcyclecounter();
  u32 n_first = 0, n_loop = 0;          // wave-level: first looks (pairs), iterations of the general loop
#endif
  u32 kmask[KW];                                // key bytes of each record dword (pad and weight off)
#pragma unroll
  for (int w = 0; w < KW; w++)
    kmask[w] = (4 * w + 4 <= kbytes) ? 0xffffffffu : (4 * w < kbytes) ? ((1u << (8 * (kbytes - 4 * w))) - 1u) : 0u;
  __syncthreads();

  // groups of 2^gshift neighbouring bins (one bin when the input is large) are dealt round-robin:
  // hashing makes them equally heavy
  for (u32 bin = blockIdx.x; bin < (AG_BINS >> gshift); bin += gridDim.x)
    { const int64_t beg = (int64_t) bounds[bin << gshift], end = (int64_t) bounds[(bin + 1) << gshift];
      if (beg >= end)
        continue;
      AG_T(0);

      // A bin is taken in R0 selections (records whose next hash bits equal r0); R0 is what the
      // previous bin of this workgroup needed (all bins are alike), so that a table that is too
      // small for whole bins is not found out again bin after bin.  A selection that still does
      // not fit is halved on the spot (depth-first), exactly once.
      bool bin_ovf = false, failed = false;
      u32  bin_fill = 0;
      for (u32 r0 = 0; r0 < R0 && !failed; r0++)
      { u32 R = R0, r = r0;
      for (;;)
        { // ---- insert every record of the bin that this round selects
          u64 round_max = 0;
          // The waves take batches of 64 x U records from a counter in LDS instead of fixed shares: the time a
          // batch takes varies (probe chains, lost races), and with fixed shares the workgroup waited at the barrier
          // below for its slowest wave 44 % of the time.  The next batch is loaded while this one is inserted.
          const u32 lane = fk_lane();
          u32 rec[U][KW], nrec[U][KW];
          int64_t base, nbase;
          { u32 g = 0;
            if (lane == 0) g = atomicAdd(&sh_next, 1u);
            g = (u32) __builtin_amdgcn_readfirstlane((int) g);
            base = beg + (int64_t) g * (64 * U);
            if (base < end)
              {
#pragma unroll
                for (int u = 0; u < U; u++)
                  { const int64_t i = base + u * 64 + lane;
                    const int64_t j = (i < end) ? i : beg;
                    const ag_rec<KW> rr = *(const ag_rec<KW> *) (recs + j * KW);
#pragma unroll
                    for (int w = 0; w < KW; w++)
                      rec[u][w] = rr.w[w];
                  }
              }
          }
          while (base < end)
            { if (*(volatile u32 *) &sh_ovf)
                break;
              { u32 g = 0;
                if (lane == 0) g = atomicAdd(&sh_next, 1u);
                g = (u32) __builtin_amdgcn_readfirstlane((int) g);
                nbase = beg + (int64_t) g * (64 * U);
                if (nbase < end)
                  {
#pragma unroll
                    for (int u = 0; u < U; u++)
                      { const int64_t i = nbase + u * 64 + lane;
                        const int64_t j = (i < end) ? i : beg;
                        const ag_rec<KW> rr = *(const ag_rec<KW> *) (recs + j * KW);
#pragma unroll
                        for (int w = 0; w < KW; w++)
                          nrec[u][w] = rr.w[w];
                      }
                  }
              }
              u32 slot0[U];
              u32 pend = 0;
#pragma unroll
              for (int u = 0; u < U; u++)
                { const int64_t i = base + u * 64 + lane;
                  u32 ha, hb;
                  fk_rec_hash<KW>(rec[u], kbytes, ha, hb);
                  slot0[u] = ha & (SLOTS - 1);
                  if (i < end && ((hb >> 16) & (R - 1)) == r)
                    pend |= (1u << u);
                }
              const bool skip_insert = (variant & 1) != 0;
              if (skip_insert)
                { round_max += pend + slot0[0];
                  pend = 0;
                }
#ifdef FK_ABLATION
              if (tid == 0 && slot0[0] != 0xffffffffu) AG_T(1);
#endif

              // First look for the two records of the batch TOGETHER: the reads of both probes in one LDS round
              // trip, then both claims in flight at once (a wave's LDS operations complete in order, so when the
              // two records of a lane want the same empty slot the second claim sees the first one's lock and
              // goes to the general loop).  Four waves per SIMD cannot hide three dependent LDS round trips per
              // record; two records per lane halve them.  A probe looks at AG_P consecutive slots at once;
              // straight-line for the whole wave (nested divergent branches cost more scalar instructions here
              // than the probes cost vector ones): most records hit their k-mer or claim an empty slot right away.
              u32  cu_[U][KW], wg_[U], sl_[U], sx[U], kind[U];
              bool dn_[U], todo[U], bmiss[U], created[U];
#pragma unroll
              for (int u = 0; u < U; u++)
                {
#pragma unroll
                  for (int w = 0; w < KW; w++)
                    cu_[u][w] = rec[u][w] & kmask[w];
                  wg_[u]  = DEDUP ? 1u : (rec[u][KW - 1] >> 16);
                  sl_[u] = slot0[u];
                }
#ifdef FK_ABLATION
              if (fk_lane() == 0) n_first += 1;
#endif
              { // a thread claims <= U slots per look at the overflow flag (it arrives with the slots: one LDS
                // round trip), which LIMIT leaves room for
                uint4 v0[U][AG_P];
                u32   ovf_now;
                if (U == 2)
                  ag_read_slots2_flag<SLOTS>(lds_base, sl_[0], sl_[U - 1], v0[0], v0[U - 1], ovf_addr, ovf_now);
                else
                  ag_read_slots_flag<SLOTS>(lds_base, sl_[0], v0[0], ovf_addr, ovf_now);
                bool hit[U];
#pragma unroll
                for (int u = 0; u < U; u++)
                  { int act;
                    u32 cact;
                    ag_classify<KW>(v0[u], cu_[u], act, kind[u], cact);
                    sx[u]   = (sl_[u] + (u32) act) & (SLOTS - 1);
                    todo[u] = ((pend >> u) & 1u) && ovf_now == 0;
                    dn_[u] = !todo[u];
                    hit[u]  = todo[u] && kind[u] == 3u && cact < (AG_HIGH >> 1);
                    bmiss[u] = false;
                    created[u] = false;
                  }
                if (KW > 3)
                  {
#pragma unroll
                    for (int u = 0; u < U; u++)
                      if (hit[u])
                        { const uint4 b = B[sx[u]];
                          bool same = (b.x == cu_[u][KW > 3 ? 3 : 0]);
                          if (KW > 4) same = same && (b.y == cu_[u][KW > 4 ? 4 : 0]);
                          if (KW > 5) same = same && (b.z == cu_[u][KW > 5 ? 5 : 0]);
                          if (KW > 6) same = same && (b.w == cu_[u][KW > 6 ? 6 : 0]);
                          bmiss[u] = !same;
                          hit[u] = same;
                        }
                  }
#pragma unroll
                for (int u = 0; u < U; u++)
                  if (hit[u])
                    { atomicAdd(&A[sx[u]].w, wg_[u]);
                      dn_[u] = true;
                    }
                u32 got[U];
#pragma unroll
                for (int u = 0; u < U; u++)
                  { got[u] = 1u;
                    if (todo[u] && kind[u] == 1u)
                      got[u] = atomicCAS(&A[sx[u]].w, 0u, AG_LOCK);
                  }
#pragma unroll
                for (int u = 0; u < U; u++)
                  if (got[u] == 0u)
                    { // key and weight appear together: the (wider) key's tail first, then ONE 16-byte write of
                      // key + count word -- LDS serves a wave's operations in order and an aligned 16-byte
                      // access in one piece, so no reader can pair this count with another key
                      if (KW > 3)
                        { B[sx[u]] = make_uint4(cu_[u][KW > 3 ? 3 : 0], KW > 4 ? cu_[u][KW > 4 ? 4 : 0] : 0u,
                                                KW > 5 ? cu_[u][KW > 5 ? 5 : 0] : 0u, KW > 6 ? cu_[u][KW > 6 ? 6 : 0] : 0u);
                          asm volatile("" ::: "memory");
                        }
                      ag_write_slot(lds_base + sx[u] * 16u, make_uint4(cu_[u][0], KW > 1 ? cu_[u][KW > 1 ? 1 : 0] : 0u,
                                                                        KW > 2 ? cu_[u][KW > 2 ? 2 : 0] : 0u, wg_[u]));
                      created[u] = true;
                      dn_[u] = true;
                    }
                u32 kcl = 0;
                u64 cany = 0;
#pragma unroll
                for (int u = 0; u < U; u++)
                  { const u64 cm = __ballot(created[u]);
                    kcl  += (u32) __popcll(cm);
                    cany |= cm;
                  }
                if (cany != 0ull && fk_lane() == (u32) (__ffsll((long long) cany) - 1))
                  { if (atomicAdd(&sh_claimed, kcl) + kcl > (u32) LIMIT)
                      sh_ovf = 1;
                  }
                // whoever is left (lost a race, met a slot being written, no hit in AG_P slots,
                // wide keys, very large counts) goes through the general loop below
#pragma unroll
                for (int u = 0; u < U; u++)
                  if (!dn_[u])
                    sl_[u] = (kind[u] == 1u || kind[u] == 2u) ? sx[u] : (kind[u] == 0u) ? ((sl_[u] + AG_P) & (SLOTS - 1))
                            : bmiss[u] ? ((sx[u] + 1) & (SLOTS - 1)) : sl_[u];
              }
#pragma unroll
              for (int u = 0; u < U; u++)
                { bool done = dn_[u];
                  u32  slot = sl_[u];
                  u32 (&cur)[KW] = cu_[u];
                  const u32 wgt = wg_[u];
                  while (!done)
                    {
#ifdef FK_ABLATION
                      if (fk_lane() == (u32) (__ffsll((long long) __ballot(1)) - 1)) n_loop += 1;
#endif
                      uint4 v[AG_P];
                      ag_read_slots<SLOTS>(lds_base, slot, v);
                      // first slot that is empty (1), being written (2) or holds this k-mer (3)
                      int act;
                      u32 kind, cact;
                      ag_classify<KW>(v, cur, act, kind, cact);
                      const u32 s = (slot + (u32) act) & (SLOTS - 1);
                      if (KW <= 3 && kind == 3u && cact < (AG_HIGH >> 1))
                        { atomicAdd(&A[s].w, wgt);                   // the common case: no return value needed
                          done = true;
                        }
                      else
                        { bool created = false;
                          if (kind == 3u)
                            { bool same = true;
                              if (KW > 3)
                                { const uint4 b = B[s];
                                  same = (b.x == cur[KW > 3 ? 3 : 0]);
                                  if (KW > 4) same = same && (b.y == cur[KW > 4 ? 4 : 0]);
                                  if (KW > 5) same = same && (b.z == cur[KW > 5 ? 5 : 0]);
                                  if (KW > 6) same = same && (b.w == cur[KW > 6 ? 6 : 0]);
                                }
                              if (!same)
                                slot = (s + 1) & (SLOTS - 1);
                              else
                                { const u32 old = atomicAdd(&A[s].w, wgt);
                                  if (old < AG_HIGH && old + wgt >= AG_HIGH)
                                    { atomicSub(&A[s].w, AG_CUT);    // stays far above 0x7fff: still saturated
                                      round_max += AG_CUT;
                                    }
                                  done = true;
                                }
                            }
                          else if (kind == 1u)
                            { if (atomicCAS(&A[s].w, 0u, AG_LOCK) == 0u)
                                { // key and weight appear together: the (wider) key's tail first, then ONE 16-byte write of
                                  // key + count word -- LDS serves a wave's operations in order and an aligned 16-byte
                                  // access in one piece, so no reader can pair this count with another key
                                  if (KW > 3)
                                    { B[s] = make_uint4(cur[KW > 3 ? 3 : 0], KW > 4 ? cur[KW > 4 ? 4 : 0] : 0u,
                                                        KW > 5 ? cur[KW > 5 ? 5 : 0] : 0u, KW > 6 ? cur[KW > 6 ? 6 : 0] : 0u);
                                      asm volatile("" ::: "memory");
                                    }
                                  ag_write_slot(lds_base + s * 16u, make_uint4(cur[0], KW > 1 ? cur[KW > 1 ? 1 : 0] : 0u,
                                                                               KW > 2 ? cur[KW > 2 ? 2 : 0] : 0u, wgt));
                                  created = true;
                                  done = true;
                                }
                              else
                                slot = s;                            // taken in between: look at it again
                            }
                          else if (kind == 2u)
                            slot = s;                                // its key is being written: look again
                          else
                            slot = (slot + AG_P) & (SLOTS - 1);
                          const u64 cm = __ballot(created);
                          if (cm != 0ull && fk_lane() == (u32) (__ffsll((long long) cm) - 1))
                            { const u32 k = (u32) __popcll(cm);
                              if (atomicAdd(&sh_claimed, k) + k > (u32) LIMIT)
                                sh_ovf = 1;
                            }
                        }
                    }
                }
              // the batch loaded meanwhile becomes the current one
              base = nbase;
#pragma unroll
              for (int u = 0; u < U; u++)
#pragma unroll
                for (int w = 0; w < KW; w++)
                  rec[u][w] = nrec[u][w];
            }
          AG_T(2);
          __syncthreads();
          AG_T(3);
          const bool ovf = (sh_ovf != 0);
          const u32  fill = sh_claimed;
          __syncthreads();
          if (ovf)
            { // more distinct k-mers than the table takes: halve the selection and start it again
              for (int i = tid; i < SLOTS; i += AG_THREADS)
                A[i].w = 0;
              if (tid == 0) { sh_claimed = 0; sh_ovf = 0; sh_next = 0; }
              my_rounds += (tid == 0);
              bin_ovf = true;
              __syncthreads();
              if (R >= AG_MAXR)
                { if (tid == 0)
                    atomicAdd(&scal[3], 1ull);
                  failed = true;
                  break;
                }
              R <<= 1;
              continue;
            }
          bin_fill = max(bin_fill, fill);

          // ---- emit: histogram, totals, table entries; the table is left empty
          my_max += round_max;
          // (all count words first: eight independent LDS reads instead of eight round trips; the k-mers seen once
          // or twice -- four out of five on read sets with sequencing errors -- are counted per wave with a ballot:
          // one LDS atomic per lane on the same two histogram bins serialised the whole sweep)
          u32 c[SLOTS / AG_THREADS], vv[SLOTS / AG_THREADS];
          u32 nq = 0, n1 = 0, n2 = 0;
#pragma unroll
          for (int j = 0; j < SLOTS / AG_THREADS; j++)
            vv[j] = A[j * AG_THREADS + tid].w;
#pragma unroll
          for (int j = 0; j < SLOTS / AG_THREADS; j++)
            { const int slot = j * AG_THREADS + tid;
              const u32 v = vv[j];
              c[j] = 0;
              if (!DEDUP)
                { n1 += (u32) __popcll(__ballot(v == 1u));
                  n2 += (u32) __popcll(__ballot(v == 2u));
                }
              if (v != 0)
                { A[slot].w = 0;
                  my_distinct += 1;
                  u32 cc = v;
                  if (DEDUP)
                    { nq += 1;
                      c[j] = v;
                      continue;
                    }
                  if (v >= sat)                                // sat = 0x7fff, MSDsort.c:498-506
                    my_max += v;
                  if (v >= 0x7fffu)
                    cc = 0x7fffu;
                  if ((variant & 2) || cc <= 2u) ;
                  else if (cc < AG_HB) atomicAdd(&lhist[cc], 1u);
                  else            atomicAdd(&hist_g[cc], 1ull);
                  if (cutoff > 0 && (int) cc >= cutoff)
                    { nq += 1;
                      c[j] = cc;
                    }
                }
            }
          if (!DEDUP && !(variant & 2) && fk_lane() == 0)
            { if (n1 != 0) atomicAdd(&lhist[1], n1);
              if (n2 != 0) atomicAdd(&lhist[2], n2);
            }
          AG_T(4);
          if ((DEDUP || cutoff > 0) && !(variant & 4))
            { u32 tot;
              const u32 off = ag_block_exscan(nq, sh_tmp, &tot);
              AG_T(5);
              if (tid == 0 && tot > 0)
                sh_base = atomicAdd(&scal[2], (u64) tot);
              __syncthreads();
              AG_T(6);
              if (tot > 0 && sh_base + tot > tcap)
                { if (tid == 0)                       // the table buffer is full (direct append to a union buffer)
                    atomicAdd(&scal[6], 1ull);
                }
              else if (tot > 0)
                { u64 o = sh_base + off;
#pragma unroll
                  for (int j = 0; j < SLOTS / AG_THREADS; j++)
                    if (c[j] != 0)
                      { const int slot = j * AG_THREADS + tid;
                        const uint4 a = A[slot];
                        u32 kd[7] = { a.x, a.y, a.z, 0u, 0u, 0u, 0u };
                        if (KW > 3)
                          { const uint4 b = B[slot];
                            kd[3] = b.x; kd[4] = b.y; kd[5] = b.z; kd[6] = b.w;
                          }
                        if (DEDUP)
                          { ag_rec<KW + 1> ro;
#pragma unroll
                            for (int w = 0; w < KW; w++)
                              ro.w[w] = kd[w];
                            ro.w[KW] = c[j];
                            *(ag_rec<KW + 1> *) (table + o * (KW + 1)) = ro;
                          }
                        else
                          { ag_rec<KW> ro;
#pragma unroll
                            for (int w = 0; w < KW - 1; w++)
                              ro.w[w] = kd[w];
                            ro.w[KW - 1] = kd[KW - 1] | (c[j] << 16);
                            *(ag_rec<KW> *) (table + o * KW) = ro;
                          }
                        o += 1;
                      }
                }
            }
          if (tid == 0) { sh_claimed = 0; sh_next = 0; }
          __syncthreads();
          AG_T(7);

          // ---- next selection below (R0, r0): sibling, or up
          while (R > R0 && r >= (R >> 1))
            { r -= (R >> 1);
              R >>= 1;
            }
          if (R == R0)
            break;
          r += (R >> 1);
        }
      }
      if (bin_ovf)
        R0 = min(R0 << 1, (u32) AG_MAXR);
      else if (R0 > 1 && bin_fill * 9 < (u32) LIMIT * 4)
        R0 >>= 1;
    }

  // flush the private histogram and the per-thread totals
  __syncthreads();
  for (int i = tid; i < AG_HB; i += AG_THREADS)
    if (lhist[i] != 0)
      atomicAdd(&hist_g[i], (u64) lhist[i]);
  u64 d = my_distinct, rd = my_rounds;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
    { my_max += __shfl_down(my_max, o, 64);
      d      += __shfl_down(d, o, 64);
      rd     += __shfl_down(rd, o, 64);
    }
  if (fk_lane() == 0)
    { if (my_max) atomicAdd(&scal[0], my_max);
      if (d)      atomicAdd(&scal[1], d);
      if (rd)     atomicAdd(&scal[5], rd);
    }
#ifdef FK_ABLATION
  if (tid == 0)
    for (int k = 0; k < 6; k++)
      atomicAdd(&scal[8 + k], ph[k]);
  if (tid == 0)
    atomicAdd(&scal[8 + 6], ph[6] + ph[7]);
  { u64 nl = n_loop, nf = n_first;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
      { nl += __shfl_down(nl, o, 64);
        nf += __shfl_down(nf, o, 64);
      }
    if (fk_lane() == 0)
      atomicAdd(&scal[8 + 7], (nf << 32) | nl);
  }
#endif
}

template <int KW>
static int aggr_t(fk_ctx *ctx, const void *d_grouped, int64_t n, int cutoff, int64_t *hist,
                  int64_t *max_inst, int64_t *ndistinct, void *d_table, int64_t cap, int64_t *ntable)
{ hipStream_t s = ctx->stream;
  if (ntable) *ntable = 0;
  if (ndistinct) *ndistinct = 0;
  if (n == 0)
    return (FK_OK);
  if (cutoff > 0 && (d_table == NULL || cap <= 0))
    { fk_set_error(ctx, "aggregate: no table buffer");
      return (FK_EINVAL);
    }
  u64 *d_bounds = (u64 *) fk_slot(ctx, FK_SLOT_AG_BOUNDS, (AG_BINS + 1) * 8);
  u64 *d_hist   = (u64 *) fk_slot(ctx, FK_SLOT_CT_HIST, (FK_HIST_BINS + AG_NSCAL) * 8);
  if (d_bounds == NULL || d_hist == NULL)
    return (FK_ENOMEM);
  u64 *d_scal = d_hist + FK_HIST_BINS;
  static bool attr_set[8] = { false };
  const size_t lds = AgCfg<KW>::LDS;
  if (!attr_set[KW])
    { auto kern = k_ag_count<KW, false>;
      FK_HIP(ctx, hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
      attr_set[KW] = true;
    }
  FK_HIP(ctx, hipMemsetAsync(d_hist, 0, (FK_HIST_BINS + AG_NSCAL) * 8, s));
  hipLaunchKernelGGL(k_ag_bounds<KW>, dim3(AG_BINS / 256 + 1), dim3(256), 0, s, (const u32 *) d_grouped, n,
                     ctx->wid.kmer_bytes, d_bounds);
  const int cus = ctx->num_cus > 0 ? ctx->num_cus : 256;
  // Neighbouring bins are merged while a table fill stays within one batch of 1024 x 8 records: small inputs
  // get full batches, and a fill never holds more records than that by choice -- a fill whose distinct k-mers
  // pass LIMIT is done over in two selections (measured at configs[2], 7,080 records and 4,570 distinct k-mers
  // per bin: one bin per fill 323 ms per step, two bins per fill -> two selections each 459 ms, one bin in two
  // selections 402 ms).
  int gshift = 0;
  while (ctx->dbg_aggr_limit <= 0 && gshift < 16 && ((n >> (16 - gshift)) << 1) <= AG_THREADS * AG_UNROLL)
    gshift += 1;
  if (ctx->dbg_aggr_gshift > 0)
    gshift = ctx->dbg_aggr_gshift - 1;
  const int limit = (ctx->dbg_aggr_limit > 0 && ctx->dbg_aggr_limit < AgCfg<KW>::LIMIT) ? ctx->dbg_aggr_limit
                                                                                        : AgCfg<KW>::LIMIT;
  hipLaunchKernelGGL((k_ag_count<KW, false>), dim3((unsigned) cus), dim3(AG_THREADS), lds, s, (const u32 *) d_grouped,
                     (const u64 *) d_bounds, ctx->wid.kmer_bytes, cutoff, d_hist, d_scal, (u32 *) d_table, limit, ctx->dbg_aggr_variant, gshift,
                     (u32) (ctx->aggr_sat > 0 ? ctx->aggr_sat : 0x7fff), (u64) cap);
  FK_LAUNCH_CHECK(ctx);
  u64 *h = ctx->h_scratch;                       // pinned, 64 KB + 64 KB: the histogram needs 256 KB
  u64 *hh = (u64 *) malloc((FK_HIST_BINS + AG_NSCAL) * 8);
  if (hh == NULL) return (FK_ENOMEM);
  (void) h;
  if (hipMemcpyAsync(hh, d_hist, (FK_HIST_BINS + AG_NSCAL) * 8, hipMemcpyDeviceToHost, s) != hipSuccess
      || hipStreamSynchronize(s) != hipSuccess)
    { free(hh);
      fk_set_error(ctx, "aggregate: HIP failure: %s", hipGetErrorString(hipGetLastError()));
      return (FK_EHIP);
    }
  if (hh[FK_HIST_BINS + 3] != 0)
    { free(hh);
      return (FK_ESTATE);                          // a bin did not fit even in AG_MAXR rounds
    }
  if (hh[FK_HIST_BINS + 6] != 0)
    { free(hh);
      return (FKX_TABLE_FULL);                     // fewer than `cap` records of room: nothing was accumulated
    }
  for (int i = 1; i < FK_HIST_BINS; i++)
    hist[i] += (int64_t) hh[i];
  *max_inst += (int64_t) hh[FK_HIST_BINS + 0];
  if (ndistinct) *ndistinct = (int64_t) hh[FK_HIST_BINS + 1];
  if (ntable) *ntable = (int64_t) hh[FK_HIST_BINS + 2];
  ctx->aggr_extra_rounds = (int64_t) hh[FK_HIST_BINS + 5];
#ifdef FK_ABLATION
  if (getenv("FK_AG_TIMING") != NULL)
    { fprintf(stderr, "ag phases (cycles of thread 0, all workgroups) n=%lld:", (long long) n);
      for (int k = 0; k < 8; k++)
        fprintf(stderr, " %llu", (unsigned long long) hh[FK_HIST_BINS + 8 + k]);
      fprintf(stderr, "  first looks %llu loop iterations %llu\n", (unsigned long long) (hh[FK_HIST_BINS + 15] >> 32),
              (unsigned long long) (hh[FK_HIST_BINS + 15] & 0xffffffffull));
    }
#endif
  free(hh);
  return (FK_OK);
}

/* d_grouped: n weighted k-mer records ordered by the low 16 bits of fk_rec_hash's b word (the
   result of fkx_group(..., key_bytes = KMER_BYTES, npasses = 2)).  Adds the histogram of their
   counts into hist[1..0x7fff] and the instances of saturated k-mers into *max_inst; with cutoff > 0
   writes the (k-mer, count) records with count >= cutoff to d_table IN NO PARTICULAR ORDER.
   FK_ESTATE: some bin holds more distinct k-mers than AG_MAXR rounds of the LDS table take. */
int fkx_aggregate(fk_ctx *ctx, const void *d_grouped, int64_t n, int cutoff, int64_t *hist,
                  int64_t *max_inst, int64_t *ndistinct, void *d_table, int64_t cap, int64_t *ntable)
{ switch (ctx->wid.kmer_stride >> 2)
  { case 1: return aggr_t<1>(ctx, d_grouped, n, cutoff, hist, max_inst, ndistinct, d_table, cap, ntable);
    case 2: return aggr_t<2>(ctx, d_grouped, n, cutoff, hist, max_inst, ndistinct, d_table, cap, ntable);
    case 3: return aggr_t<3>(ctx, d_grouped, n, cutoff, hist, max_inst, ndistinct, d_table, cap, ntable);
    case 4: return aggr_t<4>(ctx, d_grouped, n, cutoff, hist, max_inst, ndistinct, d_table, cap, ntable);
    case 5: return aggr_t<5>(ctx, d_grouped, n, cutoff, hist, max_inst, ndistinct, d_table, cap, ntable);
    default:
      fk_set_error(ctx, "k-mer stride %d not built", ctx->wid.kmer_stride);
      return (FK_EUNSUPPORTED);
  }
}

// ---------------------------------------------------------------------------------------------
// Super-mer de-duplication with the same kernel: n records of KW dwords, grouped by 16 hash bits of
// the whole record -> every distinct record once, followed by its multiplicity (KW + 1 dwords each,
// in no particular order).  Replaces two of the four grouping passes and the run detection of the
// expansion (count.c:421-426).
template <int KW>
static int dedup_t(fk_ctx *ctx, const void *d_grouped, int64_t n, void *d_out, int64_t cap, int64_t *nout)
{ hipStream_t s = ctx->stream;
  *nout = 0;
  if (n == 0)
    return (FK_OK);
  if (d_out == NULL || cap < n)
    { fk_set_error(ctx, "dedup: the output buffer must take as many records as the input (%lld)", (long long) n);
      return (FK_EINVAL);
    }
  u64 *d_bounds = (u64 *) fk_slot(ctx, FK_SLOT_AG_BOUNDS, (AG_BINS + 1) * 8);
  u64 *d_hist   = (u64 *) fk_slot(ctx, FK_SLOT_CT_HIST, (FK_HIST_BINS + AG_NSCAL) * 8);
  if (d_bounds == NULL || d_hist == NULL)
    return (FK_ENOMEM);
  u64 *d_scal = d_hist + FK_HIST_BINS;
  static bool attr_set = false;
  const size_t lds = AgCfg<KW>::LDS;
  if (!attr_set)
    { auto kern = k_ag_count<KW, true>;
      FK_HIP(ctx, hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
      attr_set = true;
    }
  FK_HIP(ctx, hipMemsetAsync(d_scal, 0, 8 * 8, s));
  hipLaunchKernelGGL(k_ag_bounds<KW>, dim3(AG_BINS / 256 + 1), dim3(256), 0, s, (const u32 *) d_grouped, n,
                     KW * 4, d_bounds);
  const int cus = ctx->num_cus > 0 ? ctx->num_cus : 256;
  int gshift = 0;
  while (gshift < 16 && (n >> (16 - gshift)) < 6000)         // most of a batch of records per table fill
    gshift += 1;
  hipLaunchKernelGGL((k_ag_count<KW, true>), dim3((unsigned) cus), dim3(AG_THREADS), lds, s, (const u32 *) d_grouped,
                     (const u64 *) d_bounds, KW * 4, 1, d_hist, d_scal, (u32 *) d_out, AgCfg<KW>::LIMIT, 0,
                     gshift, 0x7fffu, (u64) cap);
  FK_LAUNCH_CHECK(ctx);
  FK_HIP(ctx, hipMemcpyAsync(ctx->h_scratch + 4100, d_scal, 8 * 8, hipMemcpyDeviceToHost, s));
  FK_HIP(ctx, hipStreamSynchronize(s));
  if (ctx->h_scratch[4100 + 3] != 0)
    return (FK_ESTATE);
  *nout = (int64_t) ctx->h_scratch[4100 + 2];
  return (FK_OK);
}

int fkx_dedup_supermers(fk_ctx *ctx, const void *d_grouped, int64_t n, void *d_out, int64_t cap, int64_t *nout)
{ switch (ctx->wid.smer_stride >> 2)
  { case 2: return dedup_t<2>(ctx, d_grouped, n, d_out, cap, nout);
    case 3: return dedup_t<3>(ctx, d_grouped, n, d_out, cap, nout);
    case 4: return dedup_t<4>(ctx, d_grouped, n, d_out, cap, nout);
    case 5: return dedup_t<5>(ctx, d_grouped, n, d_out, cap, nout);
    case 6: return dedup_t<6>(ctx, d_grouped, n, d_out, cap, nout);
    case 7: return dedup_t<7>(ctx, d_grouped, n, d_out, cap, nout);
    default: return (FK_EUNSUPPORTED);              // the caller keeps the four-pass grouping
  }
}
