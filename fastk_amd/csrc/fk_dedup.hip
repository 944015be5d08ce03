// fk_dedup.hip -- every distinct super-mer record once, with its multiplicity, from records that two hashed
// digit passes have grouped by 16 hash bits.
//
// Replaces two of the four grouping passes over the super-mer list and the run detection of the expansion
// (Supermer_Sort MSDsort.c:458-489 -> the run-length pass of count.c:421-426).  On read sets with coverage most
// records are copies (D/S = 0.17 on 50x HiFi-shaped data), which is the case an LDS HASH TABLE is good at: a copy
// costs one look at four slots and one LDS add.  (The weighted k-mers, where most records are new, are summed by the
// counting sort of fk_aggr.hip; that kernel ran 1.6x slower than this one on the super-mers.)
//
// One persistent workgroup per CU takes hash bins in turn; a bin whose distinct records do not fit the LDS table
// is processed in 2, 4, ... rounds, each taking the records whose next hash bits select it.
#include "fk_common.h"

#define DD_THREADS 1024
#define DD_WAVES   (DD_THREADS / 64)
#define DD_LOCK    0x80000000u          // count word: 0 empty, DD_LOCK key being written, else count
#define DD_MAXR    64
#define DD_BINS    65536
#define DD_UNROLL  8                    // records per thread in a full table fill (the bin merging rule aims at 1024 x 8)
#define DD_BATCH   1                    // records per lane in a batch a wave takes at a time (1 or 2)
#define DD_P       4                    // slots one probe looks at (dd_read_slots is written for 4)

// a record as one memory operation: three dword loads / stores per lane at a 12-byte stride are three requests per
// record where one (dwordx3) does (a quarter of the replay passes' time was this)
template <int N> struct __attribute__((packed, aligned(4))) dd_rec { u32 w[N]; };

template <int KW> struct DdCfg
{ static constexpr int SLOTS = (KW <= 3) ? 8192 : 4096;
  // fill limit: once it is passed every thread may still claim DD_BATCH slots (it looks at the overflow flag once
  // per batch), and the table must never fill up
  static constexpr int LIMIT = (SLOTS * 3 / 4 < SLOTS - DD_THREADS * DD_BATCH - 64) ? SLOTS * 3 / 4
                                                                                    : SLOTS - DD_THREADS * DD_BATCH - 64;
  static constexpr size_t LDS = (size_t) SLOTS * (KW > 3 ? 32 : 16) ;
};

// position of the first record of every bin: bounds[b] = lower bound of (hash16 >= b), bounds[65536] = n
template <int KW>
__global__ __launch_bounds__(256) void k_dd_bounds(const u32 *__restrict__ recs, int64_t n, int kbytes,
                                                   u64 *__restrict__ bounds)
{ const u32 b = blockIdx.x * 256 + threadIdx.x;
  if (b > DD_BINS) return;
  if (b == DD_BINS) { bounds[b] = (u64) n; return; }
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

#ifdef FK_HOST_EMU
// (tests/csrc/hip_emu.h: the four accesses below as plain C++ for the CPU tests; `base` / `addr` are byte offsets into the
//  workgroup's dynamic LDS there, the overflow flag is reached through a pointer the kernel leaves)
static u32 *emu_dd_flag;
template <int SLOTS>
static inline void dd_read_slots(u32 base, u32 slot, uint4 (&v)[4])
{ const uint4 *t = (const uint4 *) ((const char *) emu_g->dyn_lds + base);
  for (int i = 0; i < 4; i++) v[i] = t[(slot + i) & (SLOTS - 1)];
}
static inline void dd_write_slot(u32 addr, uint4 v) { *(uint4 *) ((char *) emu_g->dyn_lds + addr) = v; }
template <int SLOTS>
static inline void dd_read_slots_flag(u32 base, u32 slot, uint4 (&v)[4], u32 fldd_addr, u32 &flag)
{ (void) fldd_addr; flag = *emu_dd_flag; dd_read_slots<SLOTS>(base, slot, v); }
template <int SLOTS>
static inline void dd_read_slots2_flag(u32 base, u32 slota, u32 slotb, uint4 (&va)[4], uint4 (&vb)[4], u32 fldd_addr, u32 &flag)
{ (void) fldd_addr; flag = *emu_dd_flag; dd_read_slots<SLOTS>(base, slota, va); dd_read_slots<SLOTS>(base, slotb, vb); }
#else
// DD_P table slots starting at `slot` as DD_P single ds_read_b128: a slot's key and count word must
// come from ONE LDS access (see k_dd_table); plain C++ loads of a uint4 may be split by the compiler
// into a 96-bit and a 32-bit read, which lets a reader pair a stale key with a published count.
template <int SLOTS>
__device__ __forceinline__ void dd_read_slots(u32 base, u32 slot, uint4 (&v)[4])
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

// One slot (key + count word) with a single ds_write_b128 (see dd_read_slots).
__device__ __forceinline__ void dd_write_slot(u32 addr, uint4 v)
{ typedef unsigned int dd_u32x4 __attribute__((ext_vector_type(4)));
  const dd_u32x4 x = { v.x, v.y, v.z, v.w };
  asm volatile("ds_write_b128 %0, %1" : : "v"(addr), "v"(x) : "memory");
}

// The same DD_P reads plus one dword (the workgroup's overflow flag) in the same batch, so that the
// flag costs no LDS round trip of its own in front of every probe.
template <int SLOTS>
__device__ __forceinline__ void dd_read_slots_flag(u32 base, u32 slot, uint4 (&v)[4], u32 fldd_addr, u32 &flag)
{  const u32 a0 = base + ((slot + 0) & (SLOTS - 1)) * 16u, a1 = base + ((slot + 1) & (SLOTS - 1)) * 16u;
  const u32 a2 = base + ((slot + 2) & (SLOTS - 1)) * 16u, a3 = base + ((slot + 3) & (SLOTS - 1)) * 16u;
  asm volatile("ds_read_b32 %4, %9\n\t"
               "ds_read_b128 %0, %5\n\t"
               "ds_read_b128 %1, %6\n\t"
               "ds_read_b128 %2, %7\n\t"
               "ds_read_b128 %3, %8\n\t"
               "s_waitcnt lgkmcnt(0)"
               : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(flag)
               : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(fldd_addr)
               : "memory");
}

// The first look of two records: 2 x DD_P slots and the overflow flag in one batch of reads.
template <int SLOTS>
__device__ __forceinline__ void dd_read_slots2_flag(u32 base, u32 slota, u32 slotb, uint4 (&va)[4], uint4 (&vb)[4],
                                                    u32 fldd_addr, u32 &flag)
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
               : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(fldd_addr)
               : "memory");
}

#endif   // FK_HOST_EMU

// The first of the DD_P slots just read that settles a probe: empty (kind 1), being written (2) or
// holding this key's first three dwords (3); act = DD_P, kind 0 when none does.  Written without
// control flow: a slot settles iff min(count word, key difference, count word ^ DD_LOCK) == 0, and
// the kind follows from the count word of the chosen slot alone (neither empty nor locked => the
// key matched).  The straightforward nested conditionals compile to ~45 exec-mask instructions per
// slot here; this form is ~7 VALU per slot.
template <int KW>
__device__ __forceinline__ void dd_classify(const uint4 (&v)[DD_P], const u32 *cur, int &act, u32 &kind,
                                            u32 &cact)
{ u32 a = DD_P, c = 0;
#pragma unroll
  for (int j = DD_P - 1; j >= 0; j--)
    { u32 e = v[j].x ^ cur[0];
      if (KW > 1) e |= v[j].y ^ cur[KW > 1 ? 1 : 0];
      if (KW > 2) e |= v[j].z ^ cur[KW > 2 ? 2 : 0];
      const u32  t = min(min(v[j].w, e), v[j].w ^ DD_LOCK);
      const bool h = (t == 0u);
      a = h ? (u32) j : a;
      c = h ? v[j].w : c;
    }
  u32 k = 3u;
  k = (c == DD_LOCK) ? 2u : k;
  k = (c == 0u) ? 1u : k;
  k = (a == (u32) DD_P) ? 0u : k;
  act = (int) a; kind = k; cact = c;
}

// exclusive scan over the 1024 threads of the block.  tmp: DD_WAVES u32 of LDS.
__device__ __forceinline__ u32 dd_block_exscan(u32 v, u32 *tmp, u32 *total)
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
  for (int w = 0; w < DD_WAVES; w++)
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
// single ds_read_b128 yields a consistent (key, count word) pair: count word 0 = empty, DD_LOCK =
// slot claimed with ds_cmpst, key not there yet (its creator then stores key + weight with ONE b128
// write, after the key's tail for wide records; LDS operations of a wave execute in order),
// anything else = published count.
// The whole record is the key, its weight 1; every distinct record comes out once, followed by a dword with its
// multiplicity (records of KW + 1 dwords), in no particular order.
template <int KW>
__global__ __launch_bounds__(DD_THREADS) void k_dd_table(const u32 *__restrict__ recs,
                                                         const u64 *__restrict__ bounds, int kbytes,
                                                         u64 *__restrict__ scal, u32 *__restrict__ table,
                                                         int LIMIT, int gshift, u64 tcap)
{ constexpr int SLOTS = DdCfg<KW>::SLOTS;
  constexpr int U = DD_BATCH;
  FK_DYN_LDS(uint4, dd_lds);
  uint4 *A     = dd_lds;                                   // [SLOTS]
  // LDS byte address of the table for the inline-asm reads (low half of the flat address)
#ifdef FK_HOST_EMU
  const u32 lds_base = 0;
#else
  const u32 lds_base = (u32) (uintptr_t) dd_lds;
#endif
  uint4 *B     = dd_lds + SLOTS;                           // [SLOTS] when KW > 3
  __shared__ u32 sh_claimed, sh_ovf, sh_next, sh_tmp[DD_WAVES];
#ifdef FK_HOST_EMU
  const u32 ovf_addr = 0;
  emu_dd_flag = &sh_ovf;
#else
  const u32 ovf_addr = (u32) (uintptr_t) &sh_ovf;
#endif
  __shared__ u64 sh_base;
  const int tid = threadIdx.x;

  for (int i = tid; i < SLOTS; i += DD_THREADS)
    A[i] = make_uint4(0, 0, 0, 0);
  if (tid == 0) { sh_claimed = 0; sh_ovf = 0; sh_next = 0; }
  u32 my_distinct = 0, my_rounds = 0;
  u32 R0 = 1;
  u32 kmask[KW];                                // key bytes of each record dword (pad and weight off)
#pragma unroll
  for (int w = 0; w < KW; w++)
    kmask[w] = (4 * w + 4 <= kbytes) ? 0xffffffffu : (4 * w < kbytes) ? ((1u << (8 * (kbytes - 4 * w))) - 1u) : 0u;
  __syncthreads();

  // groups of 2^gshift neighbouring bins (one bin when the input is large) are dealt round-robin:
  // hashing makes them equally heavy
  for (u32 bin = blockIdx.x; bin < (DD_BINS >> gshift); bin += gridDim.x)
    { const int64_t beg = (int64_t) bounds[bin << gshift], end = (int64_t) bounds[(bin + 1) << gshift];
      if (beg >= end)
        continue;

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
                    const dd_rec<KW> rr = *(const dd_rec<KW> *) (recs + j * KW);
#pragma unroll
                    for (int w = 0; w < KW; w++)
                      rec[u][w] = rr.w[w];
                  }
              }
          }
          while (base < end)
            { if (FK_WAVE_UNIFORM(*(volatile u32 *) &sh_ovf))       // (one LDS read for the wave: its lanes leave together)
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
                        const dd_rec<KW> rr = *(const dd_rec<KW> *) (recs + j * KW);
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

              // First look for the two records of the batch TOGETHER: the reads of both probes in one LDS round
              // trip, then both claims in flight at once (a wave's LDS operations complete in order, so when the
              // two records of a lane want the same empty slot the second claim sees the first one's lock and
              // goes to the general loop).  Four waves per SIMD cannot hide three dependent LDS round trips per
              // record; two records per lane halve them.  A probe looks at DD_P consecutive slots at once;
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
                  wg_[u]  = 1u;
                  sl_[u] = slot0[u];
                }
              { // a thread claims <= U slots per look at the overflow flag (it arrives with the slots: one LDS
                // round trip), which LIMIT leaves room for
                uint4 v0[U][DD_P];
                u32   ovf_now;
                if (U == 2)
                  dd_read_slots2_flag<SLOTS>(lds_base, sl_[0], sl_[U - 1], v0[0], v0[U - 1], ovf_addr, ovf_now);
                else
                  dd_read_slots_flag<SLOTS>(lds_base, sl_[0], v0[0], ovf_addr, ovf_now);
                bool hit[U];
#pragma unroll
                for (int u = 0; u < U; u++)
                  { int act;
                    u32 cact;
                    dd_classify<KW>(v0[u], cu_[u], act, kind[u], cact);
                    sx[u]   = (sl_[u] + (u32) act) & (SLOTS - 1);
                    todo[u] = ((pend >> u) & 1u) && ovf_now == 0;
                    dn_[u] = !todo[u];
                    hit[u]  = todo[u] && kind[u] == 3u;
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
                      got[u] = atomicCAS(&A[sx[u]].w, 0u, DD_LOCK);
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
                      dd_write_slot(lds_base + sx[u] * 16u, make_uint4(cu_[u][0], KW > 1 ? cu_[u][KW > 1 ? 1 : 0] : 0u,
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
                // whoever is left (lost a race, met a slot being written, no hit in DD_P slots,
                // wide keys, very large counts) goes through the general loop below
#pragma unroll
                for (int u = 0; u < U; u++)
                  if (!dn_[u])
                    sl_[u] = (kind[u] == 1u || kind[u] == 2u) ? sx[u] : (kind[u] == 0u) ? ((sl_[u] + DD_P) & (SLOTS - 1))
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
                      uint4 v[DD_P];
                      dd_read_slots<SLOTS>(lds_base, slot, v);
                      // first slot that is empty (1), being written (2) or holds this k-mer (3)
                      int act;
                      u32 kind, cact;
                      dd_classify<KW>(v, cur, act, kind, cact);
                      const u32 s = (slot + (u32) act) & (SLOTS - 1);
                      if (KW <= 3 && kind == 3u)
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
                                { atomicAdd(&A[s].w, wgt);
                                  done = true;
                                }
                            }
                          else if (kind == 1u)
                            { if (atomicCAS(&A[s].w, 0u, DD_LOCK) == 0u)
                                { // key and weight appear together: the (wider) key's tail first, then ONE 16-byte write of
                                  // key + count word -- LDS serves a wave's operations in order and an aligned 16-byte
                                  // access in one piece, so no reader can pair this count with another key
                                  if (KW > 3)
                                    { B[s] = make_uint4(cur[KW > 3 ? 3 : 0], KW > 4 ? cur[KW > 4 ? 4 : 0] : 0u,
                                                        KW > 5 ? cur[KW > 5 ? 5 : 0] : 0u, KW > 6 ? cur[KW > 6 ? 6 : 0] : 0u);
                                      asm volatile("" ::: "memory");
                                    }
                                  dd_write_slot(lds_base + s * 16u, make_uint4(cur[0], KW > 1 ? cur[KW > 1 ? 1 : 0] : 0u,
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
                            slot = (slot + DD_P) & (SLOTS - 1);
                          const u64 cm = FK_BALLOT_ACTIVE(created);      // (over the lanes still in this loop)
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
          __syncthreads();
          const bool ovf = (sh_ovf != 0);
          const u32  fill = sh_claimed;
          __syncthreads();
          if (ovf)
            { // more distinct k-mers than the table takes: halve the selection and start it again
              for (int i = tid; i < SLOTS; i += DD_THREADS)
                A[i].w = 0;
              if (tid == 0) { sh_claimed = 0; sh_ovf = 0; sh_next = 0; }
              my_rounds += (tid == 0);
              bin_ovf = true;
              __syncthreads();
              if (R >= DD_MAXR)
                { if (tid == 0)
                    atomicAdd(&scal[3], 1ull);
                  failed = true;
                  break;
                }
              R <<= 1;
              continue;
            }
          bin_fill = max(bin_fill, fill);

          // ---- emit: every claimed slot is a distinct record; the table is left empty
          u32 c[SLOTS / DD_THREADS];
          u32 nq = 0;
#pragma unroll
          for (int j = 0; j < SLOTS / DD_THREADS; j++)
            c[j] = A[j * DD_THREADS + tid].w;
#pragma unroll
          for (int j = 0; j < SLOTS / DD_THREADS; j++)
            if (c[j] != 0)
              { A[j * DD_THREADS + tid].w = 0;
                nq += 1;
              }
          my_distinct += nq;
          { u32 tot;
            const u32 off = dd_block_exscan(nq, sh_tmp, &tot);
            if (tid == 0 && tot > 0)
              sh_base = atomicAdd(&scal[2], (u64) tot);
            __syncthreads();
            if (tot > 0 && sh_base + tot > tcap)
              { if (tid == 0)                         // the output buffer is full
                  atomicAdd(&scal[6], 1ull);
              }
            else if (tot > 0)
              { u64 o = sh_base + off;
#pragma unroll
                for (int j = 0; j < SLOTS / DD_THREADS; j++)
                  if (c[j] != 0)
                    { const int slot = j * DD_THREADS + tid;
                      const uint4 a = A[slot];
                      u32 kd[7] = { a.x, a.y, a.z, 0u, 0u, 0u, 0u };
                      if (KW > 3)
                        { const uint4 b = B[slot];
                          kd[3] = b.x; kd[4] = b.y; kd[5] = b.z; kd[6] = b.w;
                        }
                      dd_rec<KW + 1> ro;
#pragma unroll
                      for (int w = 0; w < KW; w++)
                        ro.w[w] = kd[w];
                      ro.w[KW] = c[j];
                      *(dd_rec<KW + 1> *) (table + o * (KW + 1)) = ro;
                      o += 1;
                    }
              }
          }
          if (tid == 0) { sh_claimed = 0; sh_next = 0; }
          __syncthreads();

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
        R0 = min(R0 << 1, (u32) DD_MAXR);
      else if (R0 > 1 && bin_fill * 9 < (u32) LIMIT * 4)
        R0 >>= 1;
    }

  u64 d = my_distinct, rd = my_rounds;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
    { d  += __shfl_down(d, o, 64);
      rd += __shfl_down(rd, o, 64);
    }
  if (fk_lane() == 0)
    { if (d)  atomicAdd(&scal[1], d);
      if (rd) atomicAdd(&scal[5], rd);
    }
}

// ---------------------------------------------------------------------------------------------
// Super-mer de-duplication: n records of KW dwords, grouped by 16 hash bits of
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
  u64 *d_bounds = (u64 *) fk_slot(ctx, FK_SLOT_AG_BOUNDS, (DD_BINS + 1) * 8);
  u64 *d_hist   = (u64 *) fk_slot(ctx, FK_SLOT_CT_HIST, (FK_HIST_BINS + 16) * 8);
  if (d_bounds == NULL || d_hist == NULL)
    return (FK_ENOMEM);
  if (n >= (int64_t) 1 << 31)                           // a multiplicity must stay below the lock bit of a count word
    return (FK_ESTATE);
  u64 *d_scal = d_hist + FK_HIST_BINS;
  static bool attr_set = false;
  const size_t lds = DdCfg<KW>::LDS;
  if (!attr_set)
    { auto kern = k_dd_table<KW>;
      FK_HIP(ctx, hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
      attr_set = true;
    }
  FK_HIP(ctx, hipMemsetAsync(d_scal, 0, 8 * 8, s));
  hipLaunchKernelGGL(k_dd_bounds<KW>, dim3(DD_BINS / 256 + 1), dim3(256), 0, s, (const u32 *) d_grouped, n,
                     KW * 4, d_bounds);
  const int cus = ctx->num_cus > 0 ? ctx->num_cus : 256;
  int gshift = 0;
  while (gshift < 16 && (n >> (16 - gshift)) < 6000)         // most of a batch of records per table fill
    gshift += 1;
  hipLaunchKernelGGL((k_dd_table<KW>), dim3((unsigned) cus), dim3(DD_THREADS), lds, s, (const u32 *) d_grouped,
                     (const u64 *) d_bounds, KW * 4, d_scal, (u32 *) d_out, DdCfg<KW>::LIMIT, gshift, (u64) cap);
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
